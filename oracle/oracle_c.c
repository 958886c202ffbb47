/*
 * oracle_c.c - plain-C CPU restatement of the DeepMIMO channel-generation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under deepmimo_amd/ may link, load or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and there only as the checker / the timed CPU
 * baseline.  Parity status: PINNED - tests/test_oracle_golden.py checks it against the vectors oracle/gen_golden.py
 * wrote from the real reference (tests/golden/g*.npz) and against oracle_np.py on seeded random inputs.
 *
 * These are the "CPU twins" of the product's C-ABI: dmx_cpu_* take the structs of
 * include/deepmimo_amd.h with HOST pointers and the same argument meaning, so a parity test reads
 *     dmx_path_prep(...); dmx_channels_fd(...)      on the GPU   vs
 *     dmx_cpu_path_prep(...); dmx_cpu_channels_fd(...)   here.
 * It is an independent second restatement (scalar loops, libm) next to the vectorised NumPy one: the two agree to
 * ~1e-15 relative and disagree only where NumPy's SIMD float64 routines and glibc round differently.
 *
 * The reference's dtype flow is reproduced on purpose (see DESIGN.md "numerics"): which intermediates are float32
 * decides the result at the 1e-5 level.  Reference paths are relative to /root/reference/deepmimo/.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/deepmimo_amd.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define LIGHTSPEED 299792458.0 /* deepmimo_v3/consts.py:112 */
#define TWO_PI 6.283185307179586
#define HALF_PI 1.5707963267948966
#define KC 16 /* subcarriers per register block of the contraction */

/* per kept path record (the CPU twin's workspace; the GPU library has its own compact layout) */
typedef struct cpu_path {
    double aod_el, aod_az, aoa_el, aoa_az; /* rotated, FoV-masked radians (NaN = masked / absent) */
    double power;                          /* linear W incl. antenna gains; NaN = path absent (channel.py:260) */
    float delay, phase, vel, acc;
    int32_t power_is_f32;                  /* isotropic both ends: the reference keeps float32 (ant_patterns.py:167) */
    int32_t pad;
} cpu_path;

size_t dmx_cpu_workspace_bytes(const dmx_params* prm, int64_t n_ue, int32_t n_paths_loaded) {
    int32_t P = prm->num_paths < n_paths_loaded ? prm->num_paths : n_paths_loaded;
    if (P < 0) P = 0;
    return (size_t)n_ue * (size_t)P * sizeof(cpu_path);
}

/* NumPy's float32 sin / cos (x86 SIMD routine: 3-term Cody-Waite reduction by pi/2, degree-9 / degree-8
 * polynomials, quadrant select).  np.sin / np.cos of the float32 zenith angle at generator/geometry.py:301-302 go
 * through it, and its last-bit behaviour reaches the rotated angles at the 1e-8 level - libm sinf would not do. */
static void np_sincosf(float x, float* s_out, float* c_out) {
    if (!(fabsf(x) <= 71476.0625f)) { *s_out = sinf(x); *c_out = cosf(x); return; }
    volatile float qv = x * 0x1.45f306p-1f;
    qv = qv + 0x1.8p+23f;
    qv = qv - 0x1.8p+23f; /* round to nearest integer */
    const float q = qv;
    float r = fmaf(q, -0x1.921fb0p+00f, x);
    r = fmaf(q, -0x1.5110b4p-22f, r);
    r = fmaf(q, -0x1.846988p-48f, r);
    const float r2 = r * r;
    float sp = fmaf(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
    sp = fmaf(sp, r2, 0x1.11119ap-07f);
    sp = fmaf(sp, r2, -0x1.555556p-03f);
    sp = fmaf(sp, r2, 0.0f);
    sp = fmaf(sp, r, r);
    float cp = fmaf(0x1.98e616p-16f, r2, -0x1.6c06dcp-10f);
    cp = fmaf(cp, r2, 0x1.55553cp-05f);
    cp = fmaf(cp, r2, -0.5f);
    cp = fmaf(cp, r2, 1.0f);
    const int iq = (int)q, iqc = iq + 1;
    const float sv = (iq & 1) ? cp : sp, cv = (iqc & 1) ? cp : sp;
    *s_out = (iq & 2) ? -sv : sv;
    *c_out = (iqc & 2) ? -cv : cv;
}

/* _rotate_angles_batch, generator/geometry.py:244-319 - rot = (about x, about y, about z) radians */
static void rotate_angles(float el_deg, float az_deg, const double rot[3], double* th_rot, double* ph_rot) {
    const float d2r = (float)(M_PI / 180.0);
    const float th = el_deg * d2r, ph = az_deg * d2r; /* np.deg2rad(float32) stays float32, :284 */
    float st32, ct32;
    np_sincosf(th, &st32, &ct32); /* :301-302 */
    const double st = st32, ct = ct32;
    const double d = (double)ph - rot[2]; /* float32 - float64 -> float64, :294 */
    const double sd = sin(d), cd = cos(d);
    const double sx = sin(rot[0]), cx = cos(rot[0]), sy = sin(rot[1]), cy = cos(rot[1]);
    *th_rot = acos(cy * cx * ct + st * (sy * cx * cd - sx * sd));                               /* :305-306 */
    *ph_rot = atan2(cy * sx * ct + st * (sy * sx * cd + cx * sd), cy * st * cd - sy * ct);      /* :308-310 */
}

static double pymod_2pi(double x) { /* np.mod: sign of the divisor */
    double m = fmod(x, TWO_PI);
    if (m != 0.0) { if (m < 0.0) m += TWO_PI; } else { m = 0.0; }
    return m;
}

/* _apply_FoV_batch, generator/geometry.py:162-195; fov = [horizontal, vertical] radians */
static int in_fov(double th, double ph, const double fov[2]) {
    const double t = pymod_2pi(th), p = pymod_2pi(ph);
    const int az = (p <= 0 + fov[0] / 2) || (p >= TWO_PI - fov[0] / 2);
    const int el = (t <= HALF_PI + fov[1] / 2) && (t >= HALF_PI - fov[1] / 2);
    return az && el;
}

/* _pattern_halfwave_dipole, generator/ant_patterns.py:34-71 (NaN -> 0) */
static double dipole_gain(double th) {
    const double s = sin(th);
    if (!(fabs(s) > 1e-10)) return 0.0;
    const double c = cos(HALF_PI * cos(th));
    return 1.643 * (c * c / s);
}

static uint32_t float_order_key(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

float dmx_cpu_decode_max_delay(uint32_t key) {
    if (key == 0) return NAN;
    uint32_t b = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

/*
 * Twin of dmx_path_prep: Dataset._compute_rotated_angles (generator/dataset.py:310-356), _compute_fov (:461-512),
 * _compute_power_linear_ant_gain (:665-696), _compute_num_paths (:613-619), _compute_los (:569-611).
 * Side arrays cover ALL loaded paths; the workspace keeps the first P = min(num_paths, n_paths) (dataset.py:258-261).
 */
int dmx_cpu_path_prep(const dmx_rays* rays, const dmx_params* prm, void* workspace, size_t workspace_bytes,
                      const dmx_side* side, void* stream) {
    (void)stream;
    if (!rays || !prm || !workspace) return DMX_ERR_ARG;
    const int64_t N = rays->n_ue;
    const int L = rays->n_paths;
    const int P = prm->num_paths < L ? prm->num_paths : L;
    if (workspace_bytes < dmx_cpu_workspace_bytes(prm, N, L)) return DMX_ERR_WORKSPACE;
    cpu_path* ws = (cpu_path*)workspace;
    const int iso = prm->bs_pattern == DMX_PATTERN_ISOTROPIC && prm->ue_pattern == DMX_PATTERN_ISOTROPIC;
    float maxd = -INFINITY;
    for (int64_t u = 0; u < N; ++u) {
        double ue_rot[3];
        if (prm->ue_rotation_per_user) {
            for (int i = 0; i < 3; ++i) ue_rot[i] = prm->ue_rotation_per_user[3 * u + i] * (M_PI / 180.0); /* np.deg2rad, :286 */
        } else {
            memcpy(ue_rot, prm->ue_rotation, sizeof ue_rot);
        }
        int count = 0, has = 0;
        float first = NAN;
        for (int l = 0; l < L; ++l) {
            const size_t i = (size_t)u * rays->ld + l, o = (size_t)u * L + l;
            double aod_el, aod_az, aoa_el, aoa_az;
            rotate_angles(rays->aod_el[i], rays->aod_az[i], prm->bs_rotation, &aod_el, &aod_az);
            rotate_angles(rays->aoa_el[i], rays->aoa_az[i], ue_rot, &aoa_el, &aoa_az);
            if (side && side->aod_el_rot) side->aod_el_rot[o] = aod_el;
            if (side && side->aod_az_rot) side->aod_az_rot[o] = aod_az;
            if (side && side->aoa_el_rot) side->aoa_el_rot[o] = aoa_el;
            if (side && side->aoa_az_rot) side->aoa_az_rot[o] = aoa_az;
            int keep = 1;
            if (prm->fov_enabled) { /* dataset.py:484-511 */
                if (prm->bs_fov_restricted) keep &= in_fov(aod_el, aod_az, prm->bs_fov);
                if (prm->ue_fov_restricted) keep &= in_fov(aoa_el, aoa_az, prm->ue_fov);
                if (side && side->fov_mask) side->fov_mask[o] = (uint8_t)keep;
                if (!keep) aod_el = aod_az = aoa_el = aoa_az = NAN;
                if (keep && !has) { has = 1; first = rays->inter[i]; } /* first in-FoV path, :594-598 */
            }
            if (!isnan(aoa_az)) ++count; /* :613-619 */
            const float pl = powf(10.0f, rays->power[i] / 10.0f); /* dbw2watt, generator_utils.py:35 */
            double pag;
            if (iso) {
                pag = pl; /* float32 * python 1.0 stays float32 */
            } else {
                const double gt = prm->bs_pattern == DMX_PATTERN_ISOTROPIC ? 1.0 : dipole_gain(aod_el);
                const double gr = prm->ue_pattern == DMX_PATTERN_ISOTROPIC ? 1.0 : dipole_gain(aoa_el);
                pag = (double)pl * (gt * gr); /* ant_patterns.py:167-168 */
            }
            if (side && side->power_linear) side->power_linear[o] = pl;
            if (side && side->power_linear_ant_gain) side->power_linear_ant_gain[o] = pag;
            if (l < P) {
                cpu_path* w = &ws[(size_t)u * P + l];
                w->aod_el = aod_el; w->aod_az = aod_az; w->aoa_el = aoa_el; w->aoa_az = aoa_az;
                w->power = pag; w->power_is_f32 = iso; w->pad = 0;
                w->delay = rays->delay[i]; w->phase = rays->phase[i];
                w->vel = rays->doppler_vel ? rays->doppler_vel[i] : 0.0f;
                w->acc = rays->doppler_acc ? rays->doppler_acc[i] : 0.0f;
                if (rays->delay[i] > maxd) maxd = rays->delay[i]; /* np.nanmax, channel.py:231 */
            }
        }
        if (!prm->fov_enabled) { has = count > 0; first = rays->inter[(size_t)u * rays->ld]; } /* :600-606 */
        if (side && side->num_paths) side->num_paths[u] = count;
        if (side && side->los) side->los[u] = !has ? -1 : (first == 0.0f ? 1 : 0);
    }
    if (side && side->max_delay_key && maxd > -INFINITY) {
        const uint32_t k = float_order_key(maxd);
        if (k > *side->max_delay_key) *side->max_delay_key = k;
    }
    return DMX_OK;
}

/* _array_response_batch, generator/geometry.py:38-102: element m = y + Mh z, x index always 0 */
static void array_response(const int32_t shape[2], double spacing, double theta, double phi, double complex* a) {
    const int M = shape[0] * shape[1];
    if (isnan(theta)) { /* :65-80 - NaN angle -> zero column */
        for (int m = 0; m < M; ++m) a[m] = 0.0;
        return;
    }
    const double kd = 2 * M_PI * spacing; /* dataset.py:393 */
    const double gy = kd * sin(theta) * sin(phi), gz = kd * cos(theta);
    for (int m = 0; m < M; ++m) {
        const double arg = (double)(m % shape[0]) * gy + (double)(m / shape[0]) * gz;
        a[m] = CMPLX(cos(arg), sin(arg));
    }
}

/* c_l of OFDM_PathGenerator.generate (generator/channel.py:182-192) for one path; *dn_out = clipped delay in
 * samples.  Isotropic: complex64 arithmetic like the reference; dipole: float64 power -> complex128. */
static double complex path_coeff(const cpu_path* w, const dmx_params* prm, float* dn_out) {
    const float ts = (float)(1.0 / prm->bandwidth); /* float32 array / weak python float, :223, :182 */
    float dn = w->delay / ts;
    double pw = w->power;
    if (dn >= (float)prm->n_subcarriers) { pw = 0.0; dn = (float)prm->n_subcarriers; } /* :187-189 */
    *dn_out = dn;
    const float ph = w->phase * (float)(M_PI / 180.0);
    const float complex e = cexpf(CMPLXF(0.0f, ph)); /* np.exp(1j * float32) is complex64 */
    if (w->power_is_f32) {
        const float a = sqrtf((float)pw / (float)prm->n_subcarriers);
        return CMPLX((double)(a * crealf(e)), (double)(a * cimagf(e)));
    }
    const double a = sqrt(pw / (double)prm->n_subcarriers);
    return CMPLX(a * (double)crealf(e), a * (double)cimagf(e));
}

static double np_sinc(double x) { /* np.sinc */
    const double y = M_PI * (x == 0.0 ? 1.0e-20 : x);
    return sin(y) / y;
}

/* v3 Doppler factor (deepmimo_v3/generator/python/construct_deepmimo.py:267-280) in the no-filter branch: every
 * factor is a float32 array there, so the phase is float32 and the exponential complex64. */
static double complex doppler_f32(const cpu_path* w, double fc) {
    const float tau = w->delay;
    const float x = (w->vel * tau) / (float)LIGHTSPEED + (w->acc * (tau * tau)) / (float)(2 * LIGHTSPEED);
    const float ph = (float)(-2 * M_PI * fc) * x;
    const float complex e = cexpf(CMPLXF(0.0f, ph));
    return CMPLX((double)crealf(e), (double)cimagf(e));
}

/*
 * Twin of dmx_channels_fd / dmx_channels_fd_lpf (prm->rx_filter selects): the user loop of _generate_MIMO_channel,
 * generator/channel.py:264-284, on top of _compute_array_response_product, generator/dataset.py:398-417.
 * complex128 accumulation, complex64 store; a path with any NaN factor contributes nothing (np.nansum, :283).
 */
int dmx_cpu_channels_fd(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                        int64_t user_begin, int64_t user_count, void* out_c64, int32_t variant, void* stream) {
    (void)variant; (void)stream;
    if (!prm || !workspace || !out_c64) return DMX_ERR_ARG;
    if (user_begin < 0 || user_count < 0 || user_begin + user_count > n_ue) return DMX_ERR_ARG;
    const int P = prm->num_paths < n_paths_loaded ? prm->num_paths : n_paths_loaded;
    const int Mr = prm->ue_shape[0] * prm->ue_shape[1], Mt = prm->bs_shape[0] * prm->bs_shape[1];
    const int K = prm->n_selected, N = prm->n_subcarriers;
    const cpu_path* ws = (const cpu_path*)workspace;
    float complex* out = (float complex*)out_c64;
    const double x = 2 * M_PI / N;
    const int use_dop = prm->enable_doppler;
    int rc = DMX_OK;
#pragma omp parallel
    {
        double complex* at = malloc(sizeof(double complex) * (size_t)Mt * (P ? P : 1));
        double complex* ar = malloc(sizeof(double complex) * (size_t)Mr * (P ? P : 1));
        double complex* g = malloc(sizeof(double complex) * (size_t)K * (P ? P : 1));
        double complex* taps = prm->rx_filter ? malloc(sizeof(double complex) * (size_t)N) : NULL;
        uint8_t* ok = malloc((size_t)(P ? P : 1));
        double* gre = malloc(sizeof(double) * (size_t)K * (P ? P : 1));
        double* gim = malloc(sizeof(double) * (size_t)K * (P ? P : 1));
        double* hre = malloc(sizeof(double) * (size_t)(K ? K : 1));
        double* him = malloc(sizeof(double) * (size_t)(K ? K : 1));
        double* wre = malloc(sizeof(double) * (size_t)(P ? P : 1));
        double* wim = malloc(sizeof(double) * (size_t)(P ? P : 1));
        int* wl = malloc(sizeof(int) * (size_t)(P ? P : 1));
        if (!at || !ar || !g || !ok || !gre || !gim || !hre || !him || !wre || !wim || !wl || (prm->rx_filter && !taps)) {
#pragma omp critical
            rc = DMX_ERR_ARG;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t uu = 0; uu < user_count; ++uu) {
                const cpu_path* row = ws + (size_t)(user_begin + uu) * P;
                float complex* H = out + (size_t)uu * Mr * Mt * K;
                for (int l = 0; l < P; ++l) {
                    const cpu_path* w = &row[l];
                    ok[l] = 0;
                    if (isnan(w->power)) continue; /* channel.py:260 */
                    array_response(prm->bs_shape, prm->bs_spacing, w->aod_el, w->aod_az, at + (size_t)l * Mt);
                    array_response(prm->ue_shape, prm->ue_spacing, w->aoa_el, w->aoa_az, ar + (size_t)l * Mr);
                    float dn;
                    const double complex c = path_coeff(w, prm, &dn);
                    double complex* gl = g + (size_t)l * K;
                    if (prm->rx_filter) { /* channel.py:166-168, 193-194 */
                        for (int d = 0; d < N; ++d) {
                            double complex t = c * np_sinc((double)d - (double)dn);
                            if (use_dop) {
                                const double tau = (1.0 / prm->bandwidth) * d;
                                const double ph = -2 * M_PI * prm->carrier_freq *
                                                  ((double)w->vel * tau / LIGHTSPEED + (double)w->acc * (tau * tau) / (2 * LIGHTSPEED));
                                t *= CMPLX(cos(ph), sin(ph));
                            }
                            taps[d] = t;
                        }
                        for (int k = 0; k < K; ++k) {
                            double complex s = 0.0;
                            const int64_t sc = prm->selected_subcarriers[k];
                            for (int d = 0; d < N; ++d) {
                                const double y = -x * (double)((int64_t)d * sc);
                                s += taps[d] * CMPLX(cos(y), sin(y));
                            }
                            gl[k] = s;
                        }
                    } else { /* channel.py:196-197 */
                        const double complex dop = use_dop ? doppler_f32(w, prm->carrier_freq) : 1.0;
                        for (int k = 0; k < K; ++k) {
                            const double y = -x * ((double)dn * (double)prm->selected_subcarriers[k]);
                            double complex v = c * CMPLX(cos(y), sin(y));
                            if (use_dop) v *= dop;
                            gl[k] = v;
                        }
                    }
                    /* nansum: a NaN anywhere in this path's factors makes every product of the path NaN -> 0 */
                    int bad = isnan(creal(gl[0])) || isnan(cimag(gl[0]));
                    for (int m = 0; m < Mt && !bad; ++m) bad = isnan(creal(at[(size_t)l * Mt + m])) || isnan(cimag(at[(size_t)l * Mt + m]));
                    for (int m = 0; m < Mr && !bad; ++m) bad = isnan(creal(ar[(size_t)l * Mr + m])) || isnan(cimag(ar[(size_t)l * Mr + m]));
                    for (int k = 1; k < K && !bad; ++k) bad = isnan(creal(gl[k])) || isnan(cimag(gl[k]));
                    ok[l] = !bad;
                }
                /* g as split re / im rows so the k loop vectorises; same complex128 products and sums */
                for (int l = 0; l < P; ++l)
                    if (ok[l])
                        for (int k = 0; k < K; ++k) { gre[(size_t)l * K + k] = creal(g[(size_t)l * K + k]); gim[(size_t)l * K + k] = cimag(g[(size_t)l * K + k]); }
                for (int r = 0; r < Mr; ++r)
                    for (int t = 0; t < Mt; ++t) {
                        float complex* h = H + ((size_t)r * Mt + t) * K;
                        int nw = 0;
                        for (int l = 0; l < P; ++l) {
                            if (!ok[l]) continue;
                            const double complex w = ar[(size_t)l * Mr + r] * at[(size_t)l * Mt + t]; /* dataset.py:417 */
                            wre[nw] = creal(w); wim[nw] = cimag(w); wl[nw] = l; ++nw;
                        }
                        for (int k0 = 0; k0 < K; k0 += KC) { /* KC subcarriers at a time stay in registers over the path sum */
                            const int kn = K - k0 < KC ? K - k0 : KC;
                            double sr[KC], si[KC];
                            for (int k = 0; k < KC; ++k) sr[k] = si[k] = 0.0;
                            for (int j = 0; j < nw; ++j) {
                                const double wr = wre[j], wi = wim[j];
                                const double* restrict gr = gre + (size_t)wl[j] * K + k0;
                                const double* restrict gi = gim + (size_t)wl[j] * K + k0;
                                if (kn == KC) {
                                    for (int k = 0; k < KC; ++k) {
                                        sr[k] += wr * gr[k] - wi * gi[k];
                                        si[k] += wr * gi[k] + wi * gr[k];
                                    }
                                } else {
                                    for (int k = 0; k < kn; ++k) {
                                        sr[k] += wr * gr[k] - wi * gi[k];
                                        si[k] += wr * gi[k] + wi * gr[k];
                                    }
                                }
                            }
                            for (int k = 0; k < kn; ++k) { hre[k0 + k] = sr[k]; him[k0 + k] = si[k]; }
                        }
                        for (int k = 0; k < K; ++k) h[k] = CMPLXF((float)hre[k], (float)him[k]);
                    }
            }
        }
        free(at); free(ar); free(g); free(taps); free(ok); free(gre); free(gim); free(hre); free(him); free(wre); free(wim); free(wl);
    }
    return rc;
}

/* Twin of dmx_channels_td: the time-domain branch, generator/channel.py:285-287 - valid paths compacted to the front. */
int dmx_cpu_channels_td(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                        int64_t user_begin, int64_t user_count, void* out_c64, void* stream) {
    (void)stream;
    if (!prm || !workspace || !out_c64) return DMX_ERR_ARG;
    if (user_begin < 0 || user_count < 0 || user_begin + user_count > n_ue) return DMX_ERR_ARG;
    const int P = prm->num_paths < n_paths_loaded ? prm->num_paths : n_paths_loaded;
    const int Mr = prm->ue_shape[0] * prm->ue_shape[1], Mt = prm->bs_shape[0] * prm->bs_shape[1];
    const cpu_path* ws = (const cpu_path*)workspace;
    float complex* out = (float complex*)out_c64;
    double complex* at = malloc(sizeof(double complex) * (size_t)(Mt ? Mt : 1));
    double complex* ar = malloc(sizeof(double complex) * (size_t)(Mr ? Mr : 1));
    if (!at || !ar) { free(at); free(ar); return DMX_ERR_ARG; }
    for (int64_t uu = 0; uu < user_count; ++uu) {
        const cpu_path* row = ws + (size_t)(user_begin + uu) * P;
        float complex* H = out + (size_t)uu * Mr * Mt * P;
        for (size_t i = 0; i < (size_t)Mr * Mt * P; ++i) H[i] = 0.0f;
        int s = 0;
        for (int l = 0; l < P; ++l) {
            const cpu_path* w = &row[l];
            if (isnan(w->power)) continue;
            array_response(prm->bs_shape, prm->bs_spacing, w->aod_el, w->aod_az, at);
            array_response(prm->ue_shape, prm->ue_spacing, w->aoa_el, w->aoa_az, ar);
            const float ph = w->phase * (float)(M_PI / 180.0);
            const float complex e = cexpf(CMPLXF(0.0f, ph));
            double complex pg;
            if (w->power_is_f32) {
                const float a = sqrtf((float)w->power);
                pg = CMPLX((double)(a * crealf(e)), (double)(a * cimagf(e)));
            } else {
                const double a = sqrt(w->power);
                pg = CMPLX(a * (double)crealf(e), a * (double)cimagf(e));
            }
            for (int r = 0; r < Mr; ++r)
                for (int t = 0; t < Mt; ++t) {
                    const double complex v = (ar[r] * at[t]) * pg;
                    H[((size_t)r * Mt + t) * P + s] = CMPLXF((float)creal(v), (float)cimag(v));
                }
            ++s;
        }
    }
    free(at); free(ar);
    return DMX_OK;
}

int dmx_cpu_version(void) { return DMX_ABI_VERSION; }

/* threads of the user loop in dmx_cpu_channels_fd (the reference itself is single-threaded) */
void dmx_cpu_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
