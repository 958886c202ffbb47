// Stage 1 - path prep: one lane per path; a wavefront serves two users (32 lanes each) when the
// scenario has at most 32 loaded paths (DeepMIMO's MAX_PATHS is 25), one user otherwise.
//
// Replaces, for all users at once, the vectorised NumPy prologue of Dataset.compute_channels:
//   _rotate_angles_batch        geometry.py:244-319   (via dataset.py:310-356)
//   _apply_FoV_batch            geometry.py:162-195   (via dataset.py:461-512)
//   dbw2watt + AntennaPattern   generator_utils.py:35, ant_patterns.py:21-71, 145-168
//   _compute_num_paths / _los   dataset.py:569-619
//   the k-independent part of OFDM_PathGenerator.generate   channel.py:182-192
//   the array-response phase    geometry.py:85-102
// and emits per-path records (dmx_common.h) with the contributing paths compacted to the front
// of each user's row (wave ballot + prefix popcount), so stage 2 never touches a dead path.
//
// Numerics follow the reference's dtype flow on purpose (DESIGN.md "numerics"): deg2rad in
// float32, sin/cos of the zenith angle rounded to float32, everything touching the rotation in
// float64, tau/Ts as a float32 division.  This file is compiled with -ffp-contract=off so the
// float64 expressions associate exactly as NumPy evaluates them.  HBM traffic is ~40 B read and
// ~50 B written per path: noise next to stage 2's output stream, so no tuning beyond coalescing
// (lane = path => each field is one contiguous row segment per wave).
#include "dmx_common.h"
#include <math.h>

namespace dmx {

struct PrepArgs {
    dmx_rays rays;
    dmx_side side;
    WsView ws;
    // rotation
    double bsx, csx, bsy, csy, brz;       // sin/cos of BS rotation about x, y; rotation about z (rad)
    double usx, ucx, usy, ucy, urz;       // same for a constant UE rotation
    const double* ue_rot_pu;              // [n,3] degrees or nullptr
    // fov
    int fov_enabled, bs_restricted, ue_restricted;
    double bs_fh, bs_fv, ue_fh, ue_fv;    // radians
    int bs_pat, ue_pat;
    double bs_spacing, ue_spacing;
    int P;                                 // paths used for the channel
    int freq_domain;
    int n_sc;
    float ts32;                            // float32(1/bandwidth)
    int doppler;
    int rx_filter;
    int need_angles;                       // angles wanted as numbers (side outputs, FoV, dipole), not just directions
    int sort_paths;                        // frequency domain: kept paths ordered by falling amplitude (the sum does not care;
                                           // stage 2 drops product terms of a weak last K-step, k2_channel_fd_mfma.hip)
    double fc;
};

static constexpr float D2R_F = 0.017453292519943295f;       // float32(pi/180): np.deg2rad on float32
static constexpr double D2R_D = 0.017453292519943295;       // np.deg2rad on float64
static constexpr double TWO_PI = 6.283185307179586;
static constexpr double HALF_PI = 1.5707963267948966;
static constexpr double LIGHTSPEED = 299792458.0;           // deepmimo_v3/consts.py:112

// NumPy's float32 sin / cos (the SIMD loops np.sin / np.cos dispatch to for float32 arrays on x86
// with FMA): Cody-Waite reduction by pi/2 in three float32 constants, degree-9 / degree-8 minimax
// polynomials, all in float32 FMAs.  The zenith sin/cos feed arccos / atan2, which amplify a
// 1-ulp float32 difference by 1/sin(zenith) towards the rotated poles, so K1 reproduces that
// routine operation for operation; validated bit-for-bit against np.sin / np.cos on 2e6 inputs
// (tests/test_oracle_golden.py::test_numpy_f32_sincos_model).  |x| beyond the routine's range
// (7e4) falls back to sinf / cosf like NumPy falls back to libm.
__device__ __forceinline__ void np_sincosf(float x, float& s_out, float& c_out) {
    // NaN - "no path": the padding lanes of every wave and most paths of a ray-traced user - must take the polynomial (it
    // propagates NaN like np.sin does), not this branch: the wave would execute libm's large-argument sinf AND cosf for it
    if (fabsf(x) > 71476.0625f) { s_out = sinf(x); c_out = cosf(x); return; }
    float q = x * 0x1.45f306p-1f;
    q = (q + 0x1.8p+23f) - 0x1.8p+23f;                       // round to nearest integer
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
    const int iq = (int)q;
    const int iqc = iq + 1;
    float sv = (iq & 1) ? cp : sp;
    float cv = (iqc & 1) ? cp : sp;
    s_out = (iq & 2) ? -sv : sv;
    c_out = (iqc & 2) ? -cv : cv;
}

// float64 sin / cos for the LEAN instantiations (nothing but the channel depends on them: no angle output, no FoV
// compare, no dipole gain - those keep the library call, whose last-bit behaviour the FoV masks were validated with).
// Cody-Waite reduction by pi/2 in two constants (exact for |k| < 2^20) and the fdlibm kernels: 2.2e-16 against long
// double on 2e7 arguments.  |x| >= 1e5 (never an angle in degrees times pi/180) takes the library call; NaN - the padding
// lanes of every wave - must NOT: a first version sent NaN there too and every wave executed both forms.
__device__ __forceinline__ void sincos_lean(double x, double& s_out, double& c_out) {
    if (fabs(x) >= 1.0e5) { sincos(x, &s_out, &c_out); return; }
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    const int q = (int)k;                                    // NaN -> 0: the NaN of sn / cs goes through
    const double sv = (q & 1) ? cs : sn, cv = (q & 1) ? sn : cs;
    s_out = (q & 2) ? -sv : sv;
    c_out = ((q + 1) & 2) ? -cv : cv;
}

// geometry.py:284-310 for one path: the rotated direction as (cos zenith', re, im) with
//   zenith' = arccos(zc)  (geometry.py:305-306),  azimuth' = angle(re + j im)  (geometry.py:308-310).
// The angles themselves are only materialised when something needs them (side outputs, FoV, dipole
// pattern); the array-response steps use sin(zenith') = sqrt(1 - zc^2), sin(azimuth') = im / |re + j im|
// and cos(zenith') = zc, which are the same numbers without three float64 trig calls per array side.
template <bool LEAN>
__device__ __forceinline__ void rotate_dir(float el_deg, float az_deg, double sx, double cx, double sy,
                                           double cy, double rz, double& zc, double& re, double& im) {
    const float th32 = el_deg * D2R_F;
    const float ph32 = az_deg * D2R_F;
    float st32, ct32;
    np_sincosf(th32, st32, ct32);                           // np.sin / np.cos of float32 stay float32
    const double st = (double)st32, ct = (double)ct32;
    const double d = (double)ph32 - rz;
    double sd, cd;
    if constexpr (LEAN) {
        sincos_lean(d, sd, cd);
    } else {
        // the library call on a finite stand-in for NaN ("no path": padding lanes, most paths of a ray-traced user), so that
        // no wave walks its large-argument reduction for them; finite arguments get the very same bits as before
        const bool bad = isnan(d);
        sincos(bad ? 0.0 : d, &sd, &cd);
        if (bad) { sd = d; cd = d; }
    }
    zc = cy * cx * ct + st * (sy * cx * cd - sx * sd);
    re = cy * st * cd - sy * ct;
    im = cy * sx * ct + st * (sy * sx * cd + cx * sd);
}

// The same for an EXACTLY zero rotation (DeepMIMO's default, channel.py:36-46): with sin = 0 and cos = 1 every product of
// geometry.py:294-310 that carries a rotation term is an exact zero and the sums are exact, so
//   zc = cos(zenith),  re = sin(zenith) cos(azimuth),  im = sin(zenith) sin(azimuth)
// are the very numbers the general expressions give (NaN inputs propagate the same way) - 4 float64 operations instead
// of 20.  sphi = sin(azimuth') = im / |re + j im| is sign(sin zenith) sin(azimuth) up to the 1e-16 by which the float64
// sin / cos pair misses the unit circle (the general path divides by that norm): no square root, no division.
__device__ __forceinline__ void rotate_dir_zero(float el_deg, float az_deg, double& zc, double& re, double& im, double& sphi) {
    const float th32 = el_deg * D2R_F;
    const float ph32 = az_deg * D2R_F;
    float st32, ct32;
    np_sincosf(th32, st32, ct32);
    const double st = (double)st32, ct = (double)ct32;
    double sd, cd;
    sincos_lean((double)ph32, sd, cd);
    zc = isnan(sd) ? sd : ct;
    re = st * cd;
    im = st * sd;
    sphi = st > 0.0 ? sd : (st < 0.0 ? -sd : 0.0);
}

// np.mod(x, 2pi): result takes the sign of the divisor
__device__ __forceinline__ double pymod_2pi(double x) {
    double m = fmod(x, TWO_PI);
    if (m != 0.0) { if (m < 0.0) m += TWO_PI; } else { m = 0.0; }
    return m;
}

// geometry.py:180-193
__device__ __forceinline__ bool in_fov(double th, double ph, double fh, double fv) {
    const double t = pymod_2pi(th), p = pymod_2pi(ph);
    const bool az = (p <= 0 + fh / 2) || (p >= TWO_PI - fh / 2);
    const bool el = (t <= HALF_PI + fv / 2) && (t >= HALF_PI - fv / 2);
    return az && el;
}

// ant_patterns.py:34-71 (NaN -> 0)
__device__ __forceinline__ double dipole_gain(double th) {
    const double s = sin(th);
    if (!(fabs(s) > 1e-10)) return 0.0;
    const double c = cos(HALF_PI * cos(th));
    return 1.643 * (c * c / s);
}

// the bits of a 64-lane ballot that belong to group `grp` of LPU lanes, shifted down to bit 0
template <int LPU>
__device__ __forceinline__ unsigned long long group_mask(unsigned long long b, int grp) {
    if constexpr (LPU == 64) return b;
    else return (b >> (grp * LPU)) & ((1ull << LPU) - 1ull);
}

__device__ __forceinline__ uint32_t float_order_key(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// LPU = lanes per user (32: two users share a wave; 64: one user per wave, any path count).
// LEAN = nothing needs the angles as numbers (no FoV, isotropic patterns, no angle / power side outputs - what
// compute_channels and bench.py run): the arccos / atan2 / FoV / dipole code is compiled out, which takes the kernel
// from 228 to far fewer registers, i.e. from 2 to 3-4 waves per SIMD on a kernel that waits on its loads and stores.
// ZROT (with LEAN): both rotations are exactly zero and the same for every user.
#ifndef K1_WAVES
#define K1_WAVES 4                                          // waves per workgroup (every wave works alone)
#endif
template <int LPU, bool LEAN, bool ZROT = false>
__global__ __launch_bounds__(64 * K1_WAVES, (LEAN ? 16 : 8) / K1_WAVES) void k1_path_prep(PrepArgs a) {
    constexpr int UPW = 64 / LPU;                           // users per wave
    const int lane = threadIdx.x & (LPU - 1);               // lane inside the user's group
    const int grp = (threadIdx.x & 63) / LPU;               // which group of the wave
    const int64_t u_raw = ((int64_t)blockIdx.x * K1_WAVES + (threadIdx.x >> 6)) * UPW + grp;
    const bool u_ok = u_raw < a.rays.n_ue;
    if (__ballot(u_ok) == 0ull) return;                     // whole wave past the end
    const int64_t u = u_ok ? u_raw : a.rays.n_ue - 1;       // idle group shadows the last user, writes nothing
    const dmx_rays& r = a.rays;
    const int L = r.n_paths;
    const size_t row = (size_t)u * (size_t)r.ld;
    const size_t srow = (size_t)u * (size_t)L;              // dense side-product rows
    const size_t wrow = (size_t)u * (size_t)a.P;

    double usx = a.usx, ucx = a.ucx, usy = a.usy, ucy = a.ucy, urz = a.urz;
    if (a.ue_rot_pu) {                                      // per-user rotation in degrees (dataset.py:329-338)
        const double rx = a.ue_rot_pu[3 * u + 0] * D2R_D, ry = a.ue_rot_pu[3 * u + 1] * D2R_D;
        urz = a.ue_rot_pu[3 * u + 2] * D2R_D;
        sincos(rx, &usx, &ucx);
        sincos(ry, &usy, &ucy);
    }
    const bool iso = (a.bs_pat == DMX_PATTERN_ISOTROPIC) && (a.ue_pat == DMX_PATTERN_ISOTROPIC);
    const float nan32 = __int_as_float(0x7fc00000);
    const double nan64 = (double)nan32;

    int keep_base = 0, count_paths = 0;
    bool has_fov_path = false;
    float first_inter = nan32;
    float maxd = -INFINITY;
    bool any_delay = false;

    for (int j0 = 0; j0 < L; j0 += LPU) {
        const int j = j0 + lane;
        const bool in = u_ok && j < L;
        // Eight loads issued TOGETHER, from an index every lane may read (idle lanes: the user's last path), masked
        // afterwards.  Written as `in ? array[row + j] : nan`, each load sat alone in a branch of its own with an
        // `s_waitcnt vmcnt(0)` behind it: eight memory round trips in a row were the 8 us a wave lived (SQ_WAVE_CYCLES /
        // SQ_WAVES, profiles/r3_d8_summary.txt) and 0.20-0.28 ms of stage 1 per 200k users.
        const size_t jc = row + (size_t)(j < L ? j : L - 1);
        const float power_r = r.power[jc], phase_r = r.phase[jc], delay_r = r.delay[jc], aoa_az_r = r.aoa_az[jc];
        const float aoa_el_r = r.aoa_el[jc], aod_az_r = r.aod_az[jc], aod_el_r = r.aod_el[jc], inter_r = r.inter[jc];
        const bool dop_rays = a.doppler && r.doppler_vel && r.doppler_acc;       // kernel-uniform
        float dvel_r = 0.f, dacc_r = 0.f;
        if (dop_rays) { dvel_r = r.doppler_vel[jc]; dacc_r = r.doppler_acc[jc]; }    // in the same batch
        const float power = in ? power_r : nan32;
        const float phase = in ? phase_r : nan32;
        const float delay = in ? delay_r : nan32;
        const float aoa_az = in ? aoa_az_r : nan32;
        const float aoa_el = in ? aoa_el_r : nan32;
        const float aod_az = in ? aod_az_r : nan32;
        const float aod_el = in ? aod_el_r : nan32;
        const float inter = in ? inter_r : nan32;

        double zc_t, re_t, im_t, zc_r, re_r, im_r, sphi_t = 0.0, sphi_r = 0.0;
        if constexpr (ZROT) {
            rotate_dir_zero(aod_el, aod_az, zc_t, re_t, im_t, sphi_t);
            rotate_dir_zero(aoa_el, aoa_az, zc_r, re_r, im_r, sphi_r);
        } else {
            rotate_dir<LEAN>(aod_el, aod_az, a.bsx, a.csx, a.bsy, a.csy, a.brz, zc_t, re_t, im_t);
            rotate_dir<LEAN>(aoa_el, aoa_az, usx, ucx, usy, ucy, urz, zc_r, re_r, im_r);
        }
        // arccos is NaN outside [-1, 1]; np.angle is NaN only for NaN input
        double th_t = (isnan(zc_t) || fabs(zc_t) > 1.0) ? nan64 : 0.0, ph_t = (isnan(re_t) || isnan(im_t)) ? nan64 : 0.0;
        double th_r = (isnan(zc_r) || fabs(zc_r) > 1.0) ? nan64 : 0.0, ph_r = (isnan(re_r) || isnan(im_r)) ? nan64 : 0.0;
        if constexpr (!LEAN) {
            if (a.need_angles) {                             // wave-uniform
                th_t = acos(zc_t); ph_t = atan2(im_t, re_t);
                th_r = acos(zc_r); ph_r = atan2(im_r, re_r);
            }
            if (in) {
                if (a.side.aod_el_rot) a.side.aod_el_rot[srow + j] = th_t;
                if (a.side.aod_az_rot) a.side.aod_az_rot[srow + j] = ph_t;
                if (a.side.aoa_el_rot) a.side.aoa_el_rot[srow + j] = th_r;
                if (a.side.aoa_az_rot) a.side.aoa_az_rot[srow + j] = ph_r;
            }
        }

        // field of view (dataset.py:493-511): outside -> angles become NaN
        bool mask = true;
        if (!LEAN && a.fov_enabled) {
            if (a.bs_restricted) mask = mask && in_fov(th_t, ph_t, a.bs_fh, a.bs_fv);
            if (a.ue_restricted) mask = mask && in_fov(th_r, ph_r, a.ue_fh, a.ue_fv);
            mask = mask && in;
            if (in && a.side.fov_mask) a.side.fov_mask[srow + j] = mask ? 1 : 0;
            if (!mask) { th_t = nan64; ph_t = nan64; th_r = nan64; ph_r = nan64; }
            const unsigned long long mb = group_mask<LPU>(__ballot(mask), grp);
            // first in-FoV path (dataset.py:594-598); the shuffle is executed by every lane, the
            // result is kept only by groups that had no in-FoV path yet
            const int src = mb != 0ull ? __ffsll((long long)mb) - 1 : 0;
            const float cand = __shfl(inter, src, LPU);
            if (!has_fov_path && mb != 0ull) { has_fov_path = true; first_inter = cand; }
        } else if (j0 == 0) {
            first_inter = __shfl(inter, 0, LPU);             // dataset.py:602
        }
        count_paths += __popcll(group_mask<LPU>(__ballot(in && !isnan(ph_r)), grp));   // dataset.py:616-619

        // powers (generator_utils.py:35, ant_patterns.py:167-168)
        const float p10 = power / 10.0f;
        const float pl = exp10f(p10);                        // float32 pow, as NumPy evaluates 10**float32
        double pw;
        if (LEAN || iso) {
            pw = (double)pl;
        } else {
            const double gt = a.bs_pat == DMX_PATTERN_HALFWAVE_DIPOLE ? dipole_gain(th_t) : 1.0;
            const double gr = a.ue_pat == DMX_PATTERN_HALFWAVE_DIPOLE ? dipole_gain(th_r) : 1.0;
            pw = (double)pl * (gt * gr);
        }
        if constexpr (!LEAN) {
            if (in) {
                if (a.side.power_linear) a.side.power_linear[srow + j] = pl;
                if (a.side.power_linear_ant_gain) a.side.power_linear_ant_gain[srow + j] = pw;
            }
        }

        // per-path record for the first P paths (dataset.py:258-261)
        const bool used = in && j < a.P;
        if (used && !isnan(delay)) { maxd = fmaxf(maxd, delay); any_delay = true; }
        const bool valid = used && !isnan(pw);               // channel.py:260
        const float ph32 = phase * D2R_F;                    // np.deg2rad(float32)
        float e_re, e_im;
        np_sincosf(ph32, e_im, e_re);                        // complex64 exp: NumPy's float32 cos / sin (and NaN-cheap, see there)
        const bool ang_ok = !isnan(th_t) && !isnan(th_r);    // geometry.py:65 zeroes NaN-zenith columns
        float c_re, c_im, dn = 0.0f;
        bool keep;
        if (a.freq_domain) {
            dn = delay / a.ts32;                             // float32 / float32 (channel.py:183)
            double pwc = pw;
            if (dn >= (float)a.n_sc) { pwc = 0.0; dn = (float)a.n_sc; }   // channel.py:187-189
            if (LEAN || iso) {
                const float amp = sqrtf((float)pwc / (float)a.n_sc);      // float32 (channel.py:192)
                c_re = amp * e_re; c_im = amp * e_im;
            } else {
                const double amp = sqrt(pwc / (double)a.n_sc);
                c_re = (float)(amp * (double)e_re); c_im = (float)(amp * (double)e_im);
            }
            if (dop_rays && !a.rx_filter) {                                      // construct_deepmimo.py:267-280
                const double v = in ? (double)dvel_r : 0.0;
                const double ac = in ? (double)dacc_r : 0.0;
                const double tau = (double)delay;
                const double arg = -TWO_PI * a.fc * (v * tau / LIGHTSPEED + ac * (tau * tau) / (2.0 * LIGHTSPEED));
                double sd, cd;
                sincos(arg, &sd, &cd);
                const float nr = (float)((double)c_re * cd - (double)c_im * sd);
                const float ni = (float)((double)c_re * sd + (double)c_im * cd);
                c_re = nr; c_im = ni;
            }
            // nansum (channel.py:283): a path with any NaN factor contributes nothing
            keep = valid && ang_ok && !isnan(ph_t) && !isnan(ph_r) && !isnan(c_re) && !isnan(c_im) && !isnan(dn) &&
                   (c_re != 0.0f || c_im != 0.0f);          // clipped / zero-gain paths add exactly 0
        } else {
            if (LEAN || iso) {
                const float amp = sqrtf((float)pw);                        // channel.py:286
                c_re = amp * e_re; c_im = amp * e_im;
            } else {
                const double amp = sqrt(pw);
                c_re = (float)(amp * (double)e_re); c_im = (float)(amp * (double)e_im);
            }
            if (!ang_ok) { c_re *= 0.0f; c_im *= 0.0f; }                   // zero array response, NaN stays NaN
            keep = valid;                                                  // slot even if coefficient is 0
        }
        double ty = 0.0, tz = 0.0, ry = 0.0, rz = 0.0;
        if (ang_ok && ZROT) {
            ty = a.bs_spacing * (sqrt(1.0 - zc_t * zc_t) * sphi_t); tz = a.bs_spacing * zc_t;
            ry = a.ue_spacing * (sqrt(1.0 - zc_r * zc_r) * sphi_r); rz = a.ue_spacing * zc_r;
        } else if (ang_ok) {                                 // geometry.py:99-101 in revolutions (kd / 2pi = spacing)
            const double rho_t = sqrt(re_t * re_t + im_t * im_t), rho_r = sqrt(re_r * re_r + im_r * im_r);
            ty = a.bs_spacing * (sqrt(1.0 - zc_t * zc_t) * (rho_t > 0.0 ? im_t / rho_t : 0.0)); tz = a.bs_spacing * zc_t;
            ry = a.ue_spacing * (sqrt(1.0 - zc_r * zc_r) * (rho_r > 0.0 ? im_r / rho_r : 0.0)); rz = a.ue_spacing * zc_r;
        }
        const unsigned long long kb = group_mask<LPU>(__ballot(keep), grp);
        int rank = 0;
        const bool sorted = a.sort_paths && L <= LPU;          // kernel-uniform; the whole user is in this one pass
        if (sorted) {
            // rank among the kept paths by |c|^2, ties by path index: 0 = strongest.  Lanes that keep nothing carry -1.
            const float key = keep ? c_re * c_re + c_im * c_im : -1.0f;
            for (int jj = 0; jj < L; ++jj) {
                const float kj = __shfl(key, jj, LPU);
                rank += (kj > key || (kj == key && jj < lane)) ? 1 : 0;
            }
        }
        if (keep) {
            const int slot = sorted ? rank : keep_base + __popcll(kb & ((1ull << lane) - 1ull));
            a.ws.c_re[wrow + slot] = c_re; a.ws.c_im[wrow + slot] = c_im; a.ws.dn[wrow + slot] = dn;
            a.ws.tx_y[wrow + slot] = ty; a.ws.tx_z[wrow + slot] = tz;
            a.ws.rx_y[wrow + slot] = ry; a.ws.rx_z[wrow + slot] = rz;
            a.ws.dop_v[wrow + slot] = dvel_r;                                    // 0 without Doppler; `keep` lanes are `in` lanes
            a.ws.dop_a[wrow + slot] = dacc_r;
        }
        keep_base += __popcll(kb);
    }

    // wave reductions
    for (int off = LPU / 2; off > 0; off >>= 1) maxd = fmaxf(maxd, __shfl_xor(maxd, off, LPU));
    const bool anyd = group_mask<LPU>(__ballot(any_delay), grp) != 0ull;
    if (lane == 0 && u_ok) {
        a.ws.n_keep[u] = keep_base;
        if (a.side.num_paths) a.side.num_paths[u] = count_paths;
        if (a.side.los) {
            const bool has = a.fov_enabled ? has_fov_path : (count_paths > 0);
            a.side.los[u] = has ? ((first_inter == 0.0f) ? 1 : 0) : -1;    // dataset.py:604-609
        }
        // one running maximum for the whole launch: a returning atomic per user would serialise 1e5 updates on
        // one L2 word (~88 per us), so look first (relaxed, L2-served) and only update when this user raises it
        if (a.side.max_delay_key && anyd) {
            const uint32_t key = float_order_key(maxd);
            if (key > __hip_atomic_load(a.side.max_delay_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(a.side.max_delay_key, key);
        }
    }
}

int launch_path_prep(const dmx_rays& rays, const dmx_params& prm, const WsView& ws, const dmx_side& side,
                     hipStream_t stream) {
    PrepArgs a;
    a.rays = rays; a.side = side; a.ws = ws;
    a.bsx = sin(prm.bs_rotation[0]); a.csx = cos(prm.bs_rotation[0]);
    a.bsy = sin(prm.bs_rotation[1]); a.csy = cos(prm.bs_rotation[1]);
    a.brz = prm.bs_rotation[2];
    a.usx = sin(prm.ue_rotation[0]); a.ucx = cos(prm.ue_rotation[0]);
    a.usy = sin(prm.ue_rotation[1]); a.ucy = cos(prm.ue_rotation[1]);
    a.urz = prm.ue_rotation[2];
    a.ue_rot_pu = prm.ue_rotation_per_user;
    a.fov_enabled = prm.fov_enabled; a.bs_restricted = prm.bs_fov_restricted; a.ue_restricted = prm.ue_fov_restricted;
    a.bs_fh = prm.bs_fov[0]; a.bs_fv = prm.bs_fov[1]; a.ue_fh = prm.ue_fov[0]; a.ue_fv = prm.ue_fov[1];
    a.bs_pat = prm.bs_pattern; a.ue_pat = prm.ue_pattern;
    a.bs_spacing = prm.bs_spacing; a.ue_spacing = prm.ue_spacing;
    a.P = ws.P; a.freq_domain = prm.freq_domain; a.n_sc = prm.n_subcarriers;
    a.ts32 = (float)(1.0 / prm.bandwidth);
    a.doppler = prm.enable_doppler; a.fc = prm.carrier_freq; a.rx_filter = prm.rx_filter && prm.freq_domain;
    // amplitude order only where something uses it: the opt-in adaptive precision of stage 2 (frequency domain; the
    // time-domain slots keep the path order, channel.py:285-287).  25 lane shuffles per user otherwise saved.
    a.sort_paths = (prm.freq_domain && (prm.flags & DMX_FLAG_ADAPTIVE_TERMS)) ? 1 : 0;
    a.need_angles = prm.fov_enabled || prm.bs_pattern != DMX_PATTERN_ISOTROPIC || prm.ue_pattern != DMX_PATTERN_ISOTROPIC ||
                    side.aod_el_rot || side.aod_az_rot || side.aoa_el_rot || side.aoa_az_rot;
    if (rays.n_ue == 0) return DMX_OK;
    const bool lean = !a.need_angles && !side.power_linear && !side.power_linear_ant_gain && !side.fov_mask;
    bool zrot = lean && !prm.ue_rotation_per_user;
    for (int i = 0; i < 3; ++i) zrot = zrot && prm.bs_rotation[i] == 0.0 && prm.ue_rotation[i] == 0.0;
    if (rays.n_paths <= 32) {
        const unsigned grid = (unsigned)((rays.n_ue + 2 * K1_WAVES - 1) / (2 * K1_WAVES));
        if (zrot) hipLaunchKernelGGL((k1_path_prep<32, true, true>), dim3(grid), dim3(64 * K1_WAVES), 0, stream, a);
        else if (lean) hipLaunchKernelGGL((k1_path_prep<32, true>), dim3(grid), dim3(64 * K1_WAVES), 0, stream, a);
        else hipLaunchKernelGGL((k1_path_prep<32, false>), dim3(grid), dim3(64 * K1_WAVES), 0, stream, a);
    } else {
        const unsigned grid = (unsigned)((rays.n_ue + K1_WAVES - 1) / K1_WAVES);
        if (lean) hipLaunchKernelGGL((k1_path_prep<64, true>), dim3(grid), dim3(64 * K1_WAVES), 0, stream, a);
        else hipLaunchKernelGGL((k1_path_prep<64, false>), dim3(grid), dim3(64 * K1_WAVES), 0, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k1_path_prep launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

}  // namespace dmx
