// rx_filter = 1 variant: per-path subcarrier gains with the receive low-pass (sinc) filter,
//   g[l, k] = sum_{d=0}^{N-1} c_l * sinc(d - dn_l) * D_l(d) * exp(-j 2pi d sc_k / N)
// (channel.py:166-168, 193-194; D_l(d) = Doppler phase at tap delay Ts*d,
// construct_deepmimo.py:275-280, 1 when Doppler is off).  The gains go to a caller-provided table
// [user_count, P, K] complex64 that the contraction kernel (k2_channel_fd.hip, GLOAD) reads
// instead of generating exp(-j 2pi dn k / N) itself.
//
// The sum over d is a length-N DFT of the tap sequence h_l[d] = c_l sinc(d - dn_l) D_l(d) sampled at the
// selected bins, so for power-of-two N (every DeepMIMO default: 64 ... 2048) it is computed as a radix-2
// FFT in LDS (k3_lpf_fft: one workgroup per user, N log N instead of N K operations per path - at
// N = K = 512 that is 57x less work and the variant costs about as much as the plain path).  Other N use
// the direct kernel k3_lpf_gains: one workgroup per (user, kept path), taps and the N roots of unity in
// LDS, thread k walks d with an exact integer phase index (d*sc_k mod N).
#include "k2_mfma_frag.h"
#include "dmx_tuning.h"
#include <stdlib.h>

namespace dmx {

struct LpfArgs {
    int64_t user_begin;
    int N, K;
    const int32_t* sc;
    float2* gtab;
    int doppler;
    double fc, ts;
    int pack;          // k3_lpf_fft_wave only: write {f16 hi (re, im), f16 lo (re, im)} of G scaled by the user's power of two
};

static constexpr double LPF_PI = 3.141592653589793;
static constexpr double LPF_C0 = 299792458.0;

__global__ __launch_bounds__(256) void k3_lpf_gains(WsView ws, LpfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* h = reinterpret_cast<float2*>(smem);          // [N] taps
    float2* w = h + a.N;                                   // [N] exp(-j 2pi m / N)
    const int l = blockIdx.x % ws.P;
    const int64_t ul = blockIdx.x / ws.P;                  // user index inside this call
    const int64_t u = a.user_begin + ul;
    if (l >= ws.n_keep[u]) return;                         // uniform for the workgroup
    const size_t rec = (size_t)u * ws.P + l;
    const double dn = (double)ws.dn[rec];
    const float cr = ws.c_re[rec], ci = ws.c_im[rec];
    const double v = (double)ws.dop_v[rec], ac = (double)ws.dop_a[rec];
    // np.sinc(d - dn) = sin(pi (d - dn)) / (pi (d - dn)) with sin(pi (d - dn)) = -(-1)^d sin(pi dn): one float64
    // sinpi per path instead of one sin per tap
    const float s0 = (float)sinpi(dn);
    for (int d = threadIdx.x; d < a.N; d += 256) {
        const double x = (double)d - dn;
        const float sn = x == 0.0 ? 1.0f : ((d & 1) ? s0 : -s0) / (float)(LPF_PI * x);
        float hr = cr * sn, hi = ci * sn;
        if (a.doppler) {
            const double tau = a.ts * (double)d;
            const double rev = -a.fc * (v * tau / LPF_C0 + ac * (tau * tau) / (2.0 * LPF_C0));
            float s, c;
            sincos_rev(frac_rev(rev), s, c);
            const float nr = hr * c - hi * s, ni = hr * s + hi * c;
            hr = nr; hi = ni;
        }
        h[d] = make_float2(hr, hi);
        float s, c;
        sincos_rev(frac_rev(-(double)d / (double)a.N), s, c);
        w[d] = make_float2(c, s);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < a.K; k += 256) {
        int step = a.sc[k] % a.N;
        if (step < 0) step += a.N;
        int idx = 0;
        float ar = 0.f, ai = 0.f;
        for (int d = 0; d < a.N; ++d) {
            const float2 hv = h[d], wv = w[idx];
            ar = fmaf(hv.x, wv.x, ar); ar = fmaf(-hv.y, wv.y, ar);
            ai = fmaf(hv.x, wv.y, ai); ai = fmaf(hv.y, wv.x, ai);
            idx += step;
            if (idx >= a.N) idx -= a.N;
        }
        a.gtab[((size_t)ul * ws.P + l) * a.K + k] = make_float2(ar, ai);
    }
}

// ---- FFT form -------------------------------------------------------------------------------------
// One workgroup per user.  PB paths are transformed together (PB*N/2 butterflies per stage over 256
// threads).  Decimation in time: taps are written to LDS in bit-reversed order, log2(N) in-place stages,
// twiddles from a table of the N/2 roots exp(-j 2pi m / N); the selected bins are then gathered out.
__global__ __launch_bounds__(256) void k3_lpf_fft(WsView ws, LpfArgs a, int log2n, int PB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* w = reinterpret_cast<float2*>(smem);           // [N/2] roots of unity
    float2* X = w + a.N / 2;                               // [PB][N]
    float* s0tab = reinterpret_cast<float*>(X + (size_t)PB * a.N);     // [PB] sin(pi dn_l) of the batch's paths
    const int tid = threadIdx.x;
    const int64_t ul = blockIdx.x, u = a.user_begin + ul;
    const int N = a.N, half_n = N >> 1;
    const int n_keep = ws.n_keep[u];
    for (int m = tid; m < half_n; m += 256) {
        float s, c;
        sincos_rev(frac_rev(-(double)m / (double)N), s, c);
        w[m] = make_float2(c, s);
    }
    for (int l0 = 0; l0 < n_keep; l0 += PB) {
        const int nb = (n_keep - l0) < PB ? (n_keep - l0) : PB;
        __syncthreads();                                    // previous batch fully gathered / roots written
        if (tid < nb) s0tab[tid] = (float)sinpi((double)ws.dn[(size_t)u * ws.P + l0 + tid]);
        __syncthreads();
        for (int i = tid; i < nb * N; i += 256) {           // taps, bit-reversed placement
            const int b = i / N, d = i - b * N;
            const size_t rec = (size_t)u * ws.P + l0 + b;
            const double x = (double)d - (double)ws.dn[rec];
            // np.sinc(x) with sin(pi (d - dn)) = -(-1)^d sin(pi dn): no per-tap sine
            const float sn = x == 0.0 ? 1.0f : ((d & 1) ? s0tab[b] : -s0tab[b]) / (float)(LPF_PI * x);
            float hr = ws.c_re[rec] * sn, hi = ws.c_im[rec] * sn;
            if (a.doppler) {
                const double tau = a.ts * (double)d;
                const double rev = -a.fc * ((double)ws.dop_v[rec] * tau / LPF_C0 + (double)ws.dop_a[rec] * (tau * tau) / (2.0 * LPF_C0));
                float s, c;
                sincos_rev(frac_rev(rev), s, c);
                const float nr = hr * c - hi * s, ni = hr * s + hi * c;
                hr = nr; hi = ni;
            }
            X[b * N + (int)(__brev((unsigned)d) >> (32 - log2n))] = make_float2(hr, hi);
        }
        for (int st = 1; st <= log2n; ++st) {
            __syncthreads();
            const int hs = 1 << (st - 1), tw_shift = log2n - st;
            for (int j = tid; j < nb * half_n; j += 256) {
                const int b = j / half_n, jj = j - b * half_n;
                const int pos = jj & (hs - 1), i0 = ((jj >> (st - 1)) << st) + pos, i1 = i0 + hs;
                const float2 tw = w[pos << tw_shift];
                float2* xb = X + b * N;
                const float2 p = xb[i0], q = xb[i1];
                const float qr = q.x * tw.x - q.y * tw.y, qi = q.x * tw.y + q.y * tw.x;
                xb[i0] = make_float2(p.x + qr, p.y + qi);
                xb[i1] = make_float2(p.x - qr, p.y - qi);
            }
        }
        __syncthreads();
        for (int i = tid; i < nb * a.K; i += 256) {         // gather the selected bins
            const int b = i / a.K, k = i - b * a.K;
            int bin = a.sc[k] & (N - 1);                    // N is a power of two: floor-mod for negative indices too
            a.gtab[((size_t)ul * ws.P + l0 + b) * a.K + k] = X[b * N + bin];
        }
    }
}

// ---- FFT form, one WAVE per path ----------------------------------------------------------------------
// k3_lpf_fft above spends its time in 9 radix-2 stages of workgroup barriers and LDS round trips (1.9 ms per 20k users at
// N = 512: as much as half the contraction).  Here one wave transforms one path on its own: Stockham autosort passes
// of radix 8 (then one of radix 4 or 2 when log2 N is not a multiple of 3), each butterfly held in registers, two LDS
// buffers per wave (ping-pong, natural order in and out - no bit reversal), twiddles from a table of the N roots of
// unity built once per persistent workgroup, no workgroup barrier after that.  N = 512: 3 passes instead of 9 stages.
// complex arithmetic on (re, im) pairs as 2-vectors: the compiler emits v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32
typedef float kv2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    const kv2 t = kv2{a.x, a.x} * kv2{b.x, b.y};
    const kv2 r = __builtin_elementwise_fma(kv2{-a.y, a.y}, kv2{b.y, b.x}, t);
    return make_float2(r[0], r[1]);
}
__device__ __forceinline__ float2 caddf(float2 a, float2 b) { const kv2 r = kv2{a.x, a.y} + kv2{b.x, b.y}; return make_float2(r[0], r[1]); }
__device__ __forceinline__ float2 csubf(float2 a, float2 b) { const kv2 r = kv2{a.x, a.y} - kv2{b.x, b.y}; return make_float2(r[0], r[1]); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }          // a * (-j)

template <int R> __device__ __forceinline__ void dft_small(float2 (&v)[R]);
template <> __device__ __forceinline__ void dft_small<2>(float2 (&v)[2]) {
    const float2 a = v[0], b = v[1];
    v[0] = caddf(a, b); v[1] = csubf(a, b);
}
template <> __device__ __forceinline__ void dft_small<4>(float2 (&v)[4]) {
    const float2 t0 = caddf(v[0], v[2]), t1 = csubf(v[0], v[2]), t2 = caddf(v[1], v[3]), t3 = mul_mi(csubf(v[1], v[3]));
    v[0] = caddf(t0, t2); v[1] = caddf(t1, t3); v[2] = csubf(t0, t2); v[3] = csubf(t1, t3);
}
template <> __device__ __forceinline__ void dft_small<8>(float2 (&v)[8]) {
    constexpr float H = 0.70710678118654752f;
    float2 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = caddf(v[k], v[k + 4]); b[k] = csubf(v[k], v[k + 4]); }
    b[1] = make_float2((b[1].x + b[1].y) * H, (b[1].y - b[1].x) * H);          // * e^{-j pi/4}
    b[2] = mul_mi(b[2]);                                                       // * e^{-j pi/2}
    b[3] = make_float2((b[3].y - b[3].x) * H, -(b[3].x + b[3].y) * H);         // * e^{-j 3pi/4}
    dft_small<4>(a);
    dft_small<4>(b);
#pragma unroll
    for (int m = 0; m < 4; ++m) { v[2 * m] = a[m]; v[2 * m + 1] = b[m]; }
}

// Packed-fp32 forms the compiler does not select (it keeps a swizzled copy of every twiddle - 56 registers instead of 28 -
// negates with v_xor and moves halves around for the multiplications by -j: 3 instructions per complex product, 37 per
// radix-8 butterfly).  op_sel / op_sel_hi pick the 32-bit half that feeds the low / high result, neg_lo / neg_hi negate it.
// None of these reads a transcendental's result directly (taps and Doppler phasors go through compiler-selected
// multiplies first), and there are no matrix-core instructions in this file.
__device__ __forceinline__ float2 cmul_pk(float2 a, float2 b) {              // a * b: 2 instructions, b as it is
    kv2 r;
    const kv2 av = {a.x, a.y}, bv = {b.x, b.y};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(r) : "v"(av), "v"(bv));
    return make_float2(r[0], r[1]);
}
__device__ __forceinline__ float2 cadd_mj(float2 a, float2 b) {              // a + (-j) b = (a.x + b.y, a.y - b.x)
    kv2 r;
    const kv2 av = {a.x, a.y}, bv = {b.x, b.y};
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return make_float2(r[0], r[1]);
}
__device__ __forceinline__ float2 csub_mj(float2 a, float2 b) {              // a - (-j) b = (a.x - b.y, a.y + b.x)
    kv2 r;
    const kv2 av = {a.x, a.y}, bv = {b.x, b.y};
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(av), "v"(bv));
    return make_float2(r[0], r[1]);
}
__device__ __forceinline__ float2 rot1_pk(float2 b) {                        // sqrt(2) e^{-j pi/4} b = (x + y, y - x)
    kv2 r;
    const kv2 bv = {b.x, b.y};
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(bv));
    return make_float2(r[0], r[1]);
}
__device__ __forceinline__ float2 rot3_pk(float2 b) {                        // sqrt(2) e^{-j 3pi/4} b = (y - x, -x - y)
    kv2 r;
    const kv2 bv = {b.x, b.y};
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(r) : "v"(bv));
    return make_float2(r[0], r[1]);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { const kv2 r = kv2{a.x, a.y} * s; return make_float2(r[0], r[1]); }
// radix-8 butterfly, 28 packed instructions (dft_small<8> with the -j factors folded into the additions)
__device__ __forceinline__ void dft8_pk(float2 (&v)[8]) {
    constexpr float H = 0.70710678118654752f;
    float2 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = caddf(v[k], v[k + 4]); b[k] = csubf(v[k], v[k + 4]); }
    const float2 b1 = cscale(rot1_pk(b[1]), H), b3 = cscale(rot3_pk(b[3]), H);
    const float2 t0 = caddf(a[0], a[2]), t1 = csubf(a[0], a[2]), t2 = caddf(a[1], a[3]), d = csubf(a[1], a[3]);
    v[0] = caddf(t0, t2); v[2] = cadd_mj(t1, d); v[4] = csubf(t0, t2); v[6] = csub_mj(t1, d);
    const float2 u0 = cadd_mj(b[0], b[2]), u1 = csub_mj(b[0], b[2]), u2 = caddf(b1, b3), e = csubf(b1, b3);
    v[1] = caddf(u0, u2); v[3] = cadd_mj(u1, e); v[5] = csubf(u0, u2); v[7] = csub_mj(u1, e);
}

// element e of a transform buffer lives at e + e/16: the pad makes the stride-8 writes of the first pass (lane l writes
// elements 8l + r: two banks for the whole wave without it) conflict-free and costs the other passes at most 2-way
__device__ __forceinline__ int fpad(int e) { return e + (e >> 4); }

// one Stockham pass of radix R over N points: src -> dst, Ns = product of the radices already done
template <int R>
__device__ __forceinline__ void fft_pass(const float2* src, float2* dst, const float2* W, int N, int log2n, int Ns, int log2ns, int lane) {
    const int nb = N / R;
    constexpr int LR = R == 8 ? 3 : (R == 4 ? 2 : 1);
    const int tw_shift = log2n - log2ns - LR;                       // W_{Ns R}^{m} = W_N^{m << tw_shift}
    for (int j = lane; j < nb; j += 64) {
        const int k = j & (Ns - 1);
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = src[fpad(j + r * nb)];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul_pk(v[r], W[(k * r) << tw_shift]);
        }
        if constexpr (R == 8) dft8_pk(v);
        else dft_small<R>(v);
        const int j0 = ((j - k) << LR) + k;
#pragma unroll
        for (int r = 0; r < R; ++r) dst[fpad(j0 + r * Ns)] = v[r];
    }
}

__host__ __device__ inline size_t lpf_buf_elems(int N) { return (size_t)N + N / 16 + 1; }
__host__ __device__ inline size_t lpf_wave_lds_bytes(int N) { return (size_t)N * 8 + 4 * 2 * lpf_buf_elems(N) * 8; }

#ifndef K3_WAVES_PER_SIMD
#define K3_WAVES_PER_SIMD 4
#endif
__global__ __launch_bounds__(256, K3_WAVES_PER_SIMD) void k3_lpf_fft_wave(WsView ws, LpfArgs a, int log2n, int64_t user_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = a.N;
    float2* W = reinterpret_cast<float2*>(smem);                                 // [N] exp(-j 2pi m / N)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* bufA = W + N + (size_t)wave * 2 * lpf_buf_elems(N);                  // this wave's two transform buffers
    float2* bufB = bufA + lpf_buf_elems(N);
    for (int m = tid; m < N; m += 256) {
        float s, c;
        sincos_rev(frac_rev(-(double)m / (double)N), s, c);
        W[m] = make_float2(c, s);
    }
    __syncthreads();
    // the selected bins are the same for every user and path: up to 8 per lane stay in registers (K <= 512)
    constexpr int NBIN = 8;
    int binreg[NBIN];
#pragma unroll
    for (int j = 0; j < NBIN; ++j) { const int k = lane + 64 * j; binreg[j] = k < a.K ? fpad(a.sc[k] & (N - 1)) : 0; }
    const bool bins_in_regs = a.K <= 64 * NBIN;
    for (int64_t ul = blockIdx.x; ul < user_count; ul += gridDim.x) {
        const int64_t u = a.user_begin + ul;
        const int n_keep = ws.n_keep[u];
        // path records of this user, one per lane, read ONCE (a load per path in front of its transform left the wave
        // waiting ~2 us of memory latency per path); sin(pi dn) for all paths in parallel
        const size_t rb = (size_t)u * ws.P;
        const bool lok = lane < n_keep;
        const float dn_l = lok ? ws.dn[rb + lane] : 0.f, cr_l = lok ? ws.c_re[rb + lane] : 0.f, ci_l = lok ? ws.c_im[rb + lane] : 0.f;
        const float dv_l = (lok && a.doppler) ? ws.dop_v[rb + lane] : 0.f, da_l = (lok && a.doppler) ? ws.dop_a[rb + lane] : 0.f;
        const float s0_l = (float)sinpi((double)dn_l);
        float gsl = 1.0f;
        if (a.pack) {
            // the operand scale k2_fd_mfma's stage_item derives for this user: max |c| component -> [512, 1024), two
            // more bits of headroom for the sinc sum
            float m = lane < (n_keep < 32 ? n_keep : 32) ? fmaxf(fabsf(cr_l), fabsf(ci_l)) : 0.f;
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            int e;
            (void)frexpf(m, &e);
            gsl = ldexpf(1.0f, 8 - e);
        }
        for (int l = wave; l < n_keep; l += 4) {
            float dnf, cr, ci, s0, dvf, daf;
            if (l < 64) {                                                        // wave-uniform
                dnf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dn_l), l));
                cr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cr_l), l));
                ci = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ci_l), l));
                s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s0_l), l));
                dvf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv_l), l));
                daf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, da_l), l));
            } else {                                                             // beyond 64 kept paths: read per path
                dnf = ws.dn[rb + l]; cr = ws.c_re[rb + l]; ci = ws.c_im[rb + l];
                s0 = (float)sinpi((double)dnf);
                dvf = a.doppler ? ws.dop_v[rb + l] : 0.f; daf = a.doppler ? ws.dop_a[rb + l] : 0.f;
            }
            const double dv = (double)dvf, da = (double)daf;
            // np.sinc(d - dn) = sin(pi (d - dn)) / (pi (d - dn)) with sin(pi (d - dn)) = -(-1)^d sin(pi dn)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");               // the previous path's gather has read bufA/bufB
            __builtin_amdgcn_wave_barrier();
            for (int d = lane; d < N; d += 64) {
                // d - dn in float32: a multiple of ulp(dn) below 2^12, exact unless dn is tiny (then 6e-8 relative)
                const float x = (float)d - dnf;
                const float sn = x == 0.0f ? 1.0f : ((d & 1) ? s0 : -s0) * __builtin_amdgcn_rcpf((float)LPF_PI * x);
                float hr = cr * sn, hi = ci * sn;
                if (a.doppler) {
                    const double tau = a.ts * (double)d;
                    const double rev = -a.fc * (dv * tau / LPF_C0 + da * (tau * tau) / (2.0 * LPF_C0));
                    float s, c;
                    sincos_rev(frac_rev(rev), s, c);
                    const float nr = hr * c - hi * s, ni = hr * s + hi * c;
                    hr = nr; hi = ni;
                }
                bufA[fpad(d)] = make_float2(hr, hi);
            }
            float2* src = bufA;
            float2* dst = bufB;
            int log2ns = 0;
            while (log2ns < log2n) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                const int left = log2n - log2ns;
                if (left >= 3) { fft_pass<8>(src, dst, W, N, log2n, 1 << log2ns, log2ns, lane); log2ns += 3; }
                else if (left == 2) { fft_pass<4>(src, dst, W, N, log2n, 1 << log2ns, log2ns, lane); log2ns += 2; }
                else { fft_pass<2>(src, dst, W, N, log2n, 1 << log2ns, log2ns, lane); log2ns += 1; }
                float2* t = src; src = dst; dst = t;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            float2* grow = a.gtab + ((size_t)ul * ws.P + l) * a.K;
            uint2* prow = reinterpret_cast<uint2*>(grow);
            if (bins_in_regs) {
#pragma unroll
                for (int j = 0; j < NBIN; ++j) {
                    const int k = lane + 64 * j;
                    if (k < a.K) {
                        const float2 g = src[binreg[j]];
                        if (a.pack) {
                            h2 hi, lo;
                            split2_f16(g.x * gsl, g.y * gsl, hi, lo, ws.neg_one);
                            prow[k] = make_uint2(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo));
                        } else {
                            grow[k] = g;
                        }
                    }
                }
            } else {
                for (int k = lane; k < a.K; k += 64) {
                    const float2 g = src[fpad(a.sc[k] & (N - 1))];                // N is a power of two: floor-mod
                    if (a.pack) {
                        h2 hi, lo;
                        split2_f16(g.x * gsl, g.y * gsl, hi, lo, ws.neg_one);
                        prow[k] = make_uint2(__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo));
                    } else {
                        grow[k] = g;
                    }
                }
            }
        }
    }
}

// ---- N = 512 (DeepMIMO's default OFDM size), one WAVE per USER -----------------------------------------------------
// k3_lpf_fft_wave is generic in N and pays for it: 656 vector instructions per transform (profiles/r2_lpf_summary.txt),
// which is its time (VALU issue 3.2 of 4.3 ms at the headline shape).  With N fixed at 512 = 8^3 and a lane holding the
// eight points d = lane + 64 r of its transform:
//   * the taps a lane generates are exactly the eight inputs of ITS butterfly of the first Stockham pass - no LDS round
//     trip for the taps, and that pass has no twiddles;
//   * the twiddles of passes 2 and 3 depend on the lane only (exp(-j 2pi (lane & 7) r / 64), exp(-j 2pi lane r / 512)):
//     28 registers filled once per wave - no table, no lookups, no address arithmetic;
//   * a wave's reads of a pass are all issued before its writes and LDS serves a wave in order: ONE buffer per wave,
//     transformed in place;
//   * the last pass leaves bins lane + 64 r in the lane - when the selected subcarriers are 0 .. K-1 (IDENT) they are
//     scaled, split and stored from registers; other selections go through the buffer once more;
//   * sin(pi (d - dn)) = -(-1)^d sin(pi dn) and (-1)^d = (-1)^lane for every point of the lane: tap = (c * S) / (d - dn)
//     with one per-lane S per path, i.e. subtract, reciprocal, one packed multiply per tap;
//   * a wave owns a USER (all its paths in turn): no workgroup barrier, no 7 / 6 / 6 / 6 imbalance of 25 paths over 4 waves.
//   * Doppler (float64 phases per tap) is its own instantiation: its registers would halve the others' occupancy.
template <bool PACK>
__device__ __forceinline__ void store_gain(float2* grow, uint2* prow, int k, float2 g, float m1) {
    if constexpr (PACK) {
        h2 hi, lo;
        split2_f16(g.x, g.y, hi, lo, m1);
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        const u2v w = {__builtin_bit_cast(unsigned, hi), __builtin_bit_cast(unsigned, lo)};
        __builtin_nontemporal_store(w, reinterpret_cast<u2v*>(prow + k));     // read back once, by another CU
    } else {
        grow[k] = g;
    }
}
template <bool IDENT, bool PACK, bool DOPPLER>
__global__ __launch_bounds__(256) void k3_lpf_fft512(WsView ws, LpfArgs a, int64_t user_count) {
    constexpr int N = 512;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* X = reinterpret_cast<float2*>(smem) + (size_t)wave * lpf_buf_elems(N);
    float2 tw2[8], tw3[8];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
        float s, c;
        sincos_rev(frac_rev(-(double)((lane & 7) * r) / 64.0), s, c);
        tw2[r] = make_float2(c, s);
        sincos_rev(frac_rev(-(double)(lane * r) / 512.0), s, c);
        tw3[r] = make_float2(c, s);
    }
    // LDS positions of this lane: pass 1 writes elements 8 lane + r, reads of passes 2 / 3 take lane + 64 r, pass 2
    // writes 64 (lane >> 3) + (lane & 7) + 8 r (Stockham autosort, see fft_pass)
    float2* w1 = X + fpad(8 * lane);                       // + r: same pad for all eight
    float2* rd = X + fpad(lane);                           // + 68 r
    float2* w2 = X + fpad(64 * (lane >> 3) + (lane & 7));  // + 8 r + (r >> 1): 64 g + k + 8 r has pad 4 g + (k + 8 r) / 16, k < 8
    const float lanef = (float)lane;
    const unsigned sgn = (lane & 1) ? 0u : 0x80000000u;    // S = (-1)^d * (-1) * s0: odd d -> +s0, even d -> -s0
    int binreg[8];
    if constexpr (!IDENT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = lane + 64 * j; binreg[j] = k < a.K ? fpad(a.sc[k] & (N - 1)) : 0; }
    }
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t ul = (int64_t)blockIdx.x * 4 + wave; ul < user_count; ul += nwaves) {
        const int64_t u = a.user_begin + ul;
        const int n_keep = ws.n_keep[u];
        const size_t rb = (size_t)u * ws.P;
        const bool lok = lane < n_keep;
        const float dn_l = lok ? ws.dn[rb + lane] : 0.f;
        float cr_l = lok ? ws.c_re[rb + lane] : 0.f, ci_l = lok ? ws.c_im[rb + lane] : 0.f;
        const float dv_l = (lok && DOPPLER) ? ws.dop_v[rb + lane] : 0.f, da_l = (lok && DOPPLER) ? ws.dop_a[rb + lane] : 0.f;
        // sin(pi dn) / pi per path, all paths in parallel (float64 sinpi: dn reaches thousands of samples)
        const float s0_l = (float)(sinpi((double)dn_l) * (1.0 / LPF_PI));
        if constexpr (PACK) {
            // the operand scale k2_fd_mfma's stage_item derives for this user: max |c| component -> [512, 1024), two
            // more bits of headroom for the sinc sum; a power of two, so scaling c instead of G changes nothing
            float m = lane < (n_keep < 32 ? n_keep : 32) ? fmaxf(fabsf(cr_l), fabsf(ci_l)) : 0.f;
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            int e;
            (void)frexpf(m, &e);
            const float gsl = ldexpf(1.0f, 8 - e);
            cr_l *= gsl; ci_l *= gsl;
        }
        for (int l = 0; l < n_keep; ++l) {
            const float dnf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dn_l), l));
            const float cr = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cr_l), l));
            const float ci = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ci_l), l));
            const unsigned s0b = (unsigned)__builtin_amdgcn_readlane(__builtin_bit_cast(int, s0_l), l);
            const float S = __builtin_bit_cast(float, s0b ^ sgn);
            const kv2 cs = kv2{cr * S, ci * S};
            const float x0 = lanef - dnf;                  // d - dn in float32: exact unless dn is tiny (then 6e-8 relative)
            float2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const kv2 t = cs * __builtin_amdgcn_rcpf(x0 + (float)(64 * r));      // lane + 64 r is exact in float32, so is the sum
                v[r] = make_float2(t[0], t[1]);
            }
            if (dnf == rintf(dnf) && dnf >= 0.f && dnf < (float)N) {                 // wave-uniform: np.sinc(0) = 1; s0 = 0, the other taps are 0
                const int d0 = (int)dnf;
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = (lane + 64 * r == d0) ? make_float2(cr, ci) : make_float2(0.f, 0.f);
            }
            if constexpr (DOPPLER) {
                const double dv = (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dv_l), l));
                const double da = (double)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, da_l), l));
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const double tau = a.ts * (double)(lane + 64 * r);
                    const double rev = -a.fc * (dv * tau / LPF_C0 + da * (tau * tau) / (2.0 * LPF_C0));
                    float s, c;
                    sincos_rev(frac_rev(rev), s, c);
                    v[r] = cmulf(v[r], make_float2(c, s));
                }
            }
            // pass 1 (Ns = 1): no twiddles; element 8 lane + r
            dft8_pk(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) w1[r] = v[r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // pass 2 (Ns = 8)
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = rd[68 * r];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul_pk(v[r], tw2[r]);
            dft8_pk(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) w2[8 * r + (r >> 1)] = v[r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // pass 3 (Ns = 64): bins lane + 64 r
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = rd[68 * r];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul_pk(v[r], tw3[r]);
            dft8_pk(v);
            float2* grow = a.gtab + ((size_t)ul * ws.P + l) * a.K;
            uint2* prow = reinterpret_cast<uint2*>(grow);
            if constexpr (!IDENT) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");           // every lane has read its pass-3 inputs
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 8; ++r) rd[68 * r] = v[r];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = X[binreg[r]];
            }
            if (a.K == N) {                                                      // wave-uniform: no per-store guard
#pragma unroll
                for (int r = 0; r < 8; ++r) store_gain<PACK>(grow, prow, lane + 64 * r, v[r], ws.neg_one);
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (lane + 64 * r < a.K) store_gain<PACK>(grow, prow, lane + 64 * r, v[r], ws.neg_one);
            }
            if constexpr (!IDENT) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");           // the gather has read X before the next path's pass 1
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// ---- the same scheme for N = 64, 128, 256 and 1024 ------------------------------------------------------------------
// LP = min(64, N / 8) lanes transform one path (64 / LP paths per wave for N < 512), a lane holds the E = N / LP points
// j + LP m (j = lane % LP) - taps in, bins out.  Every pass reads and writes that pattern: passes 1 and 2 are radix 8
// (E / 8 butterflies per lane, butterfly t on points t + r E/8; pass 2's twiddles exp(-j 2pi (j & 7) r / 64)), pass 3
// has radix N / 64 (2, 4: 64 / LP butterflies per lane; 16 for N = 1024: one) with twiddles exp(-j 2pi (j + LP t) r / N).
// N = 64 ends after pass 2.  N = 512 keeps its own kernel above (one butterfly per lane and pass, nothing indexed).
__device__ __forceinline__ void dft16_pk(float2 (&x)[16]) {                  // x[r] -> X[r], 4 x 4
    constexpr float H = 0.70710678118654752f, C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;
    float2 u[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {                                            // DFT-4 over c of x[a + 4 c] -> u[a][q]
        float2 t[4] = {x[a], x[a + 4], x[a + 8], x[a + 12]};
        const float2 t0 = caddf(t[0], t[2]), t1 = csubf(t[0], t[2]), t2 = caddf(t[1], t[3]), d = csubf(t[1], t[3]);
        u[a][0] = caddf(t0, t2); u[a][1] = cadd_mj(t1, d); u[a][2] = csubf(t0, t2); u[a][3] = csub_mj(t1, d);
    }
    // twiddles W16^(a q): W16 = (C1, -S1), W16^2 = H (1, -1), W16^3 = (S1, -C1), W16^4 = -j, W16^6 = H (-1, -1), W16^9 = -W16
    u[1][1] = cmul_pk(u[1][1], make_float2(C1, -S1));
    u[1][2] = cscale(rot1_pk(u[1][2]), H);
    u[1][3] = cmul_pk(u[1][3], make_float2(S1, -C1));
    u[2][1] = cscale(rot1_pk(u[2][1]), H);
    u[2][2] = mul_mi(u[2][2]);
    u[2][3] = cscale(rot3_pk(u[2][3]), H);
    u[3][1] = cmul_pk(u[3][1], make_float2(S1, -C1));
    u[3][2] = cscale(rot3_pk(u[3][2]), H);
    u[3][3] = cmul_pk(u[3][3], make_float2(-C1, S1));
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                            // DFT-4 over a of u[a][q] -> X[q + 4 p]
        const float2 t0 = caddf(u[0][q], u[2][q]), t1 = csubf(u[0][q], u[2][q]), t2 = caddf(u[1][q], u[3][q]), d = csubf(u[1][q], u[3][q]);
        x[q] = caddf(t0, t2); x[q + 4] = cadd_mj(t1, d); x[q + 8] = csubf(t0, t2); x[q + 12] = csub_mj(t1, d);
    }
}

template <int LOG2N, bool IDENT, bool PACK, bool DOPPLER>
__global__ __launch_bounds__(256) void k3_lpf_fft_pow2(WsView ws, LpfArgs a, int64_t user_count) {
    constexpr int N = 1 << LOG2N;
    constexpr int LP = N >= 512 ? 64 : N / 8;              // lanes per path
    constexpr int PW = 64 / LP;                            // paths per wave
    constexpr int E = N / LP;                              // points per lane (8; 16 for N = 1024)
    constexpr int S1 = E / 8;                              // radix-8 butterflies per lane in passes 1 and 2
    constexpr int R3 = N / 64;                             // radix of pass 3 (1 = no pass 3)
    constexpr int S3 = R3 > 1 ? E / R3 : 1;                // its butterflies per lane
    static_assert(LOG2N >= 6 && LOG2N <= 10 && LOG2N != 9, "N = 64, 128, 256, 1024");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane / LP, j = lane % LP;
    float2* X = reinterpret_cast<float2*>(smem) + (size_t)(wave * PW + g) * lpf_buf_elems(N);
    float2 tw2[8];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
        float s, c;
        sincos_rev(frac_rev(-(double)((j & 7) * r) / 64.0), s, c);
        tw2[r] = make_float2(c, s);
    }
    // pass 3: exp(-j 2pi (j + LP t) r / N).  N = 1024 (one butterfly of 16): the powers 1..3 and 4, 8, 12 of w = W^j,
    // w^(4a+b) = w^(4a) w^b.  Otherwise S3 x (R3 - 1) values.
    constexpr int NT3 = R3 == 16 ? 6 : (R3 > 1 ? S3 * (R3 - 1) : 1);
    float2 tw3[NT3];
    if constexpr (R3 == 16) {
        const int pw[6] = {1, 2, 3, 4, 8, 12};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float s, c;
            sincos_rev(frac_rev(-(double)(j * pw[i]) / (double)N), s, c);
            tw3[i] = make_float2(c, s);
        }
    } else if constexpr (R3 > 1) {
#pragma unroll
        for (int t = 0; t < S3; ++t)
#pragma unroll
            for (int r = 1; r < R3; ++r) {
                float s, c;
                sincos_rev(frac_rev(-(double)((j + LP * t) * r) / (double)N), s, c);
                tw3[t * (R3 - 1) + r - 1] = make_float2(c, s);
            }
    }
    const float jf = (float)j;
    const unsigned sgn = (j & 1) ? 0u : 0x80000000u;       // LP is even: (-1)^d = (-1)^j for every point of the lane
    int binreg[E];
    if constexpr (!IDENT) {
#pragma unroll
        for (int m = 0; m < E; ++m) { const int k = j + LP * m; binreg[m] = k < a.K ? fpad(a.sc[k] & (N - 1)) : 0; }
    }
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t ul = (int64_t)blockIdx.x * 4 + wave; ul < user_count; ul += nwaves) {
        const int64_t u = a.user_begin + ul;
        const int n_keep = ws.n_keep[u];
        const size_t rb = (size_t)u * ws.P;
        const bool lok = lane < n_keep;
        const float dn_l = lok ? ws.dn[rb + lane] : 0.f;
        float cr_l = lok ? ws.c_re[rb + lane] : 0.f, ci_l = lok ? ws.c_im[rb + lane] : 0.f;
        const float dv_l = (lok && DOPPLER) ? ws.dop_v[rb + lane] : 0.f, da_l = (lok && DOPPLER) ? ws.dop_a[rb + lane] : 0.f;
        const float s0_l = (float)(sinpi((double)dn_l) * (1.0 / LPF_PI));
        if constexpr (PACK) {
            float m = lane < (n_keep < 32 ? n_keep : 32) ? fmaxf(fabsf(cr_l), fabsf(ci_l)) : 0.f;
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            int e;
            (void)frexpf(m, &e);
            const float gsl = ldexpf(1.0f, 8 - e);
            cr_l *= gsl; ci_l *= gsl;
        }
        for (int l0 = 0; l0 < n_keep; l0 += PW) {
            const int l = l0 + g;                          // this lane group's path; groups past n_keep compute zeros and store nothing
            const bool act = l < n_keep;
            const int ls = act ? l : 0;
            // every shuffle outside any `act ? :` - inside, only the active groups' lanes execute it and a source lane in an
            // idle group (the last path of a user: slot l0 + g >= the active lanes) returns 0
            const float dnf = __shfl(dn_l, ls), crs = __shfl(cr_l, ls), cis = __shfl(ci_l, ls);
            const float cr = act ? crs : 0.f, ci = act ? cis : 0.f;
            const float S = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, __shfl(s0_l, ls)) ^ sgn);
            const kv2 cs = kv2{cr * S, ci * S};
            const float x0 = jf - dnf;
            float2 v[E];
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const kv2 t = cs * __builtin_amdgcn_rcpf(x0 + (float)(LP * m));
                v[m] = make_float2(t[0], t[1]);
            }
            const bool whole = dnf == rintf(dnf) && dnf >= 0.f && dnf < (float)N;    // np.sinc(0) = 1: s0 = 0, the other taps 0
            if (__builtin_amdgcn_ballot_w64(whole) != 0) {
                const int d0 = (int)dnf;
#pragma unroll
                for (int m = 0; m < E; ++m)
                    if (whole) v[m] = (j + LP * m == d0) ? make_float2(cr, ci) : make_float2(0.f, 0.f);
            }
            if constexpr (DOPPLER) {
                const double dv = (double)__shfl(dv_l, ls), da = (double)__shfl(da_l, ls);
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const double tau = a.ts * (double)(j + LP * m);
                    const double rev = -a.fc * (dv * tau / LPF_C0 + da * (tau * tau) / (2.0 * LPF_C0));
                    float s, c;
                    sincos_rev(frac_rev(rev), s, c);
                    v[m] = cmulf(v[m], make_float2(c, s));
                }
            }
            // pass 1 (radix 8, no twiddles): butterfly b = j + LP t on points t + r S1 -> elements 8 b + r
#pragma unroll
            for (int t = 0; t < S1; ++t) {
                float2 w[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) w[r] = v[t + r * S1];
                dft8_pk(w);
                float2* dst = X + fpad(8 * (j + LP * t));
#pragma unroll
                for (int r = 0; r < 8; ++r) dst[r] = w[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // pass 2 (radix 8, Ns = 8): elements b + r N/8 = j + LP (t + r S1) -> 64 (b >> 3) + (b & 7) + 8 r
#pragma unroll
            for (int m = 0; m < E; ++m) v[m] = X[fpad(j + LP * m)];
#pragma unroll
            for (int t = 0; t < S1; ++t) {
                float2 w[8];
                w[0] = v[t];
#pragma unroll
                for (int r = 1; r < 8; ++r) w[r] = cmul_pk(v[t + r * S1], tw2[r]);
                dft8_pk(w);
                const int b = j + LP * t;
                float2* dst = X + fpad(64 * (b >> 3) + (b & 7));
#pragma unroll
                for (int r = 0; r < 8; ++r) dst[8 * r + (r >> 1)] = w[r];
            }
            if constexpr (R3 > 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                // pass 3 (radix R3, Ns = 64): elements b + 64 r = j + LP (t + r S3) -> bins b + 64 r, the same pattern
#pragma unroll
                for (int m = 0; m < E; ++m) v[m] = X[fpad(j + LP * m)];
                if constexpr (R3 == 16) {
                    float2 w[16];
                    w[0] = v[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) {
                        const float2 lo = tw3[(r & 3) - 1 < 0 ? 0 : (r & 3) - 1], hi = tw3[2 + (r >> 2)];
                        float2 tw = (r & 3) == 0 ? hi : ((r >> 2) == 0 ? lo : cmul_pk(hi, lo));
                        w[r] = cmul_pk(v[r], tw);
                    }
                    dft16_pk(w);
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = w[r];
                } else {
#pragma unroll
                    for (int t = 0; t < S3; ++t) {
                        float2 w[R3];
                        w[0] = v[t];
#pragma unroll
                        for (int r = 1; r < R3; ++r) w[r] = cmul_pk(v[t + r * S3], tw3[t * (R3 - 1) + r - 1]);
                        dft_small<R3>(w);
#pragma unroll
                        for (int r = 0; r < R3; ++r) v[t + r * S3] = w[r];
                    }
                }
            } else {
                // N = 64: pass 2 already wrote bins j + 8 r in order; fetch them back into the lane pattern
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < E; ++m) v[m] = X[fpad(j + LP * m)];
            }
            float2* grow = a.gtab + ((size_t)ul * ws.P + ls) * a.K;
            uint2* prow = reinterpret_cast<uint2*>(grow);
            if constexpr (!IDENT) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");           // every lane has read its inputs of the last pass
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < E; ++m) X[fpad(j + LP * m)] = v[m];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < E; ++m) v[m] = X[binreg[m]];
            }
#pragma unroll
            for (int m = 0; m < E; ++m)
                if (act && j + LP * m < a.K) store_gain<PACK>(grow, prow, j + LP * m, v[m], ws.neg_one);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");               // the buffer is free for the next path
            __builtin_amdgcn_wave_barrier();
        }
    }
}

int launch_channels_fd_lpf_contract(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                    const float2* gtab, float2* out, hipStream_t stream, bool packed);
bool lpf_table_packed(const dmx_params& prm, const WsView& ws);

static int launch_channels_fd_lpf_once(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                       float2* gtab, float2* out, hipStream_t stream);

// Users go through in chunks that share ONE table buffer: gains written by the FFT kernel for a chunk are read back by
// the contraction a few tens of microseconds later, while they still sit in the 256-MiB Infinity Cache, instead of
// making a 2 x 100 KB-per-user round trip through HBM against the output stream (chunk users x P x K x 8 bytes of table;
// the caller's workspace is sized for the whole call, the first chunk's worth of it is used).
int launch_channels_fd_lpf(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                           float2* gtab, float2* out, hipStream_t stream) {
    if (user_count == 0 || prm.n_selected == 0) return DMX_OK;
    // 20k users at the headline shape, before the contraction touched its next strip's table lines ahead of time: one
    // launch 6.31 ms, chunks of 512 7.60, 1024 6.75, 2048 6.09, 4096 6.62; with the touch 5.88 in one launch, 6.06 in
    // chunks of 2048 - so one launch is the default and the chunking stays a measurement hook
    const int64_t chunk = tuning_int("DMX_LPF_CHUNK", 0);            // tuning build only: users per chunk, 0 = one launch
    if (chunk <= 0 || chunk >= user_count) return launch_channels_fd_lpf_once(prm, ws, user_begin, user_count, gtab, out, stream);
    const size_t per_user = (size_t)prm.ue_shape[0] * prm.ue_shape[1] * prm.bs_shape[0] * prm.bs_shape[1] * (size_t)prm.n_selected;
    for (int64_t b = 0; b < user_count; b += chunk) {
        const int64_t cnt = (user_count - b) < chunk ? (user_count - b) : chunk;
        int rc = launch_channels_fd_lpf_once(prm, ws, user_begin + b, cnt, gtab, out + (size_t)b * per_user, stream);
        if (rc) return rc;
    }
    return DMX_OK;
}

static int launch_channels_fd_lpf_once(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                       float2* gtab, float2* out, hipStream_t stream) {
    bool packed = false;
    if (ws.P > 0) {
        LpfArgs a;
        a.user_begin = user_begin; a.N = prm.n_subcarriers; a.K = prm.n_selected; a.sc = prm.selected_subcarriers;
        a.gtab = gtab; a.doppler = prm.enable_doppler; a.fc = prm.carrier_freq; a.ts = 1.0 / prm.bandwidth;
        a.pack = 0;
        if ((size_t)a.N * 16 > 64 * 1024) { set_error("rx_filter variant supports at most 4096 subcarriers (got %d)", a.N); return DMX_ERR_SHAPE; }
        const bool pow2 = a.N >= 2 && (a.N & (a.N - 1)) == 0;
        int log2n = 0;
        while ((1 << log2n) < a.N) ++log2n;
        const bool old = tuning_int("DMX_LPF_OLD_FFT", 0) == 1;       // tuning build only: the workgroup-per-user FFT
        if (a.N == 512 && a.K <= 512 && ws.P <= 64 && !old && tuning_int("DMX_LPF_GENERIC_FFT", 0) != 1) {      // (longer selections: the generic kernel)
            // wave per user, N fixed (k3_lpf_fft512); tuning build only: DMX_LPF_GENERIC_FFT=1 takes the generic kernel below
            a.pack = packed = lpf_table_packed(prm, ws) && tuning_int("DMX_LPF_FLOAT_TABLE", 0) != 1;
            const bool ident = prm.sc_stride == 1 && (prm.sc_first & 511) == 0 && a.K <= 512;   // host promise: sc[k] = k (mod N)
            const void* kfns[8] = {(const void*)k3_lpf_fft512<false, false, false>, (const void*)k3_lpf_fft512<false, false, true>,
                                   (const void*)k3_lpf_fft512<false, true, false>,  (const void*)k3_lpf_fft512<false, true, true>,
                                   (const void*)k3_lpf_fft512<true, false, false>,  (const void*)k3_lpf_fft512<true, false, true>,
                                   (const void*)k3_lpf_fft512<true, true, false>,   (const void*)k3_lpf_fft512<true, true, true>};
            const void* kfn = kfns[(ident ? 4 : 0) + (packed ? 2 : 0) + (a.doppler ? 1 : 0)];
            const size_t smem = 4 * lpf_buf_elems(512) * 8;
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, smem) != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                per_cu = 1;
            }
            int64_t grid = (int64_t)device_cu_count() * per_cu;
            if (grid > (user_count + 3) / 4) grid = (user_count + 3) / 4;
            void* kargs[] = {(void*)&ws, (void*)&a, (void*)&user_count};
            hipError_t le = hipLaunchKernel(kfn, dim3((unsigned)grid), dim3(256), kargs, smem, stream);
            if (le != hipSuccess) { set_error("k3_lpf_fft512 launch failed: %s", hipGetErrorString(le)); return DMX_ERR_LAUNCH; }
        } else if ((a.N == 64 || a.N == 128 || a.N == 256 || a.N == 1024) && a.K <= a.N && ws.P <= 64 && !old &&
                   tuning_int("DMX_LPF_GENERIC_FFT", 0) != 1) {
            // the other default sizes, same scheme (k3_lpf_fft_pow2)
            a.pack = packed = lpf_table_packed(prm, ws) && tuning_int("DMX_LPF_FLOAT_TABLE", 0) != 1;
            const bool ident = prm.sc_stride == 1 && (prm.sc_first & (a.N - 1)) == 0;          // host promise: sc[k] = k (mod N)
            const int v = (ident ? 4 : 0) + (packed ? 2 : 0) + (a.doppler ? 1 : 0);
#define DMX_FFT_POW2_ROW(L2) {(const void*)k3_lpf_fft_pow2<L2, false, false, false>, (const void*)k3_lpf_fft_pow2<L2, false, false, true>, \
                              (const void*)k3_lpf_fft_pow2<L2, false, true, false>,  (const void*)k3_lpf_fft_pow2<L2, false, true, true>,  \
                              (const void*)k3_lpf_fft_pow2<L2, true, false, false>,  (const void*)k3_lpf_fft_pow2<L2, true, false, true>,  \
                              (const void*)k3_lpf_fft_pow2<L2, true, true, false>,   (const void*)k3_lpf_fft_pow2<L2, true, true, true>}
            static const void* const kfns[4][8] = {DMX_FFT_POW2_ROW(6), DMX_FFT_POW2_ROW(7), DMX_FFT_POW2_ROW(8), DMX_FFT_POW2_ROW(10)};
#undef DMX_FFT_POW2_ROW
            const void* kfn = kfns[a.N == 64 ? 0 : (a.N == 128 ? 1 : (a.N == 256 ? 2 : 3))][v];
            const int pw = a.N >= 512 ? 1 : 512 / a.N;                                          // paths per wave
            const size_t smem = (size_t)4 * pw * lpf_buf_elems(a.N) * 8;
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, smem) != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                per_cu = 1;
            }
            int64_t grid = (int64_t)device_cu_count() * per_cu;
            if (grid > (user_count + 3) / 4) grid = (user_count + 3) / 4;
            void* kargs[] = {(void*)&ws, (void*)&a, (void*)&user_count};
            hipError_t le = hipLaunchKernel(kfn, dim3((unsigned)grid), dim3(256), kargs, smem, stream);
            if (le != hipSuccess) { set_error("k3_lpf_fft_pow2 launch failed: %s", hipGetErrorString(le)); return DMX_ERR_LAUNCH; }
        } else if (pow2 && a.N >= 64 && a.N <= 2048 && !old) {
            // tuning build only: DMX_LPF_FLOAT_TABLE=1 keeps the float table
            a.pack = packed = lpf_table_packed(prm, ws) && tuning_int("DMX_LPF_FLOAT_TABLE", 0) != 1;
            // wave-per-path radix-8 FFT; persistent workgroups (the twiddle table is built once per workgroup)
            const size_t smem = lpf_wave_lds_bytes(a.N);
            const void* kfn = reinterpret_cast<const void*>(k3_lpf_fft_wave);
            if (smem > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
            }
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, smem) != hipSuccess || per_cu < 1) {
                (void)hipGetLastError();
                per_cu = 1;
            }
            int64_t grid = (int64_t)device_cu_count() * per_cu;
            if (grid > user_count) grid = user_count;
            hipLaunchKernelGGL(k3_lpf_fft_wave, dim3((unsigned)grid), dim3(256), smem, stream, ws, a, log2n, user_count);
        } else if (pow2) {
            // paths transformed together: every stage costs one workgroup barrier whatever PB is, so batch as
            // many paths as 32 KiB of LDS hold (8 at N = 512) - 25 paths then need 4 x 9 barriers instead of 25 x 9
            int PB = 4096 / a.N;
            if (PB < 1) PB = 1;
            if (PB > 16) PB = 16;
            if (PB > ws.P) PB = ws.P;
            const size_t smem = (size_t)(a.N / 2) * 8 + (size_t)PB * a.N * 8 + (size_t)PB * 4;
            hipLaunchKernelGGL(k3_lpf_fft, dim3((unsigned)user_count), dim3(256), smem, stream, ws, a, log2n, PB);
        } else {
            const int64_t blocks = user_count * ws.P;
            if (blocks > 0x7fffffffLL) { set_error("too many (user, path) pairs for one call"); return DMX_ERR_SHAPE; }
            hipLaunchKernelGGL(k3_lpf_gains, dim3((unsigned)blocks), dim3(256), (size_t)a.N * 16, stream, ws, a);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("k3 lpf gains launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    }
    return launch_channels_fd_lpf_contract(prm, ws, user_begin, user_count, gtab, out, stream, packed);
}

}  // namespace dmx
