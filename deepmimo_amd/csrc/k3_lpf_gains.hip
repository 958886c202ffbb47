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
#include "dmx_common.h"

namespace dmx {

struct LpfArgs {
    int64_t user_begin;
    int N, K;
    const int32_t* sc;
    float2* gtab;
    int doppler;
    double fc, ts;
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

int launch_channels_fd_lpf_contract(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                    const float2* gtab, float2* out, hipStream_t stream);

int launch_channels_fd_lpf(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                           float2* gtab, float2* out, hipStream_t stream) {
    if (user_count == 0 || prm.n_selected == 0) return DMX_OK;
    if (ws.P > 0) {
        LpfArgs a;
        a.user_begin = user_begin; a.N = prm.n_subcarriers; a.K = prm.n_selected; a.sc = prm.selected_subcarriers;
        a.gtab = gtab; a.doppler = prm.enable_doppler; a.fc = prm.carrier_freq; a.ts = 1.0 / prm.bandwidth;
        if ((size_t)a.N * 16 > 64 * 1024) { set_error("rx_filter variant supports at most 4096 subcarriers (got %d)", a.N); return DMX_ERR_SHAPE; }
        const bool pow2 = a.N >= 2 && (a.N & (a.N - 1)) == 0;
        if (pow2) {
            int log2n = 0;
            while ((1 << log2n) < a.N) ++log2n;
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
    return launch_channels_fd_lpf_contract(prm, ws, user_begin, user_count, gtab, out, stream);
}

}  // namespace dmx
