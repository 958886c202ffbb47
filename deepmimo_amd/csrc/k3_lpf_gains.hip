// rx_filter = 1 variant: per-path subcarrier gains with the receive low-pass (sinc) filter,
//   g[l, k] = sum_{d=0}^{N-1} c_l * sinc(d - dn_l) * D_l(d) * exp(-j 2pi d sc_k / N)
// (channel.py:166-168, 193-194; D_l(d) = Doppler phase at tap delay Ts*d,
// construct_deepmimo.py:275-280, 1 when Doppler is off).  The gains go to a caller-provided table
// [user_count, P, K] complex64 that the contraction kernel (k2_channel_fd.hip, GLOAD) reads
// instead of generating exp(-j 2pi dn k / N) itself.
//
// One workgroup per (user, kept path): taps h[d] are built once in LDS (sinc in float64, as the
// reference evaluates it), the N roots of unity sit in a second LDS table, and thread k walks
// d = 0..N-1 with an integer phase index (d*sc_k mod N), so the DFT twiddles are exact table
// look-ups.  Cost is N complex MACs per (path, subcarrier): 2N/M_rx/M_tx of the main contraction -
// a variant, not the headline path, so no further tuning here.
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
    for (int d = threadIdx.x; d < a.N; d += 256) {
        const double x = (double)d - dn;
        const double px = LPF_PI * x;
        const float sn = (float)(x == 0.0 ? 1.0 : sin(px) / px);                  // np.sinc
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

int launch_channels_fd_lpf_contract(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                    const float2* gtab, float2* out, hipStream_t stream);

int launch_channels_fd_lpf(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                           float2* gtab, float2* out, hipStream_t stream) {
    if (user_count == 0 || prm.n_selected == 0) return DMX_OK;
    if (ws.P > 0) {
        LpfArgs a;
        a.user_begin = user_begin; a.N = prm.n_subcarriers; a.K = prm.n_selected; a.sc = prm.selected_subcarriers;
        a.gtab = gtab; a.doppler = prm.enable_doppler; a.fc = prm.carrier_freq; a.ts = 1.0 / prm.bandwidth;
        const size_t smem = (size_t)a.N * 16;
        if (smem > 64 * 1024) { set_error("rx_filter variant supports at most 4096 subcarriers (got %d)", a.N); return DMX_ERR_SHAPE; }
        const int64_t blocks = user_count * ws.P;
        if (blocks > 0x7fffffffLL) { set_error("too many (user, path) pairs for one call"); return DMX_ERR_SHAPE; }
        hipLaunchKernelGGL(k3_lpf_gains, dim3((unsigned)blocks), dim3(256), smem, stream, ws, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("k3_lpf_gains launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    }
    return launch_channels_fd_lpf_contract(prm, ws, user_begin, user_count, gtab, out, stream);
}

}  // namespace dmx
