// Stage 2, time domain: out[u, rx, tx, s] = a_rx[rx,s] * a_tx[tx,s] * sqrt(p_s) e^{j phase_s}
// for the s-th valid path of user u (valid paths compacted to the front), zero for the remaining
// slots.  Replaces the TD branch of _generate_MIMO_channel (channel.py:285-287).
//
// The output is only [M_rx, M_tx, P] per user (P <= 25): an HBM-store-bound elementwise kernel,
// thread = output element (slot fastest => contiguous complex64 stores).  k4_td_tab first builds the two
// factor tables b_rx[r][s] = c_s a_rx[r,s] and a_tx[t][s] in LDS (float64-range-reduced phases, one sin/cos
// per table entry instead of one per output element) and multiplies them out; k4_td is the table-free form
// for panels whose tables exceed LDS.
#include "dmx_common.h"
#include "dmx_tuning.h"
#include <stdlib.h>

namespace dmx {

struct TdArgs {
    int64_t user_begin;
    int m_rx, m_tx, ue_mh, bs_mh;
};

__global__ __launch_bounds__(256) void k4_td(WsView ws, TdArgs a, float2* __restrict__ out) {
    const int64_t u = a.user_begin + blockIdx.x;
    const int P = ws.P;
    const size_t per_user = (size_t)a.m_rx * a.m_tx * P;
    float2* __restrict__ o = out + (size_t)blockIdx.x * per_user;
    const int n_keep = ws.n_keep[u];
    const size_t rb = (size_t)u * P;
    for (size_t i = threadIdx.x; i < per_user; i += 256) {
        const int s = (int)(i % P);
        const int m = (int)((i / P) % a.m_tx);
        const int r = (int)(i / ((size_t)P * a.m_tx));
        float2 v = make_float2(0.f, 0.f);
        if (s < n_keep) {
            const int ty = m % a.bs_mh, tz = m / a.bs_mh, ry = r % a.ue_mh, rz = r / a.ue_mh;
            const double ph = (double)ty * ws.tx_y[rb + s] + (double)tz * ws.tx_z[rb + s] +
                              (double)ry * ws.rx_y[rb + s] + (double)rz * ws.rx_z[rb + s];
            float sn, cs;
            sincos_rev(frac_rev(ph), sn, cs);
            const float cr = ws.c_re[rb + s], ci = ws.c_im[rb + s];
            v.x = cr * cs - ci * sn;
            v.y = cr * sn + ci * cs;
        }
        o[i] = v;
    }
}

template <bool PAIRS>
__global__ __launch_bounds__(256) void k4_td_tab(WsView ws, TdArgs a, float2* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int P = ws.P;
    float2* brx = reinterpret_cast<float2*>(smem);          // [m_rx][P]
    float2* atx = brx + (size_t)a.m_rx * P;                 // [m_tx][P]
    const int64_t u = a.user_begin + blockIdx.x;
    const size_t per_user = (size_t)a.m_rx * a.m_tx * P;
    float2* __restrict__ o = out + (size_t)blockIdx.x * per_user;
    const int n_keep = ws.n_keep[u];
    const size_t rb = (size_t)u * P;
    for (int i = threadIdx.x; i < (a.m_rx + a.m_tx) * P; i += 256) {
        const int e = i / P, s = i - e * P;                 // e < m_rx: receive element, else transmit element
        float2 v = make_float2(0.f, 0.f);
        if (s < n_keep) {
            float sn, cs;
            if (e < a.m_rx) {
                sincos_rev(frac_rev((double)(e % a.ue_mh) * ws.rx_y[rb + s] + (double)(e / a.ue_mh) * ws.rx_z[rb + s]), sn, cs);
                const float cr = ws.c_re[rb + s], ci = ws.c_im[rb + s];
                v = make_float2(cr * cs - ci * sn, cr * sn + ci * cs);
            } else {
                const int t = e - a.m_rx;
                sincos_rev(frac_rev((double)(t % a.bs_mh) * ws.tx_y[rb + s] + (double)(t / a.bs_mh) * ws.tx_z[rb + s]), sn, cs);
                v = make_float2(cs, sn);
            }
        }
        brx[i] = v;                                         // the two tables are contiguous
    }
    __syncthreads();
    // thread = PAIR of consecutive output elements, 512 elements apart per step (16-byte non-temporal stores: a wave writes
    // 1 KiB per instruction); (slot s, transmit element m, receive element r) of the pair's first element are advanced
    // incrementally - three 64-bit divisions per element were most of this kernel's instructions.  PAIRS needs an even
    // element count per user (16-byte aligned user blocks); otherwise one element per thread, 256 apart.
    constexpr unsigned STEP = PAIRS ? 512u : 256u;
    const unsigned tid = threadIdx.x;
    const unsigned e0 = PAIRS ? 2u * tid : tid;
    const unsigned uP = (unsigned)P, uT = (unsigned)a.m_tx;
    unsigned sidx = e0 % uP, q = e0 / uP;                                   // q = r * m_tx + m
    unsigned m = q % uT, r = q / uT;
    const unsigned ds = STEP % uP, dq = STEP / uP;
    const unsigned dm = dq % uT, dr = dq / uT;
    typedef float f4v __attribute__((ext_vector_type(4)));
    for (size_t i = e0; i < per_user; i += STEP) {
        const float2 b = brx[r * uP + sidx], t = atx[m * uP + sidx];
        const float2 v0 = make_float2(b.x * t.x - b.y * t.y, b.x * t.y + b.y * t.x);
        if (PAIRS) {
            unsigned s1 = sidx + 1, m1 = m, r1 = r;
            if (s1 == uP) { s1 = 0; if (++m1 == uT) { m1 = 0; ++r1; } }
            const float2 b1 = brx[r1 * uP + s1], t1 = atx[m1 * uP + s1];    // i + 1 < per_user: per_user is even
            f4v w = {v0.x, v0.y, b1.x * t1.x - b1.y * t1.y, b1.x * t1.y + b1.y * t1.x};
            __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(o + i));
        } else {
            o[i] = v0;
        }
        sidx += ds;
        unsigned carry = 0;
        if (sidx >= uP) { sidx -= uP; carry = 1; }
        m += dm + carry;
        r += dr;
        if (m >= uT) { m -= uT; ++r; }
        if (m >= uT) { m -= uT; ++r; }                                      // dm + carry can reach m_tx
    }
}

int launch_channels_td(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                       float2* out, hipStream_t stream) {
    if (user_count == 0 || ws.P == 0) return DMX_OK;
    TdArgs a;
    a.user_begin = user_begin;
    a.m_rx = prm.ue_shape[0] * prm.ue_shape[1];
    a.m_tx = prm.bs_shape[0] * prm.bs_shape[1];
    a.ue_mh = prm.ue_shape[0];
    a.bs_mh = prm.bs_shape[0];
    const size_t smem = (size_t)(a.m_rx + a.m_tx) * ws.P * 8;
    const bool pairs = ((size_t)a.m_rx * a.m_tx * ws.P) % 2 == 0 && ((uintptr_t)out % 16) == 0 && !tuning_set("DMX_TD_NARROW");   // tuning build only: 8-byte stores
    if (smem <= 64 * 1024 && pairs)
        hipLaunchKernelGGL(k4_td_tab<true>, dim3((unsigned)user_count), dim3(256), smem, stream, ws, a, out);
    else if (smem <= 64 * 1024)
        hipLaunchKernelGGL(k4_td_tab<false>, dim3((unsigned)user_count), dim3(256), smem, stream, ws, a, out);
    else
        hipLaunchKernelGGL(k4_td, dim3((unsigned)user_count), dim3(256), 0, stream, ws, a, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k4_td launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

}  // namespace dmx
