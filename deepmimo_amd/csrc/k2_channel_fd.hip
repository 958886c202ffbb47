// Stage 2, frequency domain - fp32 vector kernel (variant 1).
//
// out[u, rx, tx, k] = sum_l a_rx[rx,l] * a_tx[tx,l] * c_l * exp(-j 2pi dn_l sc_k / N)
// replaces Dataset._compute_array_response_product (dataset.py:398-417; the reference materialises a
// complex128 [N, M_rx, M_tx, L] tensor - never built here) and the per-user loop of
// _generate_MIMO_channel (channel.py:264-284; complex128 broadcast product + nansum).
//
// Mapping: one 256-thread workgroup per user.  Per user the two small factor tables
//   b_rx[r][l] = c_l * a_rx[r,l]   and   a_tx[t][l]
// are built once in LDS (sincos of a float64-range-reduced phase, so element-index multiples of the
// per-path step cost no accuracy).  Each wave then owns 64-subcarrier chunks: lane = subcarrier, so
// the lane keeps its L subcarrier phasors g_l = exp(-j 2pi frac(dn_l*k/N)) in registers (the
// frac() is done in float64: dn*k needs 34 bits, SURVEY finding 6), forms t_l = b_rx[r][l]*g_l per
// receive element and walks the transmit elements with 4*L FMAs per output; the a_tx row is a
// wave-uniform LDS broadcast read, shared by RB receive elements to halve LDS traffic.
// Stores are 8 B/lane, 512 B contiguous per wave instruction (K is the fastest output index).
//
// Roofline: 8*M_rx*M_tx*K output bytes per user vs 8*L flop per output element: at L = 25 this is
// at the fp32 ridge of the chip (25 flop/B), so this kernel is VALU-bound; the MFMA variant
// (k2_channel_fd_mfma.hip) removes that bound.  This one stays as the exact-fp32, any-shape path.
#include "dmx_common.h"

namespace dmx {

typedef float v2f __attribute__((ext_vector_type(2)));

struct FdArgs {
    int64_t user_begin;
    int m_rx, m_tx, ue_mh, bs_mh;
    int K;
    const int32_t* sc;
    double inv_n;
    int txt;        // transmit elements per LDS tile
    const float2* gtab;   // rx_filter variant: precomputed path gains [user_count, P, K] (k3_lpf_gains.hip)
    int l0;               // first path slot of this pass (users with more than 32 kept paths take several passes)
    int accumulate;       // 1: add to what earlier passes wrote
};

// Everything one workgroup does for one user with LPA (multiple of 4) path slots; slots beyond the
// user's n_act paths hold zero table entries, so the unrolled loops need no guards.
// Synchronisation among the WPU waves that share one user's tables: a workgroup barrier for WPU = 4; for
// WPU = 1 the wave's own LDS operations are already executed in order, so only the compiler has to be held back.
template <int WPU>
__device__ __forceinline__ void user_sync() {
    if constexpr (WPU == 4) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int LPA, int RB, bool GLOAD, int WPU>
__device__ __forceinline__ void fd_user(const WsView& ws, const FdArgs& a, float2* __restrict__ o, int64_t u,
                                        int64_t u_local, int n_act, unsigned char* smem) {
    constexpr int NTU = WPU * 64;                               // threads working on this user
    double* q = reinterpret_cast<double*>(smem);                 // [LPA]  dn_l / N
    float* brx = reinterpret_cast<float*>(q + LPA);             // [m_rx][2*LPA]
    float* atx = brx + (size_t)a.m_rx * 2 * LPA;                // [txt][2*LPA]
    const int tid = threadIdx.x % NTU, lane = tid & 63, wave = tid >> 6;
    const size_t rb = (size_t)u * ws.P + a.l0;

    if (tid < LPA) q[tid] = tid < n_act ? (double)ws.dn[rb + tid] * a.inv_n : 0.0;
    for (int i = tid; i < a.m_rx * LPA; i += NTU) {
        const int r = i / LPA, l = i - r * LPA;
        float re = 0.f, im = 0.f;
        if (l < n_act) {
            const int y = r % a.ue_mh, z = r / a.ue_mh;
            float s, c;
            sincos_rev(frac_rev((double)y * ws.rx_y[rb + l] + (double)z * ws.rx_z[rb + l]), s, c);
            // with precomputed gains the path coefficient c_l is already inside g
            const float cr = GLOAD ? 1.0f : ws.c_re[rb + l], ci = GLOAD ? 0.0f : ws.c_im[rb + l];
            re = cr * c - ci * s;
            im = cr * s + ci * c;
        }
        brx[(size_t)r * 2 * LPA + 2 * l] = re;
        brx[(size_t)r * 2 * LPA + 2 * l + 1] = im;
    }

    const int nchunks = (a.K + 63) >> 6;
    for (int tx0 = 0; tx0 < a.m_tx; tx0 += a.txt) {
        const int ntx = (a.m_tx - tx0) < a.txt ? (a.m_tx - tx0) : a.txt;
        user_sync<WPU>();                                        // previous tile fully consumed
        for (int i = tid; i < ntx * LPA; i += NTU) {
            const int t = i / LPA, l = i - t * LPA;
            float s = 0.f, c = 0.f;
            if (l < n_act) {
                const int m = tx0 + t, y = m % a.bs_mh, z = m / a.bs_mh;
                sincos_rev(frac_rev((double)y * ws.tx_y[rb + l] + (double)z * ws.tx_z[rb + l]), s, c);
            }
            atx[(size_t)t * 2 * LPA + 2 * l] = c;
            atx[(size_t)t * 2 * LPA + 2 * l + 1] = s;
        }
        user_sync<WPU>();

        for (int ch = wave; ch < nchunks; ch += WPU) {
            const int kidx = (ch << 6) + lane;
            const bool kok = kidx < a.K;
            const double kk = (double)(kok ? a.sc[kidx] : 0);
            float g_re[LPA], g_im[LPA];
#pragma unroll
            for (int l = 0; l < LPA; ++l) {
                if constexpr (GLOAD) {
                    float2 v = make_float2(0.f, 0.f);
                    if (kok && l < n_act) v = a.gtab[((size_t)u_local * ws.P + a.l0 + l) * a.K + kidx];
                    g_re[l] = v.x; g_im[l] = v.y;
                } else {
                    float s, c;
                    sincos_rev(frac_rev(q[l] * kk), s, c);      // exp(-j 2pi x) = cos - j sin
                    g_re[l] = c; g_im[l] = -s;
                    __builtin_amdgcn_sched_barrier(0);           // one phasor at a time: keeps live temporaries low
                }
            }
            for (int rx0 = 0; rx0 < a.m_rx; rx0 += RB) {
                v2f t[RB][LPA];                                  // t_l = b_rx[r][l] * g_l as (re, im) pairs
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const int r = (rx0 + b) < a.m_rx ? (rx0 + b) : (a.m_rx - 1);
                    const float4* brow = reinterpret_cast<const float4*>(brx + (size_t)r * 2 * LPA);
#pragma unroll
                    for (int l = 0; l < LPA; l += 2) {
                        const float4 v = brow[l / 2];
                        t[b][l].x = v.x * g_re[l] - v.y * g_im[l];
                        t[b][l].y = v.x * g_im[l] + v.y * g_re[l];
                        t[b][l + 1].x = v.z * g_re[l + 1] - v.w * g_im[l + 1];
                        t[b][l + 1].y = v.z * g_im[l + 1] + v.w * g_re[l + 1];
                    }
                }
                for (int tx = 0; tx < ntx; ++tx) {
                    const float4* arow = reinterpret_cast<const float4*>(atx + (size_t)tx * 2 * LPA);
                    // H = sum (ar + j ai)(tr + j ti): accA += ar*(tr,ti), accB += ai*(tr,ti);
                    // re = accA.x - accB.y, im = accA.y + accB.x  -> two packed FMAs per path, no swapped copies.
                    v2f accA[RB], accB[RB];
#pragma unroll
                    for (int b = 0; b < RB; ++b) { accA[b] = v2f{0.f, 0.f}; accB[b] = v2f{0.f, 0.f}; }
#pragma unroll
                    for (int l = 0; l < LPA; l += 2) {
                        const float4 v = arow[l / 2];
#pragma unroll
                        for (int b = 0; b < RB; ++b) {
                            accA[b] = __builtin_elementwise_fma(v2f{v.x, v.x}, t[b][l], accA[b]);
                            accB[b] = __builtin_elementwise_fma(v2f{v.y, v.y}, t[b][l], accB[b]);
                            accA[b] = __builtin_elementwise_fma(v2f{v.z, v.z}, t[b][l + 1], accA[b]);
                            accB[b] = __builtin_elementwise_fma(v2f{v.w, v.w}, t[b][l + 1], accB[b]);
                        }
                    }
                    if (kok) {
#pragma unroll
                        for (int b = 0; b < RB; ++b) {
                            if (rx0 + b < a.m_rx) {
                                float2* dst = o + ((size_t)(rx0 + b) * a.m_tx + (tx0 + tx)) * a.K + kidx;
                                float2 v = make_float2(accA[b].x - accB[b].y, accA[b].y + accB[b].x);
                                if (a.accumulate) { const float2 prev = *dst; v.x += prev.x; v.y += prev.y; }
                                *dst = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// LPMAX = slots the launch provides LDS/registers for (>= P); each user runs the smallest body
// that holds its n_act compacted paths (real ray-traced users have far fewer than P paths).
// WPU = waves per user: 4 (one user per 256-thread workgroup) or 1 (four users per workgroup, one wave each,
// no workgroup barrier) for shapes whose subcarriers fit one 64-lane chunk - there three of four waves would
// idle and the per-workgroup fixed costs (launch, table build, barriers) dominate the tiny per-user work.
template <int LPMAX, int RB, bool GLOAD, int WPU>
__global__ __launch_bounds__(256, 2) void k2_fd_valu(WsView ws, FdArgs a, float2* __restrict__ out, int64_t user_count,
                                                     int smem_per_user) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    constexpr int UPW = 4 / WPU, NTU = WPU * 64;
    const int slot = threadIdx.x / NTU, tid = threadIdx.x % NTU;
    const int64_t ul = (int64_t)blockIdx.x * UPW + slot;
    if (ul >= user_count) return;                                // whole waves only: slot boundaries are wave boundaries
    unsigned char* smem = smem_all + (size_t)slot * smem_per_user;
    const int64_t u = a.user_begin + ul;
    const size_t per_user = (size_t)a.m_rx * a.m_tx * a.K;
    float2* __restrict__ o = out + (size_t)ul * per_user;
    int n_act = ws.n_keep[u] - a.l0;
    n_act = n_act < LPMAX ? n_act : LPMAX;
    if (n_act <= 0) {                                            // channel.py:270-271: stays all-zero
        if (!a.accumulate)
            for (size_t i = tid; i < per_user; i += NTU) o[i] = make_float2(0.f, 0.f);
        return;
    }
    const int n4 = (n_act + 3) >> 2;
    if constexpr (LPMAX >= 32) { if (n4 == 8) { fd_user<32, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem); return; } }
    if constexpr (LPMAX >= 28) { if (n4 == 7) { fd_user<28, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem); return; } }
    if constexpr (LPMAX >= 24) { if (n4 == 6) { fd_user<24, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem); return; } }
    if constexpr (LPMAX >= 20) { if (n4 == 5) { fd_user<20, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem); return; } }
    if constexpr (LPMAX >= 16) { if (n4 == 4) { fd_user<16, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem); return; } }
    if constexpr (LPMAX >= 12) { if (n4 == 3) { fd_user<12, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem); return; } }
    if constexpr (LPMAX >= 8) { if (n4 == 2) { fd_user<8, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem); return; } }
    fd_user<4, RB, GLOAD, WPU>(ws, a, o, u, ul, n_act, smem);
}

template <int LP, int RB, bool GLOAD, int WPU>
static int launch_valu_w(const WsView& ws, FdArgs a, int64_t user_count, float2* out, hipStream_t stream) {
    constexpr int UPW = 4 / WPU;
    const size_t fixed = (size_t)LP * 8 + (size_t)a.m_rx * 2 * LP * 4;
    const size_t budget = (64 * 1024) / UPW;                     // LDS share of one user
    if (fixed + (size_t)2 * LP * 4 > budget) {
        set_error("UE array of %d elements does not fit the LDS tables (max about %d)", a.m_rx, (int)(budget / (8 * LP)) - 2);
        return DMX_ERR_SHAPE;
    }
    int txt = (int)((budget - fixed) / ((size_t)2 * LP * 4));
    if (txt > a.m_tx) txt = a.m_tx;
    a.txt = txt;
    const size_t per_user = align_up(fixed + (size_t)txt * 2 * LP * 4, 16);
    const int64_t blocks = (user_count + UPW - 1) / UPW;
    hipLaunchKernelGGL((k2_fd_valu<LP, RB, GLOAD, WPU>), dim3((unsigned)blocks), dim3(256), per_user * UPW, stream, ws, a, out,
                       user_count, (int)per_user);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k2_fd_valu launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

template <int LP, int RB, bool GLOAD = false>
static int launch_valu(const WsView& ws, const FdArgs& a, int64_t user_count, float2* out, hipStream_t stream) {
    // one wave per user (four users per workgroup, no workgroup barrier) for tiny panels, whatever the subcarrier
    // count: measured, 200k users x 25 paths, 8 pairs: K=64 1.89 -> 1.33 ms, K=128 4.07 -> 2.14, K=256 4.79 -> 3.71,
    // K=512 8.39 -> 6.95; 4 pairs x 256 3.36 -> 2.42.  From 9 pairs on the four-wave form (or the matrix cores) wins.
    // (Turning the phasors from one 64-subcarrier chunk to the next by a per-path rotation instead of a fresh
    // sin/cos was tried here: the kernel already spills at 256 VGPRs and the extra live state made it 2.4x slower.)
    const size_t need = (size_t)LP * 8 + (size_t)(a.m_rx + a.m_tx) * 2 * LP * 4;
    if (a.m_rx * a.m_tx <= 8 && need <= 16 * 1024 - 16)
        return launch_valu_w<LP, RB, GLOAD, 1>(ws, a, user_count, out, stream);
    return launch_valu_w<LP, RB, GLOAD, 4>(ws, a, user_count, out, stream);
}

static int launch_fd_valu_any(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                              const float2* gtab, float2* out, hipStream_t stream, int l0 = 0, int accumulate = 0) {
    FdArgs a;
    a.l0 = l0; a.accumulate = accumulate;
    a.user_begin = user_begin;
    a.m_rx = prm.ue_shape[0] * prm.ue_shape[1];
    a.m_tx = prm.bs_shape[0] * prm.bs_shape[1];
    a.ue_mh = prm.ue_shape[0];
    a.bs_mh = prm.bs_shape[0];
    a.K = prm.n_selected;
    a.sc = prm.selected_subcarriers;
    a.inv_n = 1.0 / (double)prm.n_subcarriers;
    a.txt = 0;
    a.gtab = gtab;
    const int P = (ws.P - l0) < 32 ? (ws.P - l0) : 32;       // path slots of this pass
    // RB = receive elements sharing one a_tx row read.  2 halves the LDS traffic but doubles the t
    // registers; beyond 16 path slots that would spill, so the long-path bodies use RB = 1.
    const bool rb2 = a.m_rx >= 2;
    if (gtab) {
        if (P <= 8) return launch_valu<8, 1, true>(ws, a, user_count, out, stream);
        if (P <= 16) return launch_valu<16, 1, true>(ws, a, user_count, out, stream);
        if (P <= 28) return launch_valu<28, 1, true>(ws, a, user_count, out, stream);
        if (P <= 32) return launch_valu<32, 1, true>(ws, a, user_count, out, stream);
    } else {
        if (P <= 8) return rb2 ? launch_valu<8, 2>(ws, a, user_count, out, stream) : launch_valu<8, 1>(ws, a, user_count, out, stream);
        if (P <= 16) return rb2 ? launch_valu<16, 2>(ws, a, user_count, out, stream) : launch_valu<16, 1>(ws, a, user_count, out, stream);
        if (P <= 28) return launch_valu<28, 1>(ws, a, user_count, out, stream);
        if (P <= 32) return launch_valu<32, 1>(ws, a, user_count, out, stream);
    }
    set_error("internal: %d path slots in one pass", P);
    return DMX_ERR_SHAPE;
}

// More than 32 path slots (beyond DeepMIMO's MAX_PATHS = 25, but allowed): the first 32 slots go through the
// chosen kernel, every further block of 32 through the fp32 vector kernel in accumulate mode (users whose kept
// paths ended earlier return at once).  The output is read back once per extra pass - the price of the rare case.
static int launch_extra_path_passes(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                    const float2* gtab, float2* out, hipStream_t stream) {
    for (int l0 = 32; l0 < ws.P; l0 += 32) {
        int rc = launch_fd_valu_any(prm, ws, user_begin, user_count, gtab, out, stream, l0, 1);
        if (rc) return rc;
    }
    return DMX_OK;
}

int launch_channels_fd_mfma(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                            float2* out, int config, hipStream_t stream);
bool fd_mfma_supported(const dmx_params& prm, const WsView& ws);
bool fd_mfma_preferred(const dmx_params& prm, const WsView& ws);
int launch_channels_fd_small(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                             float2* out, hipStream_t stream);
bool fd_small_preferred(const dmx_params& prm, const WsView& ws);
int launch_channels_fd_fold(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count, float2* out,
                            int chunk_blocks, hipStream_t stream);
bool fd_fold_preferred(const dmx_params& prm, const WsView& ws);

// what variant 0 runs for this shape: 9 small-output kernel, 12 folded matrix-core kernel (few antenna pairs, uniformly
// spaced subcarriers), 2 matrix cores, 1 fp32 vector kernel
int fd_auto_choice(const dmx_params& prm, const WsView& ws) {
    if (fd_fold_preferred(prm, ws)) return 12;
    if (fd_small_preferred(prm, ws)) return 9;
    if (fd_mfma_preferred(prm, ws)) return 2;
    return 1;
}

int launch_channels_fd(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                       float2* out, int variant, hipStream_t stream) {
    if (user_count == 0) return DMX_OK;
    if (variant >= 2 && variant != 9 && variant != 12 && !fd_mfma_supported(prm, ws)) {
        set_error("MFMA variant does not support this shape");
        return DMX_ERR_SHAPE;
    }
    int rc;
    if (variant == 0) variant = fd_auto_choice(prm, ws);
    if (variant == 9)
        rc = launch_channels_fd_small(prm, ws, user_begin, user_count, out, stream);
    else if (variant == 12)
        rc = launch_channels_fd_fold(prm, ws, user_begin, user_count, out, 0, stream);
    else if (variant >= 2)
        rc = launch_channels_fd_mfma(prm, ws, user_begin, user_count, out, variant >= 3 ? variant - 2 : 0, stream);
    else
        rc = launch_fd_valu_any(prm, ws, user_begin, user_count, nullptr, out, stream);
    return rc ? rc : launch_extra_path_passes(prm, ws, user_begin, user_count, nullptr, out, stream);
}

int launch_channels_fd_mfma_gload(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                  const float2* gtab, float2* out, hipStream_t stream, bool packed);

// the gains table may be written in the packed f16 form when the matrix-core kernel is its only reader (at most 32 path
// slots: further slots go through the fp32 vector kernel's accumulate passes, which read floats)
bool lpf_table_packed(const dmx_params& prm, const WsView& ws) { return fd_mfma_preferred(prm, ws) && ws.P <= 32; }

int launch_channels_fd_lpf_contract(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                    const float2* gtab, float2* out, hipStream_t stream, bool packed) {
    int rc;
    if (fd_mfma_preferred(prm, ws)) rc = launch_channels_fd_mfma_gload(prm, ws, user_begin, user_count, gtab, out, stream, packed);
    else rc = launch_fd_valu_any(prm, ws, user_begin, user_count, gtab, out, stream);
    return rc ? rc : launch_extra_path_passes(prm, ws, user_begin, user_count, gtab, out, stream);
}

}  // namespace dmx
