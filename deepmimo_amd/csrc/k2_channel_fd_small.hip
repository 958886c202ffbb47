// Stage 2, frequency domain - small-output kernel (variant 9): one WAVE per user, no workgroup barrier.
//
// DeepMIMO's default selects ONE subcarrier (channel.py:57) and dataset-building scripts rarely take more than a
// few dozen, so a user's block [M_rx, M_tx, K] is often a few KB.  The matrix-core kernel spends ~5 us of fixed
// latency per user in that regime (A' tiles through LDS, three workgroup barriers, 1024 threads for 64 outputs), the
// fp32 vector kernel puts one subcarrier on a lane and leaves 63 lanes idle at K = 1.  Here a wave owns a user:
//   tables   b_rx[rx][l] = c_l * a_rx[rx,l],  a_tx[tx][l],  g[l][k] = exp(-j 2pi dn_l sc_k / N)   in the wave's LDS slice
//            ((M_rx + M_tx + K) * L sin/cos per user instead of M_rx * M_tx * L + L * K)
//   outputs  lane = (antenna pair p, chunk of KC subcarriers): w_l = b_rx[rx][l] * a_tx[tx][l] once per path,
//            KC complex FMAs with it; lanes walk the user's block linearly -> coalesced stores.
// Waves of a workgroup never talk to each other; LDS traffic of one wave is ordered by the hardware, the fences only
// stop the compiler from moving reads over writes.  fp32 arithmetic with the float64 phase reduction of the other
// kernels (dataset.py:398-417 + channel.py:264-284; same record layout, first 32 kept paths, the rest through
// launch_extra_path_passes).  Bound: VALU / LDS issue (tables), far below HBM: the regime is latency, not bandwidth.
#include "dmx_common.h"

namespace dmx {

struct SmallArgs {
    int64_t user_begin, user_count;
    int m_rx, m_tx, ue_mh, bs_mh;
    int K;
    const int32_t* sc;
    double inv_n;
    int ld;          // table row stride in path slots = min(P, 32)
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int KC>
__global__ __launch_bounds__(256) void k2_fd_small(WsView ws, SmallArgs a, float2* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int ld = a.ld, K = a.K, M = a.m_rx * a.m_tx;
    const int per_wave = (a.m_rx + a.m_tx + K) * ld;
    float2* brx = reinterpret_cast<float2*>(smem_raw) + (size_t)wave * per_wave;   // [m_rx][ld]
    float2* atx = brx + (size_t)a.m_rx * ld;                                        // [m_tx][ld]
    float2* g = atx + (size_t)a.m_tx * ld;                                          // [ld][K]
    const int nchunk = (K + KC - 1) / KC;
    const int total = M * nchunk;

    for (int64_t ul = (int64_t)blockIdx.x * wpb + wave; ul < a.user_count; ul += (int64_t)gridDim.x * wpb) {
        const int64_t u = a.user_begin + ul;
        float2* o = out + (size_t)ul * M * K;
        int n_act = ws.n_keep[u];
        n_act = n_act < ld ? n_act : ld;
        if (n_act == 0) {                                                   // channel.py:270-271
            for (int i = lane; i < M * K; i += 64) o[i] = make_float2(0.f, 0.f);
            continue;
        }
        const size_t rb = (size_t)u * ws.P;
        wave_lds_fence();                                                   // previous user's table reads are done
        for (int i = lane; i < a.m_rx * n_act; i += 64) {
            const int r = i / n_act, l = i - r * n_act;
            float s, c;
            sincos_rev(frac_rev((double)(r % a.ue_mh) * ws.rx_y[rb + l] + (double)(r / a.ue_mh) * ws.rx_z[rb + l]), s, c);
            const float cr = ws.c_re[rb + l], ci = ws.c_im[rb + l];
            brx[r * ld + l] = make_float2(cr * c - ci * s, cr * s + ci * c);
        }
        for (int i = lane; i < a.m_tx * n_act; i += 64) {
            const int t = i / n_act, l = i - t * n_act;
            float s, c;
            sincos_rev(frac_rev((double)(t % a.bs_mh) * ws.tx_y[rb + l] + (double)(t / a.bs_mh) * ws.tx_z[rb + l]), s, c);
            atx[t * ld + l] = make_float2(c, s);
        }
        for (int i = lane; i < n_act * K; i += 64) {
            const int l = i / K, k = i - l * K;
            float s, c;
            sincos_rev(frac_rev((double)ws.dn[rb + l] * a.inv_n * (double)a.sc[k]), s, c);
            g[l * K + k] = make_float2(c, -s);                              // exp(-j 2pi x) = cos - j sin
        }
        wave_lds_fence();

        for (int e = lane; e < total; e += 64) {
            const int p = e / nchunk, k0 = (e - p * nchunk) * KC;
            const int rx = p / a.m_tx, tx = p - rx * a.m_tx;
            const float2* br = brx + rx * ld;
            const float2* at = atx + tx * ld;
            int kj[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) kj[j] = (k0 + j) < K ? (k0 + j) : (K - 1);
            float2 acc[KC];
#pragma unroll
            for (int j = 0; j < KC; ++j) acc[j] = make_float2(0.f, 0.f);
            for (int l = 0; l < n_act; ++l) {
                const float2 b = br[l], t = at[l];
                const float wr = b.x * t.x - b.y * t.y, wi = b.x * t.y + b.y * t.x;
                const float2* gl = g + l * K;
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    const float2 v = gl[kj[j]];
                    acc[j].x += wr * v.x - wi * v.y;
                    acc[j].y += wr * v.y + wi * v.x;
                }
            }
            float2* dst = o + (size_t)p * K + k0;
#pragma unroll
            for (int j = 0; j < KC; ++j)
                if (k0 + j < K) dst[j] = acc[j];
        }
    }
}

// One wave's tables: four waves share a workgroup while 4 x tables fit the 64 KB a workgroup gets by default, then
// two, then one; a single wave may take up to 156 KB (of the CU's 160 KB) with the dynamic-LDS attribute raised.
static constexpr size_t SMALL_LDS_MAX = 156 * 1024;
static int small_waves_per_block(const dmx_params& prm, const WsView& ws) {
    const int ld = ws.P < 32 ? ws.P : 32;
    const size_t bytes = (size_t)(prm.ue_shape[0] * prm.ue_shape[1] + prm.bs_shape[0] * prm.bs_shape[1] + prm.n_selected) * ld * 8;
    if (bytes == 0) return 0;
    if (bytes * 4 <= 64 * 1024) return 4;
    if (bytes * 2 <= 64 * 1024) return 2;
    if (bytes <= SMALL_LDS_MAX) return 1;
    return 0;
}

bool fd_small_supported(const dmx_params& prm, const WsView& ws) { return small_waves_per_block(prm, ws) > 0; }

// Automatic choice, from tools/small_k_sweep.sh (200k users, 25 paths, ms for matrix-core | vector | this kernel):
//   64 pairs x K=1   2.08 | 2.98 | 0.14      64 pairs x K=8    3.71 | 5.09 | 1.08     1024 pairs x K=2  16.1 | 78.6 | 2.6
//   256 pairs x K=8  5.32 | 34.7 | 7.15      256 pairs x K=16  6.56 | 23.3 | 9.6
//   8 pairs x K=1    1.90 | 0.44 | 0.10      8 pairs x K=16    3.50 | 0.96 | 0.53     8 pairs x K=64    4.02 | 1.32 | 1.91
// i.e. up to ~8 subcarriers this kernel wins unless the tables push it to one wave per workgroup; below 24 antenna
// pairs (where the matrix cores are not used) it wins up to 16 subcarriers, the subcarrier-per-lane kernel beyond.
bool fd_small_preferred(const dmx_params& prm, const WsView& ws) {
    const int wpb = small_waves_per_block(prm, ws);
    if (!wpb) return false;
    const int K = prm.n_selected;
    const int64_t M = (int64_t)prm.ue_shape[0] * prm.ue_shape[1] * prm.bs_shape[0] * prm.bs_shape[1];
    if (M < 24) return K <= 16;
    if (K <= 4) return true;
    return K <= 8 && wpb >= 2 && M * K <= 2048;
}

template <int KC>
static int launch_small_t(const WsView& ws, const SmallArgs& a, dim3 g, dim3 b, size_t smem, float2* out, hipStream_t stream) {
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k2_fd_small<KC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMALL_LDS_MAX);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL((k2_fd_small<KC>), g, b, smem, stream, ws, a, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k2_fd_small launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

int launch_channels_fd_small(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                             float2* out, hipStream_t stream) {
    const int wpb = small_waves_per_block(prm, ws);
    if (!wpb) { set_error("small-output kernel: tables of one user do not fit the LDS"); return DMX_ERR_SHAPE; }
    SmallArgs a;
    a.user_begin = user_begin; a.user_count = user_count;
    a.m_rx = prm.ue_shape[0] * prm.ue_shape[1];
    a.m_tx = prm.bs_shape[0] * prm.bs_shape[1];
    a.ue_mh = prm.ue_shape[0];
    a.bs_mh = prm.bs_shape[0];
    a.K = prm.n_selected;
    a.sc = prm.selected_subcarriers;
    a.inv_n = 1.0 / (double)prm.n_subcarriers;
    a.ld = ws.P < 32 ? ws.P : 32;
    const size_t smem = (size_t)wpb * (a.m_rx + a.m_tx + a.K) * a.ld * 8;
    // persistent: as many workgroups as the LDS lets be resident (160 KB per CU), at most 8 waves per SIMD
    int per_cu = (int)((size_t)160 * 1024 / smem);
    if (per_cu * wpb > 32) per_cu = 32 / wpb;
    if (per_cu < 1) per_cu = 1;
    int64_t grid = (user_count + wpb - 1) / wpb;
    if (grid > (int64_t)256 * per_cu) grid = (int64_t)256 * per_cu;
    const dim3 g((unsigned)grid), b(64 * wpb);
    if (a.K >= 4) return launch_small_t<4>(ws, a, g, b, smem, out, stream);
    if (a.K >= 2) return launch_small_t<2>(ws, a, g, b, smem, out, stream);
    return launch_small_t<1>(ws, a, g, b, smem, out, stream);
}

}  // namespace dmx
