// Fused consumer of H that writes NO [N, ., K] tensor: the beam-sweep reduction of
// docs/manual.ipynb cell 105,
//     mean_amplitude[u, b] = np.abs(F1 @ dataset.channel).mean(axis=1).mean(axis=-1)
//                          = 1/(M_rx K) * sum_rx sum_k | sum_tx F[b,tx] H[u,rx,tx,k] |,
// straight from the ray records.  The beam-space channel Y[(rx,b), k] = sum_l a_rx[rx,l] f[b,l] G[l,k] is the same
// split-precision matrix-core contraction as k2_channel_fd_mfma.hip (f = F a_tx from k2b_beam_project), but the 32x32
// accumulator tiles never leave the registers: each (re, im) lane pair forms |Y|, every lane keeps 16 running row sums
// over the subcarriers it sees, and one wave reduction + LDS add per user yields the [n_beams] means.  HBM traffic
// per user: the path records and n_beams floats - the kernel is bound by the matrix cores (and the B' generation),
// not by memory: 105 GB of beam-space output (64 beams, headline shape) are not written and not read back.
//
// Mapping (one 512-thread workgroup per user, persistent): the waves own ROW tiles here (their A' fragments stay in
// registers for the whole user), so the B' fragments are needed by every wave and live in LDS in register layout
// (conflict-free ds_read_b128 / ds_write_b128).  Because no interleaved (re, im) stream has to come out, the real GEMM's
// columns are NOT interleaved here: a WIDE strip covers 32 subcarriers with TWO accumulators - one whose columns are
// the real parts (B' rows (Re G, -Im G)), one for the imaginary parts ((Im G, Re G)) - so |Y|^2 = re^2 + im^2 forms
// inside one lane (no lane-pair exchange, no duplicated square roots), and both B' operand sets come from ONE f16 split
// of G per (path, subcarrier): the (Re, -Im) pair is the split with the upper sign flipped, the (Im, Re) pair the same
// word rotated by 16 bits.  Chunks of 2 wide strips (64 subcarriers, 32 KB) in two buffers: each wave generates one
// K-step of one wide strip of chunk c + 1, then runs its tile against chunk c - one barrier per chunk; with fewer than
// eight row tiles the waves split the chunk's wide strips among themselves.
#include "k2_mfma_frag.h"
#include "dmx_tuning.h"
#include <stdlib.h>

namespace dmx {

static constexpr int BP_SLOT = 16 * 1024;       // one wide strip: 4 K-steps x {re hi, re lo, im hi, im lo} x 64 lanes x 16 B
static constexpr int BP_BUFS = 2;               // chunk buffers: generation of chunk c + 1 beside the products of chunk c
// A workgroup of NW waves owns 32 * NW rows and works through chunks of NW / 4 wide strips (every wave generates one
// K-step of one wide strip).  NW = 8: 2 workgroups per CU, their 4 + 4 waves per SIMD run in two lockstep groups;
// NW = 4: 4 workgroups per CU, every SIMD holds one wave of each - four independent phases, so one workgroup's
// generation / |Y| epilogue overlaps the others' matrix-core work (the B' fragments are generated once per 128 rows then).

struct BeamPowArgs {
    int64_t user_begin;
    int m_rx, ue_mh, n_beams;
    int M;                   // rows (rx, beam)
    int K;
    const int32_t* sc;
    double inv_n;
    const float2* ftab;      // [user_count, n_beams, P]
    const int32_t* fexp;     // [user_count]
    float* out;              // [user_count, n_beams] mean amplitude
    int32_t* best;           // [user_count] argmax_b (first maximum), -1 without paths; may be nullptr
    int adaptive;            // 1 = a weak last K-step may take one product term
};

__host__ __device__ inline size_t beam_pow_lds_bytes(int M, int NW) {
    const size_t nblk = ((size_t)M + 32 * NW - 1) / (32 * NW);                 // per row block: NW waves x 32 partial row sums
    return (size_t)BP_BUFS * (NW / 4) * BP_SLOT + LPAD * (8 + 4 + 4) + 16 + nblk * NW * 32 * 4;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void k2c_beam_power(WsView ws, BeamPowArgs a, int64_t user_count) {
    constexpr int BP_WAVES = NW, BP_CHUNK = NW / 4, MAX_ROWS = 32 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* bbuf = smem;                                                      // [BP_BUFS][BP_CHUNK][BP_SLOT]
    float* bbuf_f = reinterpret_cast<float*>(smem);
    float2* qtab = reinterpret_cast<float2*>(smem + (size_t)BP_BUFS * BP_CHUNK * BP_SLOT);     // [32]
    float* crtab = reinterpret_cast<float*>(qtab + LPAD);
    float* citab = crtab + LPAD;
    float* misc = citab + LPAD;                                                      // [4]
    float* rs = misc + 4;                                          // [nblk][8 waves][32] partial row sums of |Y| (no atomics:
                                                                   // the summation order is fixed, results are reproducible)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: tile / strip assignments stay scalar
    const int col = lane & 31, hh = lane >> 5;
    const int B = a.n_beams, M = a.M, P = ws.P;
    const int nwide = (a.K + 31) >> 5;                              // wide strips: 32 subcarriers each
    const int nblk = (M + MAX_ROWS - 1) / MAX_ROWS;

    for (int64_t ul = blockIdx.x; ul < user_count; ul += gridDim.x) {
        const int64_t u = a.user_begin + ul;
        int n_act = ws.n_keep[u];
        n_act = n_act < LPAD ? n_act : LPAD;
        if (n_act == 0) {                                                            // channel.py:270-271: H = 0
            for (int b = tid; b < B; b += BP_WAVES * 64) a.out[(size_t)ul * B + b] = 0.f;
            if (a.best && tid == 0) a.best[ul] = -1;
            continue;
        }
        const size_t rb = (size_t)u * P;
        // ---- per-user path tables (as stage_item of k2_channel_fd_mfma.hip)
        if (wave == 0) {
            float m = 0.f;
            if (lane < n_act) m = fmaxf(fabsf(ws.c_re[rb + lane]), fabsf(ws.c_im[rb + lane]));
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            // adaptive precision as in k2_channel_fd_mfma.hip (stage_item): weak last K-step -> one product term
            const int l0w = ((n_act - 1) >> 3) << 3;
            float a2 = 0.f;
            if (lane < n_act) a2 = fmaf(ws.c_re[rb + lane], ws.c_re[rb + lane], ws.c_im[rb + lane] * ws.c_im[rb + lane]);
            float m2 = a2, mw2 = lane >= l0w ? a2 : 0.f;
            for (int off = 32; off > 0; off >>= 1) { m2 = fmaxf(m2, __shfl_xor(m2, off)); mw2 = fmaxf(mw2, __shfl_xor(mw2, off)); }
            if (lane == 0) misc[3] = (a.adaptive && l0w >= 8 && mw2 * 4194304.0f <= m2) ? 1.f : 0.f;
            int e;
            (void)frexpf(m, &e);
            const float gs = ldexpf(1.0f, 10 - e);
            if (lane < LPAD) {
                const bool ok = lane < n_act;
                const double q = ok ? (double)ws.dn[rb + lane] * a.inv_n : 0.0;
                const double qh = rint(q * 4096.0) * (1.0 / 4096.0);
                qtab[lane] = make_float2((float)qh, (float)(q - qh));
                crtab[lane] = ok ? ws.c_re[rb + lane] * gs : 0.f;
                citab[lane] = ok ? ws.c_im[rb + lane] * gs : 0.f;
            }
            const int ea = 6 - a.fexp[ul];                      // projected responses scaled so that max |f| is in [32, 64)
            if (lane == 0) { misc[0] = ldexpf(1.0f, e - 10 - ea); misc[1] = ldexpf(1.0f, ea); misc[2] = gs; }
        }
        __syncthreads();
        const float ascale = misc[1];
        const int nfull = ((n_act + 7) >> 3) - (misc[3] != 0.f ? 1 : 0);     // K-steps with all three product terms

        for (int blk = 0; blk < nblk; ++blk) {
            const int row0 = blk * MAX_ROWS;
            const int nrows = (M - row0) < MAX_ROWS ? (M - row0) : MAX_ROWS;
            const int ntiles = (nrows + 31) >> 5;
            const int ntp = ntiles <= 1 ? 1 : (ntiles <= 2 ? 2 : (ntiles <= 4 ? 4 : 8));
            const int tile = wave & (ntp - 1), grp = wave / ntp, ngrp = BP_WAVES / ntp;
            const bool active = tile < ntiles;
            // ---- A' fragments of this wave's tile: A[(rx,b), l] = a_rx[rx,l] f[b,l], element j of K-step s is path
            // 8s + 4h + (j>>1), component j&1
            h8 Ah[4], Al[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { Ah[s] = h8{0, 0, 0, 0, 0, 0, 0, 0}; Al[s] = Ah[s]; }
            if (active) {
                const int r = row0 + (tile << 5) + col;
                const bool pok = r < M;
                const int rx = pok ? r / B : 0, bm = pok ? r - rx * B : 0;
                const double yr = (double)(rx % a.ue_mh), zr = (double)(rx / a.ue_mh);
                const float2* frow = a.ftab + ((size_t)ul * B + bm) * P;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (8 * s < n_act) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int pl = 8 * s + 4 * hh + jj;
                            h2 vh = {(_Float16)0.f, (_Float16)0.f}, vl = vh;
                            if (pok && pl < n_act) {
                                float sn, cs;
                                sincos_rev(frac_rev(yr * ws.rx_y[rb + pl] + zr * ws.rx_z[rb + pl]), sn, cs);
                                const float2 f = frow[pl];
                                split2_f16((cs * f.x - sn * f.y) * ascale, (cs * f.y + sn * f.x) * ascale, vh, vl, ws.neg_one);
                            }
                            Ah[s][2 * jj] = vh[0]; Ah[s][2 * jj + 1] = vh[1];
                            Al[s][2 * jj] = vl[0]; Al[s][2 * jj + 1] = vl[1];
                        }
                    }
                }
            }
            typedef float pv2 __attribute__((ext_vector_type(2)));   // packed fp32: one v_pk_* instruction per two rows
            pv2 rowsum[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) rowsum[i] = pv2{0.f, 0.f};

            // Two buffers of BP_CHUNK wide strips: every wave first generates its share of chunk c + 1 (one K-step of
            // one wide strip: lane = (subcarrier, path group)), then runs its tile against chunk c - generation and
            // matrix-core work of different waves share a phase, ONE barrier per chunk.
            auto generate = [&](int c0, int buf) {
                const int wsi = c0 + (wave >> 2), st = wave & 3;
                const int kidx = (wsi << 5) + col;
                const bool kok = wsi < nwide && kidx < a.K;
                const int kki = kok ? a.sc[kidx] : 0;
                const float kl = (float)(kki & 4095), kf = (float)kki;
                h8* slot = reinterpret_cast<h8*>(bbuf + (size_t)(buf * BP_CHUNK + (wave >> 2)) * BP_SLOT);
                h8 rh = h8{0, 0, 0, 0, 0, 0, 0, 0}, rl = rh, ih = rh, il = rh;
                if (kok && 8 * st < n_act) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int pl = 8 * st + 4 * hh + jj;
                        const float2 q = qtab[pl];
                        const float p1 = q.x * kl;
                        float sn, cs;
                        sincos_rev(fmaf(q.y, kf, p1 - rintf(p1)), sn, cs);
                        const float cr = crtab[pl], ci = citab[pl];
                        h2 vh, vl;
                        split2_f16(cr * cs + ci * sn, ci * cs - cr * sn, vh, vl, ws.neg_one);       // (Re G, Im G), G = c e^{-jx}
                        const unsigned hb = __builtin_bit_cast(unsigned, vh), lb = __builtin_bit_cast(unsigned, vl);
                        const h2 reh = __builtin_bit_cast(h2, hb ^ 0x80000000u), rel = __builtin_bit_cast(h2, lb ^ 0x80000000u);
                        const h2 imh = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(hb, hb, 16));
                        const h2 iml = __builtin_bit_cast(h2, __builtin_amdgcn_alignbit(lb, lb, 16));
                        rh[2 * jj] = reh[0]; rh[2 * jj + 1] = reh[1]; rl[2 * jj] = rel[0]; rl[2 * jj + 1] = rel[1];
                        ih[2 * jj] = imh[0]; ih[2 * jj + 1] = imh[1]; il[2 * jj] = iml[0]; il[2 * jj + 1] = iml[1];
                    }
                }
                slot[(4 * st + 0) * 64 + lane] = rh;
                slot[(4 * st + 1) * 64 + lane] = rl;
                slot[(4 * st + 2) * 64 + lane] = ih;
                slot[(4 * st + 3) * 64 + lane] = il;
            };
            __syncthreads();                                                         // the previous row block has read both buffers
            generate(0, 0);
            __syncthreads();
            int buf = 0;
            for (int c0 = 0; c0 < nwide; c0 += BP_CHUNK, buf ^= 1) {
                if (c0 + BP_CHUNK < nwide) generate(c0 + BP_CHUNK, buf ^ 1);
                if (active) {
                    for (int j = grp; j < BP_CHUNK && c0 + j < nwide; j += ngrp) {
                        const h8* slot = reinterpret_cast<const h8*>(bbuf + (size_t)(buf * BP_CHUNK + j) * BP_SLOT);
                        f16v are, aim;
#pragma unroll
                        for (int i = 0; i < 16; ++i) { are[i] = 0.f; aim[i] = 0.f; }
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            if (8 * s < n_act) {
                                const h8 rh = slot[(4 * s + 0) * 64 + lane], rl = slot[(4 * s + 1) * 64 + lane];
                                const h8 ih = slot[(4 * s + 2) * 64 + lane], il = slot[(4 * s + 3) * 64 + lane];
                                are = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s], rh, are, 0, 0, 0);
                                aim = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s], ih, aim, 0, 0, 0);
                                if (s >= nfull) continue;
                                are = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s], rl, are, 0, 0, 0);
                                aim = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s], il, aim, 0, 0, 0);
                                are = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s], rh, are, 0, 0, 0);
                                aim = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s], ih, aim, 0, 0, 0);
                            }
                        }
                        // |Y| of this lane's subcarrier for the 16 rows it holds: both parts are in this lane
                        DMX_MFMA_RESULT_GUARD2(are, aim);
                        // (vector instructions add to the matrix-core time on this chip, square roots do not:
                        // profiles/r2_mfma_valu_overlap.txt - hence packed mul / fma / add around two v_sqrt_f32)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const pv2 re = {are[2 * i], are[2 * i + 1]}, im = {aim[2 * i], aim[2 * i + 1]};
                            const pv2 p = __builtin_elementwise_fma(re, re, im * im);
                            rowsum[i] += pv2{__builtin_amdgcn_sqrtf(p.x), __builtin_amdgcn_sqrtf(p.y)};
                        }
                    }
                }
                __syncthreads();                       // chunk c + 1 is complete; chunk c's buffer may be overwritten
            }
            if (active) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = (i & 1) ? rowsum[i >> 1].y : rowsum[i >> 1].x;
                    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
                    if (col == 0) rs[((blk * BP_WAVES + wave) << 5) + (i & 3) + 8 * (i >> 2) + 4 * hh] = v;
                }
            }
        }
        __syncthreads();
        const float norm = misc[0] / ((float)a.m_rx * (float)a.K);
        for (int b = tid; b < B; b += BP_WAVES * 64) {
            float t = 0.f;
            for (int rx = 0; rx < a.m_rx; ++rx) {
                const int r = rx * B + b, blk = r / MAX_ROWS, w = r - blk * MAX_ROWS, tile = w >> 5;
                const int nrows = (M - blk * MAX_ROWS) < MAX_ROWS ? (M - blk * MAX_ROWS) : MAX_ROWS;
                const int ntiles = (nrows + 31) >> 5;
                const int ntp = ntiles <= 1 ? 1 : (ntiles <= 2 ? 2 : (ntiles <= 4 ? 4 : 8));
                for (int g = 0; g < BP_WAVES / ntp; ++g) t += rs[((blk * BP_WAVES + g * ntp + tile) << 5) + (w & 31)];
            }
            t *= norm;
            a.out[(size_t)ul * B + b] = t;
            bbuf_f[b] = t;                               // the strip buffer is idle here
        }
        __syncthreads();
        if (a.best && tid == 0) {
            int arg = 0;
            float mx = bbuf_f[0];
            for (int b = 1; b < B; ++b) { if (bbuf_f[b] > mx) { mx = bbuf_f[b]; arg = b; } }
            a.best[ul] = arg;
        }
        __syncthreads();                                  // rs / tables / strip buffer are rewritten by the next user
    }
}

int launch_beam_project(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                        const float2* codebook, int n_beams, void* beam_ws, hipStream_t stream, BeamTabs* tabs);

int launch_beam_power(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                      const float2* codebook, int n_beams, void* beam_ws, float* out_amp, int32_t* out_best, hipStream_t stream) {
    if (user_count == 0 || n_beams == 0) return DMX_OK;
    BeamTabs t;
    int rc = launch_beam_project(prm, ws, user_begin, user_count, codebook, n_beams, beam_ws, stream, &t);
    if (rc) return rc;
    BeamPowArgs a;
    a.user_begin = user_begin;
    a.m_rx = prm.ue_shape[0] * prm.ue_shape[1];
    a.ue_mh = prm.ue_shape[0];
    a.n_beams = n_beams;
    a.M = a.m_rx * n_beams;
    a.K = prm.n_selected;
    a.sc = prm.selected_subcarriers;
    a.inv_n = 1.0 / (double)prm.n_subcarriers;
    a.ftab = t.ftab;
    a.fexp = t.fexp;
    a.out = out_amp;
    a.best = out_best;
    a.adaptive = (prm.flags & DMX_FLAG_ADAPTIVE_TERMS) ? 1 : 0;     // default: three product terms everywhere
    const int nw = tuning_int("DMX_BEAM_WAVES", 8) == 4 ? 4 : 8;    // tuning build only
    const size_t smem = beam_pow_lds_bytes(a.M, nw);
    if (smem > 160 * 1024) { set_error("%d x %d (rx, beam) rows are too many for the beam-power kernel", a.m_rx, n_beams); return DMX_ERR_SHAPE; }
    const void* kfn = nw == 4 ? reinterpret_cast<const void*>(k2c_beam_power<4>) : reinterpret_cast<const void*>(k2c_beam_power<8>);
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, nw * 64, smem) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    int64_t grid = (int64_t)device_cu_count() * per_cu;
    // a few users per workgroup let the dispatcher balance the tail (k2_channel_fd_mfma.hip: ITEMS_PER_WG)
    const int64_t g4 = user_count / 4;
    if (g4 > grid) grid = g4 < 4 * grid ? g4 : 4 * grid;
    if (grid > user_count) grid = user_count;
    if (nw == 4) hipLaunchKernelGGL(k2c_beam_power<4>, dim3((unsigned)grid), dim3(256), smem, stream, ws, a, user_count);
    else hipLaunchKernelGGL(k2c_beam_power<8>, dim3((unsigned)grid), dim3(512), smem, stream, ws, a, user_count);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error("k2c_beam_power launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

}  // namespace dmx
