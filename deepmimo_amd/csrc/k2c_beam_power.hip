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
// registers for the whole user), so the B' fragments of a 32-column strip are needed by every wave: each wave builds
// ONE strip of an 8-strip chunk and parks its fragments in LDS in register layout (8 KB per strip, conflict-free
// ds_read_b128 / ds_write_b128), then every wave runs its tile against the chunk's strips.  With fewer than eight
// row tiles the waves also split the chunk's strips among themselves.
#include "k2_mfma_frag.h"

namespace dmx {

static constexpr int BP_WAVES = 8;
static constexpr int BP_SLOT = 8 * 1024;        // one strip's B' fragments: 4 K-steps x {hi, lo} x 64 lanes x 16 B

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
};

__host__ __device__ inline size_t beam_pow_lds_bytes(int M) {
    const size_t nblk = ((size_t)M + MAX_ROWS - 1) / MAX_ROWS;                 // per row block: 8 waves x 32 partial row sums
    return (size_t)BP_WAVES * BP_SLOT + LPAD * (8 + 4 + 4) + 16 + nblk * BP_WAVES * 32 * 4;
}

// v[i] += (v[i] of the neighbouring lane, lane ^ 1) for the 16 squared accumulator values: one VALU instruction each
// with the DPP operand folded in (quad_perm [1,0,3,2]; the compiler's own form of `v + dpp(v)` is v_mov_b32_dpp + add).
// A VGPR written by a VALU instruction needs two wait states before a DPP read: one s_nop for the whole group, inside
// the statement (the compiler pads nothing in front of an asm string).
#define BP_DPP(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
__device__ __forceinline__ void pair_sums16(float (&v)[16]) {
    asm("s_nop 1\n\t" BP_DPP(0) BP_DPP(1) BP_DPP(2) BP_DPP(3) BP_DPP(4) BP_DPP(5) BP_DPP(6) BP_DPP(7)
        BP_DPP(8) BP_DPP(9) BP_DPP(10) BP_DPP(11) BP_DPP(12) BP_DPP(13) BP_DPP(14) BP_DPP(15)
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}

__global__ __launch_bounds__(BP_WAVES * 64, 4) void k2c_beam_power(WsView ws, BeamPowArgs a, int64_t user_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* bbuf = smem;                                                      // [8][BP_SLOT]
    float* bbuf_f = reinterpret_cast<float*>(smem);
    float2* qtab = reinterpret_cast<float2*>(smem + (size_t)BP_WAVES * BP_SLOT);     // [32]
    float* crtab = reinterpret_cast<float*>(qtab + LPAD);
    float* citab = crtab + LPAD;
    float* misc = citab + LPAD;                                                      // [4]
    float* rs = misc + 4;                                          // [nblk][8 waves][32] partial row sums of |Y| (no atomics:
                                                                   // the summation order is fixed, results are reproducible)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: tile / strip assignments stay scalar
    const int col = lane & 31, hh = lane >> 5;
    const int B = a.n_beams, M = a.M, P = ws.P;
    const size_t twoK = (size_t)2 * a.K;
    const int nstrips = (int)((twoK + 31) >> 5);
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
        const float ascale = misc[1], gscale = misc[2];

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
                                split2_f16((cs * f.x - sn * f.y) * ascale, (cs * f.y + sn * f.x) * ascale, vh, vl);
                            }
                            Ah[s][2 * jj] = vh[0]; Ah[s][2 * jj + 1] = vh[1];
                            Al[s][2 * jj] = vl[0]; Al[s][2 * jj + 1] = vl[1];
                        }
                    }
                }
            }
            float rowsum[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) rowsum[i] = 0.f;

            for (int c0 = 0; c0 < nstrips; c0 += BP_WAVES) {
                __syncthreads();                                                     // the previous chunk has been consumed
                if (c0 + wave < nstrips) {
                    const BLane bl = b_lane(c0 + wave, col, hh, twoK, a.sc);
                    h8* slot = reinterpret_cast<h8*>(bbuf + (size_t)wave * BP_SLOT);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        h8 bh, blo;
                        gen_b_step(s, bl, hh, n_act, qtab, crtab, citab, nullptr, a.K, gscale, bh, blo);
                        if (!bl.kok) { bh = h8{0, 0, 0, 0, 0, 0, 0, 0}; blo = bh; }  // columns past the selection add nothing
                        slot[(2 * s) * 64 + lane] = bh;
                        slot[(2 * s + 1) * 64 + lane] = blo;
                    }
                }
                __syncthreads();
                if (active) {
                    for (int j = grp; j < BP_WAVES && c0 + j < nstrips; j += ngrp) {
                        const h8* slot = reinterpret_cast<const h8*>(bbuf + (size_t)j * BP_SLOT);
                        f16v acc;
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            if (8 * s < n_act) {
                                const h8 bh = slot[(2 * s) * 64 + lane], blo = slot[(2 * s + 1) * 64 + lane];
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s], bh, acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s], blo, acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s], bh, acc, 0, 0, 0);
                            }
                        }
                        // |Y| of the (re, im) lane pair; both lanes of a pair add the same value (halved at the end)
                        float sq[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) sq[i] = acc[i] * acc[i];
                        pair_sums16(sq);
#pragma unroll
                        for (int i = 0; i < 16; ++i) rowsum[i] += __builtin_amdgcn_sqrtf(sq[i]);
                    }
                }
            }
            if (active) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = rowsum[i];
                    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
                    if (col == 0) rs[((blk * BP_WAVES + wave) << 5) + (i & 3) + 8 * (i >> 2) + 4 * hh] = 0.5f * v;
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
    const size_t smem = beam_pow_lds_bytes(a.M);
    if (smem > 160 * 1024) { set_error("%d x %d (rx, beam) rows are too many for the beam-power kernel", a.m_rx, n_beams); return DMX_ERR_SHAPE; }
    const void* kfn = reinterpret_cast<const void*>(k2c_beam_power);
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, BP_WAVES * 64, smem) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    int64_t grid = (int64_t)device_cu_count() * per_cu;
    // a few users per workgroup let the dispatcher balance the tail (k2_channel_fd_mfma.hip: ITEMS_PER_WG)
    const int64_t g4 = user_count / 4;
    if (g4 > grid) grid = g4 < 4 * grid ? g4 : 4 * grid;
    if (grid > user_count) grid = user_count;
    hipLaunchKernelGGL(k2c_beam_power, dim3((unsigned)grid), dim3(BP_WAVES * 64), smem, stream, ws, a, user_count);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error("k2c_beam_power launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

}  // namespace dmx
