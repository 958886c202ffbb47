// Stage 2, frequency domain - split-precision MFMA kernel (variant 2; the automatic choice for M >= 24 rows, K >= 8).
//
// Per user the channel block is a skinny complex GEMM  H[p, k] = sum_l A[p,l] * G[l,k]  with
//   p = (rx, tx) antenna pair (M = M_rx*M_tx rows),  A[p,l] = a_rx[rx,l] * a_tx[tx,l]   (unit modulus)
//   G[l,k] = c_l * exp(-j 2pi dn_l sc_k / N)
// (dataset.py:398-417 + channel.py:192-197, 281-284).  At L = 25 the fp32 form of this product sits
// exactly on the chip's fp32 ridge (25 flop per output byte), so the fp32 vector kernel is
// VALU-bound at ~30 % of HBM.  Here it runs on the f16 matrix cores (16x the fp32 rate) as a REAL
// GEMM whose output is already the interleaved (re, im) float stream of the result tensor:
//   C[p][2k+c] = sum_kk A'[p][kk] * B'[kk][2k+c],   kk = 2l + {0: re, 1: im},
//   A'[p][2l] = Re A, A'[p][2l+1] = Im A;  B'[2l][2k] = Re G, B'[2l+1][2k] = -Im G,
//                                          B'[2l][2k+1] = Im G, B'[2l+1][2k+1] = Re G.
// Precision: every operand x is split x = hi + lo with hi = f16(x), lo = f16(x - hi); three
// exact-product MFMAs (hi*hi + hi*lo + lo*hi) accumulate into ONE fp32 accumulator, dropping only
// lo*lo (2^-22 relative).  Operands are pre-scaled by powers of two (A' by 64, G by the per-user
// 2^e that puts max|c_l| in [512, 1024)) so that even if the matrix core flushed f16 subnormals the
// loss stays below 1e-6 of the user's peak; the epilogue multiplies the exact inverse power of two
// back.  Measured error vs the fp64-accumulating reference: ~2e-6 of the user's peak (tolerance 5e-5).
//
// Mapping (one 1024-thread workgroup = one user x one block of <= 256 antenna pairs):
//   phase 1  all threads: A' hi/lo tiles -> LDS (row = pair, 64 f16 of kk + 16 B pad: the 144-B row
//            stride makes the ds_read_b128 fragment reads bank-conflict-free).
//   phase 2  each wave owns 32-column strips (16 subcarriers, re/im interleaved).  Lane = column:
//            it builds its own B' fragments in registers (lane pairs 2k/2k+1 split the sincos work
//            and swap results with one DPP-style shuffle), then walks the 32-row tiles:
//            8 ds_read_b128 + 12 v_mfma_f32_32x32x16_f16 + 16 buffer_store_dword (nt) per tile.  The
//            32x32 accumulator has its column on the lane, so every store instruction writes two
//            contiguous 128-B row segments of the output - no LDS transpose, no shuffles; the row is
//            a scalar offset of the buffer instruction, so the epilogue is 1 VALU op per store.
// HBM traffic = the 8*M*K output bytes (written once) + ~1 KB of path records per user.
#include "k2_mfma_frag.h"
#include "dmx_tuning.h"
#include <stdlib.h>

namespace dmx {

struct MfmaArgs {
    int64_t user_begin;
    int m_rx, m_tx, ue_mh, bs_mh;
    int M;           // antenna pairs
    int K;           // selected subcarriers
    const int32_t* sc;
    double inv_n;
    int nblk;        // row blocks per user
    int rows;        // LDS rows (multiple of 32)
    // fused TX-codebook projection (k2b_beam_project): rows are (rx, beam) instead of (rx, tx)
    int n_beams;             // 0 = plain channel
    const float2* ftab;      // [user_count, n_beams, P]  f[b,l] = sum_tx F[b,tx] a_tx[tx,l]
    const int32_t* fexp;     // [user_count]              exponent of max |f| per user
    // rx_filter variant: per-path subcarrier gains precomputed by k3_lpf_* instead of generated here
    const float2* gtab;      // [user_count, P, K] or nullptr
    const uint2* gpack;      // the same table in packed f16 hi/lo form (load_b_step_packed) or nullptr
    int adaptive;            // 1 = a weak last K-step may take one product term (stage_item)
    int alias_table;         // tuning build only (DMX_LPF_ALIAS_TABLE=1): every user reads user 0's gains - the table out of L2, to price its HBM
                             // traffic (a probe for default-policy loads: with the nt DMA loads that ship, all workgroups re-fetch the same evict-first lines)
};

// One 32-row tile of one strip: 2*NS ds_read_b128 of A' issued together, 3*NS MFMAs, 16 stores.  NS = K-steps that
// hold a kept path (8 paths per step); a template parameter so that the body is one straight-line block - with the
// steps guarded at run time every ds_read sat alone in its basic block in front of an `s_waitcnt lgkmcnt(0)`.
// 32x32 accumulator: column on the lane, register i is row (i&3) + 8*(i>>2) + 4*(lane>>5).  Output through a
// buffer descriptor over this workgroup's row block: the per-lane part of the address is one 32-bit voffset per
// strip, the row of each store is a scalar soffset, and rows past the block end fall outside num_records and are
// dropped by the hardware range check.
template <bool NT, int NS, bool LW>
__device__ __forceinline__ void mfma_tile(int pt, const unsigned char* Ahi, const unsigned char* Alo, int col, int hh,
                                          const h8 (&Bhi)[4], const h8 (&Blo)[4], const BLane& bl,
                                          __amdgpu_buffer_rsrc_t orsrc, unsigned row_bytes, float oscale) {
    const size_t abase = (size_t)((pt << 5) + col) * ROW_BYTES + (size_t)hh * 16;
    h8 ah[NS], al[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        ah[s] = *reinterpret_cast<const h8*>(Ahi + abase + s * 32);
        al[s] = *reinterpret_cast<const h8*>(Alo + abase + s * 32);
    }
    __builtin_amdgcn_sched_barrier(0);          // keep the reads together in front: the scheduler otherwise sinks each to its use
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], Bhi[s], acc, 0, 0, 0);
        if (LW && s == NS - 1) continue;        // weak last K-step: one product term (stage_item, "last_weak")
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], Blo[s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], Bhi[s], acc, 0, 0, 0);
    }
    DMX_MFMA_RESULT_GUARD(acc);
    if (bl.kok) {
        const unsigned tile_off = (unsigned)(pt << 5) * row_bytes;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned soff = tile_off + (unsigned)((i & 3) + 8 * (i >> 2)) * row_bytes;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i] * oscale), orsrc, bl.lane_off, soff, NT ? 2 : 0);
        }
    }
}

// Same tile with the K-steps guarded at run time and one A' fragment live at a time: 28 fewer live registers.  The
// grouped form above pushes the 128-VGPR kernel into spilling a few long-lived values, reloaded once per work item:
// nothing against eight tiles per strip, +11-15 % on row blocks of one or two tiles (16 / 64 rows x 512 subcarriers:
// 3.6 -> 4.0 and 5.5 -> 6.4 ms per 100k users).  The grouped and pipelined kernels (MODE 1, 2) are therefore only launched for >= 128-row blocks.
template <bool NT>
__device__ __forceinline__ void mfma_tile_rt(int pt, const unsigned char* Ahi, const unsigned char* Alo, int col, int hh, int n_act, int n_full,
                                             const h8 (&Bhi)[4], const h8 (&Blo)[4], const BLane& bl,
                                             __amdgpu_buffer_rsrc_t orsrc, unsigned row_bytes, float oscale) {
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const size_t abase = (size_t)((pt << 5) + col) * ROW_BYTES + (size_t)hh * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (8 * s < n_act) {
            const h8 ah = *reinterpret_cast<const h8*>(Ahi + abase + s * 32);
            const h8 al = *reinterpret_cast<const h8*>(Alo + abase + s * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bhi[s], acc, 0, 0, 0);
            if (s < n_full) {                   // K-steps from n_full on are weak: one product term
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Blo[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bhi[s], acc, 0, 0, 0);
            }
        }
    }
    DMX_MFMA_RESULT_GUARD(acc);
    if (bl.kok) {
        const unsigned tile_off = (unsigned)(pt << 5) * row_bytes;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned soff = tile_off + (unsigned)((i & 3) + 8 * (i >> 2)) * row_bytes;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i] * oscale), orsrc, bl.lane_off, soff, NT ? 2 : 0);
        }
    }
}

// Software-pipelined strip: the matrix-core chain of tile t + 1 is issued with the scale-and-store of tile t threaded
// between its instructions (two accumulators).  A wave otherwise emits its 16 stores in one burst after a ~400-cycle
// chain during which it stores nothing; with 4 waves per SIMD those bursts leave the memory pipeline idle in between.
template <bool NT, int I0, int I1>
__device__ __forceinline__ void store_rows(const f16v& acc, float oscale, const BLane& bl, __amdgpu_buffer_rsrc_t orsrc,
                                           unsigned tile_off, unsigned row_bytes) {
    if (!bl.kok) return;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const unsigned soff = tile_off + (unsigned)((i & 3) + 8 * (i >> 2)) * row_bytes;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[i] * oscale), orsrc, bl.lane_off, soff, NT ? 2 : 0);
    }
}

// chain of tile `pt` into a fresh accumulator while `prev` (tile pt - 1) is stored, a few rows after every K-step
template <bool NT, int NS, bool STORE_PREV, bool LW>
__device__ __forceinline__ f16v chain_tile(int pt, f16v& prev, const unsigned char* Ahi, const unsigned char* Alo, int col, int hh,
                                           const h8 (&Bhi)[4], const h8 (&Blo)[4], const BLane& bl,
                                           __amdgpu_buffer_rsrc_t orsrc, unsigned row_bytes, float oscale) {
    const size_t abase = (size_t)((pt << 5) + col) * ROW_BYTES + (size_t)hh * 16;
    const unsigned prev_off = (unsigned)((pt - 1) << 5) * row_bytes;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    h8 ah = *reinterpret_cast<const h8*>(Ahi + abase);
    h8 al = *reinterpret_cast<const h8*>(Alo + abase);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bhi[s], acc, 0, 0, 0);
        if (!(LW && s == NS - 1)) {            // weak last K-step: one product term
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Blo[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bhi[s], acc, 0, 0, 0);
        }
        if (s + 1 < NS) {                      // next K-step's fragments: the read latency passes behind the stores below
            ah = *reinterpret_cast<const h8*>(Ahi + abase + (s + 1) * 32);
            al = *reinterpret_cast<const h8*>(Alo + abase + (s + 1) * 32);
        }
        if constexpr (STORE_PREV) {
            // 16 rows over NS K-steps
            if constexpr (NS == 4) {
                if (s == 0) store_rows<NT, 0, 4>(prev, oscale, bl, orsrc, prev_off, row_bytes);
                if (s == 1) store_rows<NT, 4, 8>(prev, oscale, bl, orsrc, prev_off, row_bytes);
                if (s == 2) store_rows<NT, 8, 12>(prev, oscale, bl, orsrc, prev_off, row_bytes);
                if (s == 3) store_rows<NT, 12, 16>(prev, oscale, bl, orsrc, prev_off, row_bytes);
            } else if constexpr (NS == 3) {
                if (s == 0) store_rows<NT, 0, 6>(prev, oscale, bl, orsrc, prev_off, row_bytes);
                if (s == 1) store_rows<NT, 6, 11>(prev, oscale, bl, orsrc, prev_off, row_bytes);
                if (s == 2) store_rows<NT, 11, 16>(prev, oscale, bl, orsrc, prev_off, row_bytes);
            } else if constexpr (NS == 2) {
                if (s == 0) store_rows<NT, 0, 8>(prev, oscale, bl, orsrc, prev_off, row_bytes);
                if (s == 1) store_rows<NT, 8, 16>(prev, oscale, bl, orsrc, prev_off, row_bytes);
            } else {
                store_rows<NT, 0, 16>(prev, oscale, bl, orsrc, prev_off, row_bytes);
            }
            __builtin_amdgcn_sched_barrier(0);       // keep this K-step's stores behind its MFMAs and in front of the next reads
        }
    }
    // right behind the chain, whoever reads the accumulator next (the next tile's threaded stores, the strip's last
    // stores, or a register copy where two paths of strip_tiles meet)
    DMX_MFMA_RESULT_GUARD(acc);
    return acc;
}

template <bool NT, int NS, bool PIPE, bool LW>
__device__ __forceinline__ void strip_tiles(int ntiles, const unsigned char* Ahi, const unsigned char* Alo, int col, int hh,
                                            const h8 (&Bhi)[4], const h8 (&Blo)[4], const BLane& bl,
                                            __amdgpu_buffer_rsrc_t orsrc, unsigned row_bytes, float oscale) {
    if constexpr (!PIPE) {
        for (int pt = 0; pt < ntiles; ++pt)
            mfma_tile<NT, NS, LW>(pt, Ahi, Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale);
        return;
    }
    f16v a0 = {};
    a0 = chain_tile<NT, NS, false, LW>(0, a0, Ahi, Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale);
    int pt = 1;
    for (; pt + 1 < ntiles; pt += 2) {                                   // a0 holds tile pt - 1
        f16v a1 = chain_tile<NT, NS, true, LW>(pt, a0, Ahi, Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale);
        a0 = chain_tile<NT, NS, true, LW>(pt + 1, a1, Ahi, Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale);
    }
    if (pt < ntiles) {
        f16v a1 = chain_tile<NT, NS, true, LW>(pt, a0, Ahi, Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale);
        store_rows<NT, 0, 16>(a1, oscale, bl, orsrc, (unsigned)(pt << 5) * row_bytes, row_bytes);   // stored right behind its chain
    } else {
        store_rows<NT, 0, 16>(a0, oscale, bl, orsrc, (unsigned)((ntiles - 1) << 5) * row_bytes, row_bytes);
    }
}

// What one (user, row block) work item keeps in LDS: the A' hi / lo tiles and the per-user path tables.
struct ItemLds {
    unsigned char* Ahi;      // [rows][144 B]
    unsigned char* Alo;
    float2* qtab;            // [32] dn_l / N as (multiple of 2^-12, remainder)
    float* crtab;            // [32] scaled c_l
    float* citab;
    float* misc;             // [4]  per-user output / operand scales
};
__host__ __device__ inline size_t item_lds_bytes(int rows) { return (size_t)2 * rows * ROW_BYTES + LPAD * (8 + 4 + 4) + 16; }
__device__ __forceinline__ ItemLds item_lds(unsigned char* base, int rows) {
    ItemLds L;
    L.Ahi = base;
    L.Alo = base + (size_t)rows * ROW_BYTES;
    L.qtab = reinterpret_cast<float2*>(base + (size_t)2 * rows * ROW_BYTES);
    L.crtab = reinterpret_cast<float*>(L.qtab + LPAD);
    L.citab = L.crtab + LPAD;
    L.misc = L.citab + LPAD;
    return L;
}

struct ItemPos {
    int64_t ul, u;
    int row0, nrows, n_act;
    size_t rb;
};
__device__ __forceinline__ ItemPos item_pos(const WsView& ws, const MfmaArgs& a, int64_t work) {
    ItemPos p;
    p.ul = work / a.nblk;
    const int blk = (int)(work % a.nblk);
    p.u = a.user_begin + p.ul;
    p.row0 = blk * MAX_ROWS;
    p.nrows = (a.M - p.row0) < MAX_ROWS ? (a.M - p.row0) : MAX_ROWS;     // valid rows of this block
    const int n = ws.n_keep[p.u];
    p.n_act = n < LPAD ? n : LPAD;
    p.rb = (size_t)p.u * ws.P;
    return p;
}

// Stage 1 of a work item: the per-user tables (wave 0) and the A' tiles (thread = antenna pair x slice of the path
// slots) into `L`.  No barrier inside; nothing here reads what another thread of this call wrote.
// NW = waves per workgroup (4, 8 or 16).
template <int NW>
__device__ __forceinline__ void stage_item(const WsView& ws, const MfmaArgs& a, int64_t work, const ItemLds& L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ItemPos ip = item_pos(ws, a, work);
    const int n_act = ip.n_act;
    if (n_act == 0) return;                                              // consume_item writes zeros, reads nothing
    // uniform per-item bases and 32-bit path indices: indexed as array[row base + l] the compiler keeps eight 64-bit
    // per-lane offsets alive across the whole item loop (16 VGPRs the tile loops need)
    const float* __restrict__ c_re = ws.c_re + ip.rb;
    const float* __restrict__ c_im = ws.c_im + ip.rb;
    const float* __restrict__ dnp = ws.dn + ip.rb;
    const double* __restrict__ rx_y = ws.rx_y + ip.rb;
    const double* __restrict__ rx_z = ws.rx_z + ip.rb;
    const double* __restrict__ tx_y = ws.tx_y + ip.rb;
    const double* __restrict__ tx_z = ws.tx_z + ip.rb;

    // per-user power-of-two scale: max |c_l| component -> [512, 1024)
    if (wave == 0) {
        float m = 0.f;
        if (lane < n_act) m = fmaxf(fabsf(c_re[lane]), fabsf(c_im[lane]));
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        // Adaptive precision: when every path of the LAST K-step (stage 1 orders the kept paths by falling amplitude) is
        // at least 66 dB below the user's strongest one - |c| <= 2^-11 max |c| - that K-step's products are taken as
        // A'hi B'hi alone: each then carries a relative error of at most 2^-9 (both factors truncated to f16) instead of
        // 2^-21, i.e. at most 8 paths x 2^-20 = 7.6e-6 of the strongest path's amplitude in the worst case (1e-6 rms), next
        // to the 2e-6 the 3-term products leave and the 5e-5 tolerance (test_adaptive_precision_weak_tail_worst_case).
        // Saves 2 of the 12 MFMAs per tile whenever the last K-step is weak (25 kept paths: it holds the weakest one).
        const int l0w = ((n_act - 1) >> 3) << 3;
        float a2 = 0.f;
        if (lane < n_act) a2 = fmaf(c_re[lane], c_re[lane], c_im[lane] * c_im[lane]);
        float m2 = a2, mw2 = lane >= l0w ? a2 : 0.f;
        for (int off = 32; off > 0; off >>= 1) { m2 = fmaxf(m2, __shfl_xor(m2, off)); mw2 = fmaxf(mw2, __shfl_xor(mw2, off)); }
        const bool last_weak = a.adaptive && l0w >= 8 && mw2 * 4194304.0f <= m2;    // |c|^2 ratio 2^-22
        int e;
        (void)frexpf(m, &e);                                             // m = f * 2^e, f in [0.5, 1)
        const float gs = ldexpf(1.0f, 10 - e);
        if (lane == 0) L.misc[3] = last_weak ? 1.f : 0.f;
        if (lane < LPAD) {
            const bool ok = lane < n_act;
            const double q = ok ? (double)dnp[lane] * a.inv_n : 0.0;
            const double qh = rint(q * 4096.0) * (1.0 / 4096.0);
            L.qtab[lane] = make_float2((float)qh, (float)(q - qh));
            L.crtab[lane] = ok ? c_re[lane] * gs : 0.f;
            L.citab[lane] = ok ? c_im[lane] * gs : 0.f;
        }
        // A' scale: 64 for unit-modulus array responses; with a codebook the projected responses f are
        // scaled per user so that max |f| lands in [32, 64)
        const int ea = a.n_beams ? 6 - a.fexp[ip.ul] : 6;
        // rx_filter: |G| can exceed |c| by the sinc sum (a few x); two bits of headroom keep G*gs < 2^13
        if (a.gtab || a.gpack) e += 2;
        if (lane == 0) { L.misc[0] = ldexpf(1.0f, e - 10 - ea); L.misc[1] = ldexpf(1.0f, ea); L.misc[2] = ldexpf(1.0f, 10 - e); }
    }

    constexpr int LSPLIT = NW / 4, LPER = LPAD / LSPLIT;
    const int r = tid & 255, l0 = (tid >> 8) * LPER;
    if (r >= a.rows) return;
    const int p = ip.row0 + r;
    const bool pok = p < a.M;
    const int ncolA = a.n_beams ? a.n_beams : a.m_tx;                    // second index of the row pair
    const int rx = pok ? p / ncolA : 0, tx = pok ? p - rx * ncolA : 0;
    const double yr = (double)(rx % a.ue_mh), zr = (double)(rx / a.ue_mh);
    const double yt = (double)(tx % a.bs_mh), zt = (double)(tx / a.bs_mh);
    h2* rhi = reinterpret_cast<h2*>(L.Ahi + (size_t)r * ROW_BYTES);
    h2* rlo = reinterpret_cast<h2*>(L.Alo + (size_t)r * ROW_BYTES);
    if (a.n_beams) {
        // beam-space rows: A[(rx,b), l] = a_rx[rx,l] * f[b,l]  (f from k2b_beam_project)
        const float2* frow = a.ftab + ((size_t)ip.ul * a.n_beams + tx) * ws.P;
        const float ascale = ldexpf(1.0f, 6 - a.fexp[ip.ul]);
        for (int l = l0; l < l0 + LPER; ++l) {
            h2 vh = {(_Float16)0.f, (_Float16)0.f}, vl = vh;
            if (pok && l < n_act) {
                float s, c;
                sincos_rev(frac_rev(yr * rx_y[l] + zr * rx_z[l]), s, c);
                const float2 f = frow[l];
                split2_f16((c * f.x - s * f.y) * ascale, (c * f.y + s * f.x) * ascale, vh, vl, ws.neg_one);
            }
            rhi[l] = vh;
            rlo[l] = vl;
        }
    } else {
        for (int l = l0; l < l0 + LPER; ++l) {
            h2 vh = {(_Float16)0.f, (_Float16)0.f}, vl = vh;
            if (pok && l < n_act) {
                const double ph = yr * rx_y[l] + zr * rx_z[l] + yt * tx_y[l] + zt * tx_z[l];
                float s, c;
                sincos_rev(frac_rev(ph), s, c);
                split2_f16(c * A_SCALE, s * A_SCALE, vh, vl, ws.neg_one);
            }
            rhi[l] = vh;
            rlo[l] = vl;
        }
    }
}

// ---- rx_filter path, GSRC = 3: the packed gains of a strip through LDS-DMA ---------------------------------------------
// On gfx950 loads, stores and LDS-DMA share ONE in-order vmcnt: a wave that loads its next strip's gains (GSRC = 2) waits
// behind the 128 stores it has just issued.  A load ISSUED IN FRONT of those stores does not wait for them, but 32 more
// live registers spill ("not kept").  LDS-DMA needs no registers: each wave owns a 4-KiB slot [32 path rows]
// [16 subcarriers] of packed entries; right after it has read a strip's fragments out of the slot it requests the next
// strip's (or the next work item's first strip's) entries into it - 4 x `global_load_lds_dwordx4`, lane = (row 8j +
// lane/8, 16-byte piece lane%8) - and only then starts the tiles and their stores.  At the next strip `s_waitcnt
// vmcnt(N)` with N = the stores issued since (<= 63) retires the DMA without draining the stores.
//   * The slots are a static __shared__ array of their own: the compiler then knows that the A' fragment reads (dynamic
//     LDS) do not alias the DMA's destination and puts no vmcnt wait in front of them.
//   * The slot itself is read by inline asm (two statements of 4 x ds_read2_b64 + s_waitcnt lgkmcnt(0)): a visible read
//     of the DMA's destination would get the compiler's conservative vmcnt wait - the drain this form exists to avoid.
//     Their results pass through ordinary vector instructions (select, rotate) before any MFMA reads them, and they
//     stand behind the previous strip's last 16 stores: nothing of them is near a matrix-core instruction.
//   * Only for K a multiple of 16 (every strip full, every piece inside its row, 16-byte aligned): other K take GSRC = 2.
// What it buys, priced with every user reading user 0's rows (tuning build, DMX_LPF_ALIAS_TABLE=1: the table out of
// L2; headline shape x 100k users, contraction alone = total - 1.86 ms of FFT): register loads 19.4 ms with the table in
// HBM, 17.4 out of L2; this form 18.8 and 17.1 (the plain kernel, 16 waves: 16.8).  I.e. the queueing was 0.5 ms of the
// 3.6 the table-fed contraction loses on the plain one; 1.8-2 ms are the 10 GB of table coming out of HBM between the
// stores, which only a table that never leaves the CU would remove.
static constexpr int DMA_SLOT_BYTES = 4096;

__device__ __forceinline__ void dma_strip(const uint2* __restrict__ prow, int K, int kidx0, int nrows, unsigned char* slot, int lane) {
    const int sub = lane >> 3, piece = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * j + sub;
        if (row < nrows) {
            const uint2* g = prow + (size_t)row * K + kidx0 + 2 * piece;
            // aux = 2 (nt): the table is read once - same box, 100k users: 20.72 -> 20.42 ms for FFT + contraction
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(slot + j * 1024), 16, 0, 2);
        }
    }
}

typedef unsigned u4v __attribute__((ext_vector_type(4)));
// fragments of one strip out of the wave's slot: entry (path l, subcarrier c) at l * 128 + c * 8
__device__ __forceinline__ void read_b_slot(const unsigned char* slot, int col, int hh, int c, int n_act, h8 (&Bhi)[4], h8 (&Blo)[4]) {
    const uint32_t a0 = (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)slot + (uint32_t)hh * 512u + (uint32_t)(col >> 1) * 8u;
    const uint32_t a1 = a0 + 2048u;
    const unsigned rot = c ? 16u : 0u, sgn = c ? 0u : 0x80000000u;
    // two K-steps per statement: 16 transient registers instead of 32
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        u4v g[4];                                                       // g[2 (s & 1) + (jj >> 1)] = entries jj, jj + 1 of K-step s
        asm volatile(
            "ds_read2_b64 %0, %4 offset1:16\n\t"
            "ds_read2_b64 %1, %4 offset0:32 offset1:48\n\t"
            "ds_read2_b64 %2, %4 offset0:128 offset1:144\n\t"
            "ds_read2_b64 %3, %4 offset0:160 offset1:176\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]) : "v"(half ? a1 : a0) : "memory");
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int s = 2 * half + s2;
            Bhi[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            Blo[s] = Bhi[s];
            if (8 * s >= n_act) continue;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool ok = 8 * s + 4 * hh + jj < n_act;             // rows from n_act on were not requested: stale bytes
                const unsigned gx = ok ? g[2 * s2 + (jj >> 1)][2 * (jj & 1)] : 0u, gy = ok ? g[2 * s2 + (jj >> 1)][2 * (jj & 1) + 1] : 0u;
                const unsigned xh = __builtin_amdgcn_alignbit(gx, gx, rot) ^ sgn;
                const unsigned xl = __builtin_amdgcn_alignbit(gy, gy, rot) ^ sgn;
                const h2 ph = __builtin_bit_cast(h2, xh), pl2 = __builtin_bit_cast(h2, xl);
                Bhi[s][2 * jj] = ph[0]; Bhi[s][2 * jj + 1] = ph[1];
                Blo[s][2 * jj] = pl2[0]; Blo[s][2 * jj + 1] = pl2[1];
            }
        }
    }
}

// retire the DMA that was issued in front of the last `ntiles` x 16 stores of this wave (in-order vmcnt; at most 63 countable)
__device__ __forceinline__ void wait_dma_behind_stores(int ntiles) {
    if (ntiles >= 4) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else if (ntiles == 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (ntiles == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (ntiles == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Stage 2 of a work item, after a barrier behind stage_item: a wave owns one 32-column strip at a time (B'
// fragments in registers) and walks the row tiles with it, so the waves of the workgroup fill one 32-row band of the
// user's block together and their stores stay within a few DRAM pages.  No barrier inside.
// GSRC: where the subcarrier gains G[l,k] come from - 0 generated here (plain path), 1 float table, 2 packed f16 table
// (rx_filter path).  A template parameter: the plain kernel carries none of the table code.
// GSRC = 3: `prefetched` says that this wave's slot already holds (or is receiving) the item's first strip; the return
// value says the same for the next item.
template <bool NT, int NW, int MODE, int GSRC>
__device__ __forceinline__ bool consume_item(const WsView& ws, const MfmaArgs& a, float* __restrict__ out, int64_t work,
                                             int64_t next_work, const ItemLds& L, bool prefetched) {
    constexpr int NTHR = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ItemPos ip = item_pos(ws, a, work);
    const size_t twoK = (size_t)2 * a.K;
    float* __restrict__ o = out + ((size_t)ip.ul * a.M + ip.row0) * twoK;
    const int n_act = ip.n_act;
    if (n_act == 0) {                                                    // channel.py:270-271
        const size_t nel = (size_t)ip.nrows * twoK;
        for (size_t i = tid; i < nel; i += NTHR) o[i] = 0.f;
        return false;
    }
    const float oscale = L.misc[0];
    const float gscale = L.misc[2];
    const bool last_weak = L.misc[3] != 0.f;                             // workgroup-uniform
    const float2* grow = GSRC == 1 ? a.gtab + (size_t)ip.ul * ws.P * a.K : nullptr;
    const int64_t tul = a.alias_table ? 0 : ip.ul;                       // user whose rows of the gains table are read
    const uint2* prow = GSRC == 2 ? a.gpack + (size_t)tul * ws.P * a.K : nullptr;
    (void)prefetched;

    const int col = lane & 31, hh = lane >> 5;
    const unsigned row_bytes = (unsigned)twoK * 4u;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(o, 0, (int)((unsigned)ip.nrows * row_bytes), 0x00020000);
    const int nstrips = (int)((twoK + 31) >> 5);
    const int ntiles = (ip.nrows + 31) >> 5;
    unsigned touched = 0;
    bool next_prefetched = false;
    unsigned char* slot = nullptr;
    int nn_next = 0;
    if constexpr (GSRC == 3) {
        __shared__ __attribute__((aligned(16))) unsigned char dma_slots[NW * DMA_SLOT_BYTES];
        slot = dma_slots + wave * DMA_SLOT_BYTES;
        // path count of the next item's user, fetched HERE (behind the barrier, nothing in flight): a load result first
        // used between the strips' stores would make the compiler drain them
        if (next_work >= 0) {
            const int nn = __builtin_amdgcn_readfirstlane(ws.n_keep[a.user_begin + next_work / a.nblk]);
            nn_next = nn < LPAD ? nn : LPAD;
        }
        if (wave < nstrips) {
            if (!prefetched) dma_strip(a.gpack + (size_t)tul * ws.P * a.K, a.K, wave * 16, n_act, slot, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // behind the item's barrier nothing else is in flight
        }
    }
    for (int strip = wave; strip < nstrips; strip += NW) {
        const BLane bl = b_lane(strip, col, hh, twoK, a.sc);
        h8 Bhi[4], Blo[4];
        if constexpr (GSRC == 3) {
            const uint2* urow = a.gpack + (size_t)tul * ws.P * a.K;
            if (strip != wave) wait_dma_behind_stores(ntiles);
            read_b_slot(slot, col, hh, bl.c, n_act, Bhi, Blo);
            const int ns = strip + NW;
            if (ns < nstrips) {
                dma_strip(urow, a.K, ns * 16, n_act, slot, lane);
            } else if (next_work >= 0) {                                 // the next item's first strip of this wave
                const int64_t nul = a.alias_table ? 0 : next_work / a.nblk;
                dma_strip(a.gpack + (size_t)nul * ws.P * a.K, a.K, wave * 16, nn_next, slot, lane);
                next_prefetched = true;
            }
        } else {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                if constexpr (GSRC == 2) load_b_step_packed(st, bl, hh, n_act, prow, a.K, Bhi[st], Blo[st]);
                else gen_b_step(ws.neg_one, st, bl, hh, n_act, L.qtab, L.crtab, L.citab, grow, a.K, gscale, Bhi[st], Blo[st]);
            }
        }
        if constexpr (GSRC == 2) {
            // The table reads are cold HBM misses in front of a dependent MFMA chain (with every user aliased to one
            // table - L2 hits - this path is 1.3 ms per 20k users faster).  So each wave TOUCHES the lines of the strip
            // it will load next (32 path rows x one 128-B line: one dword load by 32 lanes), one strip ahead - the
            // next item's first strip after its last one - and finds them in L2 when it gets there.  The value is
            // consumed by an empty asm behind this strip's fragments, so the compiler keeps the load and its wait.
            asm volatile("" :: "v"(touched), "v"(__builtin_bit_cast(unsigned, h2{Bhi[0][0], Bhi[0][1]})));
            const int ns = strip + NW;
            const uint2* nrow = nullptr;
            const int kmax = a.K - 1;                                     // stay inside the row (K need not be a multiple of 16)
            if (ns < nstrips) nrow = prow + (size_t)col * a.K + (size_t)(ns * 16 < kmax ? ns * 16 : kmax);
            else if (next_work >= 0) nrow = a.gpack + (size_t)(next_work / a.nblk) * ws.P * a.K + (size_t)col * a.K + (size_t)(wave * 16 < kmax ? wave * 16 : kmax);
            touched = 0;
            if (nrow && hh == 0 && col < ws.P) touched = *reinterpret_cast<const volatile unsigned*>(nrow);
        }
        if constexpr (MODE != 0) {
            constexpr bool PIPE = MODE == 2;
            switch (((n_act + 7) >> 3) + (last_weak ? 4 : 0)) {           // workgroup-uniform
                case 1: strip_tiles<NT, 1, PIPE, false>(ntiles, L.Ahi, L.Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale); break;
                case 2: strip_tiles<NT, 2, PIPE, false>(ntiles, L.Ahi, L.Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale); break;
                case 3: strip_tiles<NT, 3, PIPE, false>(ntiles, L.Ahi, L.Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale); break;
                case 4: strip_tiles<NT, 4, PIPE, false>(ntiles, L.Ahi, L.Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale); break;
                case 6: strip_tiles<NT, 2, PIPE, true>(ntiles, L.Ahi, L.Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale); break;
                case 7: strip_tiles<NT, 3, PIPE, true>(ntiles, L.Ahi, L.Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale); break;
                default: strip_tiles<NT, 4, PIPE, true>(ntiles, L.Ahi, L.Alo, col, hh, Bhi, Blo, bl, orsrc, row_bytes, oscale); break;
            }
        } else {
            const int n_full = ((n_act + 7) >> 3) - (last_weak ? 1 : 0);
            for (int pt = 0; pt < ntiles; ++pt)
                mfma_tile_rt<NT>(pt, L.Ahi, L.Alo, col, hh, n_act, n_full, Bhi, Blo, bl, orsrc, row_bytes, oscale);
        }
    }
    return next_prefetched;
}

// One (user, row block) per loop iteration.  Launched with one workgroup per work item, or persistently (grid =
// what is resident at once, workgroups stride over the work items).
template <bool NT, int NW, int MODE, int GSRC>
__global__ __launch_bounds__(NW * 64, 4) void k2_fd_mfma(WsView ws, MfmaArgs a, float* __restrict__ out, int64_t total) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ItemLds L = item_lds(smem, a.rows);
    bool prefetched = false;                                             // GSRC = 3, per wave: see consume_item
    for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
        stage_item<NW>(ws, a, w, L);
        __syncthreads();
        prefetched = consume_item<NT, NW, MODE, GSRC>(ws, a, out, w, (w + gridDim.x < total) ? w + gridDim.x : (int64_t)-1, L, prefetched);
        __syncthreads();                                                 // the next item's tiles overwrite these
    }
}

// Fused consumer of H (docs/manual.ipynb cell 105: `F1 @ dataset.channel`): for a TX
// codebook F [B, M_tx] the beam-space channel Y[u,rx,b,k] = sum_tx F[b,tx] H[u,rx,tx,k] is the SAME
// contraction with the transmit array response replaced by its projection f[b,l] = sum_tx F[b,tx] a_tx[tx,l].
// This kernel builds f per user (a_tx table in LDS, B*L*M_tx complex MACs: ~3 % of the contraction) and the
// power-of-two exponent of max|f| that k2_fd_mfma uses to scale its f16 operands.  The [N,M_rx,M_tx,K]
// tensor is never written: output (and HBM traffic) shrinks by M_tx / B.
struct BeamArgs {
    int64_t user_begin;
    int m_tx, bs_mh, n_beams;
    const float2* F;         // [n_beams, m_tx] complex64, row-major
    float2* ftab;            // [user_count, n_beams, P]
    int32_t* fexp;           // [user_count]
};

__global__ __launch_bounds__(256) void k2b_beam_project(WsView ws, BeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2* atx = reinterpret_cast<float2*>(smem);                       // [m_tx][P]
    __shared__ float wmax[4];
    const int tid = threadIdx.x;
    const int64_t ul = blockIdx.x, u = a.user_begin + ul;
    const int P = ws.P;
    int n_act = ws.n_keep[u];
    n_act = n_act < LPAD ? n_act : LPAD;
    const size_t rb = (size_t)u * P;
    for (int i = tid; i < a.m_tx * P; i += 256) {
        const int t = i / P, l = i - t * P;
        float s = 0.f, c = 0.f;
        if (l < n_act) {
            const int y = t % a.bs_mh, z = t / a.bs_mh;
            sincos_rev(frac_rev((double)y * ws.tx_y[rb + l] + (double)z * ws.tx_z[rb + l]), s, c);
        }
        atx[i] = make_float2(c, s);
    }
    __syncthreads();
    float m = 0.f;
    for (int i = tid; i < a.n_beams * P; i += 256) {
        const int b = i / P, l = i - b * P;
        float fr = 0.f, fi = 0.f;
        if (l < n_act) {
            const float2* Frow = a.F + (size_t)b * a.m_tx;
            for (int t = 0; t < a.m_tx; ++t) {
                const float2 w = Frow[t], v = atx[t * P + l];
                fr = fmaf(w.x, v.x, fr); fr = fmaf(-w.y, v.y, fr);
                fi = fmaf(w.x, v.y, fi); fi = fmaf(w.y, v.x, fi);
            }
        }
        a.ftab[((size_t)ul * a.n_beams + b) * P + l] = make_float2(fr, fi);
        m = fmaxf(m, fmaxf(fabsf(fr), fabsf(fi)));
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        int e = 0;
        if (m > 0.f) (void)frexpf(m, &e);
        a.fexp[ul] = e;
    }
}

// Matrix-core form of the projection (used when the codebook fits LDS: B <= 128 beams, B*M_tx <= ~7.5k):
// per user the [B x M_tx] . [M_tx x L] complex product is the same interleaved real GEMM as the main
// contraction, f'[b][2l+c] = sum_kk F'[b][kk] T'[kk][2l+c] with kk = 2tx + {re, im}.  The codebook's hi/lo f16
// split (scaled so max|F| is in [512, 1024)) is staged ONCE per persistent workgroup; each wave then takes
// users in turn, builds the a_tx B'-fragments of one K-step in registers (lane = (path, re/im) column, lane
// pairs share the sin/cos work as in gen_b_fragments) and issues 3 MFMAs per 32-beam tile and K-step:
// 96 MFMAs + ~130 sin/cos pairs per user at 64 beams x 64 elements, instead of 1e5 scalar complex MACs.
template <int NBT>
__global__ __launch_bounds__(256, 2) void k2b_beam_project_mfma(WsView ws, BeamArgs a, int64_t user_count, int kkpad, int fstride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Fhi = smem;                                              // [NBT*32][fstride]
    unsigned char* Flo = smem + (size_t)NBT * 32 * fstride;
    __shared__ float wmax_s[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.n_beams, P = ws.P;

    // codebook -> LDS (hi/lo f16), scaled by the power of two that puts max |F| component in [512, 1024)
    float m = 0.f;
    for (int i = tid; i < B * a.m_tx; i += 256) { const float2 w = a.F[i]; m = fmaxf(m, fmaxf(fabsf(w.x), fabsf(w.y))); }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (lane == 0) wmax_s[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax_s[0], wmax_s[1]), fmaxf(wmax_s[2], wmax_s[3]));
    int eF = 0;
    if (m > 0.f) (void)frexpf(m, &eF);
    const float sF = ldexpf(1.0f, 10 - eF);
    const float oscale = ldexpf(1.0f, eF - 10 - 6);                          // 1 / (sF * A_SCALE)
    const int half_kk = kkpad >> 1;
    for (int i = tid; i < NBT * 32 * half_kk; i += 256) {
        const int b = i / half_kk, t = i - b * half_kk;
        h2 vh = {(_Float16)0.f, (_Float16)0.f}, vl = vh;
        if (b < B && t < a.m_tx) {
            const float2 w = a.F[(size_t)b * a.m_tx + t];
            split2_f16(w.x * sF, w.y * sF, vh, vl, ws.neg_one);
        }
        reinterpret_cast<h2*>(Fhi + (size_t)b * fstride)[t] = vh;
        reinterpret_cast<h2*>(Flo + (size_t)b * fstride)[t] = vl;
    }
    __syncthreads();

    const int colr = lane & 31, hh = lane >> 5, c = lane & 1;
    const int ksteps = kkpad >> 4;
    for (int64_t ul = (int64_t)blockIdx.x * 4 + wave; ul < user_count; ul += (int64_t)gridDim.x * 4) {
        const int64_t u = a.user_begin + ul;
        int n_act = ws.n_keep[u];
        n_act = n_act < LPAD ? n_act : LPAD;
        const size_t rb = (size_t)u * P;
        float* fout = reinterpret_cast<float*>(a.ftab + (size_t)ul * B * P);     // [B][P] complex = [B][2P] floats
        float wmax = 0.f;
        const int nct = (2 * n_act + 31) >> 5;
        for (int ct = 0; ct < nct; ++ct) {
            const int col = (ct << 5) + colr, l = col >> 1;
            const bool lok = l < n_act;
            const double ty = lok ? ws.tx_y[rb + l] : 0.0, tz = lok ? ws.tx_z[rb + l] : 0.0;
            f16v acc[NBT];
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[bt][i] = 0.f;
            for (int s = 0; s < ksteps; ++s) {
                // B' fragment of this K-step: rows kk = 16s + 8h + j  <->  element tx = 8s + 4h + (j>>1), part j&1
                float mr[2], mi[2], orr[2], oi[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int tx = 8 * s + 4 * hh + 2 * c + t;
                    float sn = 0.f, cs = 0.f;
                    if (lok && tx < a.m_tx) {
                        sincos_rev(frac_rev((double)(tx % a.bs_mh) * ty + (double)(tx / a.bs_mh) * tz), sn, cs);
                        sn *= A_SCALE; cs *= A_SCALE;
                    }
                    mr[t] = cs; mi[t] = sn;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) { orr[t] = __shfl_xor(mr[t], 1); oi[t] = __shfl_xor(mi[t], 1); }
                float gr[4], gi[4];
                gr[0] = c ? orr[0] : mr[0]; gi[0] = c ? oi[0] : mi[0];
                gr[1] = c ? orr[1] : mr[1]; gi[1] = c ? oi[1] : mi[1];
                gr[2] = c ? mr[0] : orr[0]; gi[2] = c ? mi[0] : oi[0];
                gr[3] = c ? mr[1] : orr[1]; gi[3] = c ? mi[1] : oi[1];
                h8 Bh, Bl;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float e0 = c ? gi[jj] : gr[jj];
                    const float e1 = c ? gr[jj] : -gi[jj];
                    h2 ph, pl2;
                    split2_f16(e0, e1, ph, pl2, ws.neg_one);
                    Bh[2 * jj] = ph[0]; Bh[2 * jj + 1] = ph[1];
                    Bl[2 * jj] = pl2[0]; Bl[2 * jj + 1] = pl2[1];
                }
#pragma unroll
                for (int bt = 0; bt < NBT; ++bt) {
                    const size_t aoff = (size_t)((bt << 5) + colr) * fstride + (size_t)s * 32 + (size_t)hh * 16;
                    const h8 ah = *reinterpret_cast<const h8*>(Fhi + aoff);
                    const h8 al = *reinterpret_cast<const h8*>(Flo + aoff);
                    acc[bt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bh, acc[bt], 0, 0, 0);
                    acc[bt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bl, acc[bt], 0, 0, 0);
                    acc[bt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bh, acc[bt], 0, 0, 0);
                }
            }
            DMX_MFMA_RESULT_GUARD_FENCED();
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int b = (bt << 5) + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    const float v = acc[bt][i] * oscale;
                    if (lok && b < B) { fout[(size_t)b * 2 * P + col] = v; wmax = fmaxf(wmax, fabsf(v)); }
                }
        }
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
        if (lane == 0) {
            int e = 0;
            if (wmax > 0.f) (void)frexpf(wmax, &e);
            a.fexp[ul] = e;
        }
    }
}

bool fd_mfma_supported(const dmx_params& prm, const WsView& ws) {
    (void)ws;               // any path count: this kernel takes the first 32 kept paths, k2_channel_fd.hip adds the rest
    return prm.n_selected >= 1;
}

// Automatic choice against the subcarrier-per-lane vector kernel (few subcarriers go to the small-output kernel first,
// k2_channel_fd_small.hip).  Measured, 200k users x 25 paths, ms for vector | matrix cores (tools/ab_bench.py):
//   16 pairs: K=32 4.3 | 2.7, K=128 6.3 | 3.5, K=512 13.3 | 7.6        12 pairs: K=512 10.9 | 7.4     9 pairs: K=512 8.8 | 7.3
//    8 pairs: K=64 1.3 | 2.8, K=256 3.7 | 5.2, K=512 6.9 | 7.3, K=1024 13.5 | 11.6           4 pairs: K=1024 8.4 | 11.5
// The matrix-core time hardly depends on the pair count there (it is the B' generation), the vector kernel's is
// proportional to pairs x subcarriers.
bool fd_mfma_preferred(const dmx_params& prm, const WsView& ws) {
    const int M = prm.ue_shape[0] * prm.ue_shape[1] * prm.bs_shape[0] * prm.bs_shape[1];
    const int K = prm.n_selected;
    if (!fd_mfma_supported(prm, ws)) return false;
    if (M >= 9) return true;                  // 9 pairs x 64 subcarriers: 3.6 | 2.8
    return M == 8 && K >= 1024;
}

// Persistent launches: workgroups stride over the work items.  Exactly as many workgroups as are resident at once
// (registers and LDS decide: ask the runtime) is NOT the best grid: every CU then gets the same number of items and
// the kernel ends with the slowest CU.  A few items per workgroup keep the amortised start-up and let the dispatcher
// balance: at the headline shape 1 item per workgroup 18.6 ms, 390 (= resident grid) 18.3, 1.5 17.3, 3 16.9, 6 17.0,
// 12 17.1, 24 17.4 (tools/ab_bench.py, one process, interleaved).
static constexpr int ITEMS_PER_WG = 4;
static constexpr int ITEMS_PER_WG8 = 8;     // 8-wave workgroups: 0 (resident grid) 17.3 ms, 2 17.2, 4 16.4, 8 16.4, 16 16.5; config 5: 32.8, 32.8, 32.3, 31.2, 31.7
static int64_t resident_grid(const void* kfn, int threads, size_t smem, int64_t blocks, int items_per_wg) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, threads, smem) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    int64_t grid = (int64_t)device_cu_count() * per_cu;
    if (items_per_wg > 0) {
        // a few items per workgroup, but never fewer than four grids' worth of workgroups when the work allows it
        // (config 2 = 10k items: 0.43 ms with 2500 workgroups, 0.48 with 1250)
        int64_t g = blocks / items_per_wg;
        if (g < 4 * grid) g = blocks < 4 * grid ? blocks : 4 * grid;
        if (g > grid) grid = g;
    }
    return blocks < grid ? blocks : grid;
}

template <bool NT, int NW, int MODE = 0, int GSRC = 0>
static int launch_mfma_t(const WsView& ws, const MfmaArgs& a, int64_t blocks, size_t smem, float2* out, hipStream_t stream,
                         bool persistent = true, int items_per_wg = ITEMS_PER_WG) {
    const void* kfn = reinterpret_cast<const void*>(k2_fd_mfma<NT, NW, MODE, GSRC>);
    if (smem > 64 * 1024) {     // per device and cheap: no cached flag, so every GPU of a process gets it
        hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, MFMA_LDS_MAX);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    }
    int64_t grid = blocks;
    if (persistent) grid = resident_grid(kfn, NW * 64, smem, blocks, items_per_wg);
    hipLaunchKernelGGL((k2_fd_mfma<NT, NW, MODE, GSRC>), dim3((unsigned)grid), dim3(NW * 64), smem, stream, ws, a, reinterpret_cast<float*>(out), blocks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k2_fd_mfma launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

static int launch_mfma_any(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                           float2* out, int config, int n_beams, const float2* ftab, const int32_t* fexp,
                           const float2* gtab, hipStream_t stream, const uint2* gpack = nullptr);

int launch_channels_fd_mfma_gload(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                                  const float2* gtab, float2* out, hipStream_t stream, bool packed) {
    if (packed)
        return launch_mfma_any(prm, ws, user_begin, user_count, out, 0, 0, nullptr, nullptr, nullptr, stream,
                               reinterpret_cast<const uint2*>(gtab));
    return launch_mfma_any(prm, ws, user_begin, user_count, out, 0, 0, nullptr, nullptr, gtab, stream);
}

int launch_channels_fd_mfma(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                            float2* out, int config, hipStream_t stream) {
    return launch_mfma_any(prm, ws, user_begin, user_count, out, config, 0, nullptr, nullptr, nullptr, stream);
}

size_t beam_workspace_bytes(int64_t user_count, int n_beams, int P) {
    return align_up((size_t)user_count * (size_t)n_beams * (size_t)P * 8, 256) + align_up((size_t)user_count * 4, 256);
}

// projection f[b,l] = sum_tx F[b,tx] a_tx[tx,l] of every user into `beam_ws` (beam_workspace_bytes); `tabs` receives the
// two arrays inside it that the contraction kernels read
int launch_beam_project(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                        const float2* codebook, int n_beams, void* beam_ws, hipStream_t stream, BeamTabs* tabs) {
    BeamArgs b;
    b.user_begin = user_begin;
    b.m_tx = prm.bs_shape[0] * prm.bs_shape[1];
    b.bs_mh = prm.bs_shape[0];
    b.n_beams = n_beams;
    b.F = codebook;
    b.ftab = reinterpret_cast<float2*>(beam_ws);
    b.fexp = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(beam_ws) +
                                        align_up((size_t)user_count * (size_t)n_beams * (size_t)ws.P * 8, 256));
    // matrix-core projection when the codebook's f16 hi/lo image fits 64 KiB of LDS, scalar kernel otherwise
    const int kkpad = (2 * b.m_tx + 15) / 16 * 16;
    const int fstride = kkpad * 2 + 16;
    const int nbt = (n_beams + 31) / 32;
    const size_t smem_m = (size_t)2 * (nbt <= 1 ? 1 : (nbt <= 2 ? 2 : 4)) * 32 * fstride;
    if (nbt <= 4 && smem_m <= 64 * 1024) {
        int64_t grid = (user_count + 3) / 4;
        if (grid > 2048) grid = 2048;                      // persistent: the codebook is staged once per workgroup
        if (nbt <= 1) hipLaunchKernelGGL(k2b_beam_project_mfma<1>, dim3((unsigned)grid), dim3(256), smem_m, stream, ws, b, user_count, kkpad, fstride);
        else if (nbt <= 2) hipLaunchKernelGGL(k2b_beam_project_mfma<2>, dim3((unsigned)grid), dim3(256), smem_m, stream, ws, b, user_count, kkpad, fstride);
        else hipLaunchKernelGGL(k2b_beam_project_mfma<4>, dim3((unsigned)grid), dim3(256), smem_m, stream, ws, b, user_count, kkpad, fstride);
    } else {
        const size_t smem = (size_t)b.m_tx * (ws.P > 0 ? ws.P : 1) * 8;
        if (smem > 64 * 1024) { set_error("BS panel of %d elements x %d paths does not fit the beam-projection table", b.m_tx, ws.P); return DMX_ERR_SHAPE; }
        hipLaunchKernelGGL(k2b_beam_project, dim3((unsigned)user_count), dim3(256), smem, stream, ws, b);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k2b_beam_project launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    tabs->ftab = b.ftab;
    tabs->fexp = b.fexp;
    return DMX_OK;
}

int launch_channels_fd_beams(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                             const float2* codebook, int n_beams, void* beam_ws, float2* out, hipStream_t stream) {
    if (user_count == 0 || prm.n_selected == 0 || n_beams == 0) return DMX_OK;
    BeamTabs t;
    int rc = launch_beam_project(prm, ws, user_begin, user_count, codebook, n_beams, beam_ws, stream, &t);
    if (rc) return rc;
    return launch_mfma_any(prm, ws, user_begin, user_count, out, 0, n_beams, t.ftab, t.fexp, nullptr, stream);
}

static int launch_mfma_any(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                           float2* out, int config, int n_beams, const float2* ftab, const int32_t* fexp,
                           const float2* gtab, hipStream_t stream, const uint2* gpack) {
    MfmaArgs a;
    a.n_beams = n_beams; a.ftab = ftab; a.fexp = fexp; a.gtab = gtab; a.gpack = gpack;
    a.adaptive = !n_beams && !gtab && !gpack && (prm.flags & DMX_FLAG_ADAPTIVE_TERMS);   // default: three product terms everywhere
    a.alias_table = tuning_int("DMX_LPF_ALIAS_TABLE", 0);
    a.user_begin = user_begin;
    a.m_rx = prm.ue_shape[0] * prm.ue_shape[1];
    a.m_tx = prm.bs_shape[0] * prm.bs_shape[1];
    a.ue_mh = prm.ue_shape[0];
    a.bs_mh = prm.bs_shape[0];
    a.M = a.m_rx * (n_beams ? n_beams : a.m_tx);
    a.K = prm.n_selected;
    a.sc = prm.selected_subcarriers;
    a.inv_n = 1.0 / (double)prm.n_subcarriers;
    const int nstrips = (2 * a.K + 31) / 32;
    // rx_filter with the packed table, every strip full and whole 256-row blocks: gains through LDS-DMA (GSRC = 3) in ONE
    // 16-wave workgroup per CU (A' tiles 73.7 KB + 16 slots 64 KB).  Same box, headline shape x 100k users, FFT +
    // contraction: register loads (GSRC = 2, two 8-wave workgroups) 21.2 ms, this 20.7, with nt DMA and nt table stores
    // 20.2-20.3; 8 waves x 128-row blocks (two workgroups per CU, the fragments read twice per user) 21.5.
    // tuning build only: DMX_LPF_DMA=0 takes the register loads
    const int dma = (gpack && a.K % 16 == 0 && a.M >= MAX_ROWS && nstrips >= 16) ? tuning_int("DMX_LPF_DMA", 1) : 0;
    a.nblk = (a.M + MAX_ROWS - 1) / MAX_ROWS;
    const int mrows = a.M < MAX_ROWS ? a.M : MAX_ROWS;
    a.rows = (mrows + 31) / 32 * 32;
    const size_t smem = item_lds_bytes(a.rows);
    const int64_t blocks = user_count * a.nblk;
    if (blocks > 0x7fffffffLL) { set_error("too many workgroups for one call"); return DMX_ERR_SHAPE; }
    if ((size_t)a.rows * (size_t)a.K * 8 >= (size_t)1 << 31) {      // buffer descriptor / 32-bit offsets per row block
        set_error("%d selected subcarriers are too many for one 256-row block", a.K);
        return DMX_ERR_SHAPE;
    }
    // 16-wave form, three tile-loop bodies (separate kernels: together in one they spill inside the loops):
    //   MODE 0  fewer than 128 rows per block (< 4 tiles per strip): run-time-guarded tile, no spills (mfma_tile_rt)
    //   MODE 1  >= 128 rows, at most 16 path slots: grouped reads, plain loop
    //   MODE 2  >= 128 rows, more than 16 path slots (long MFMA chains): software-pipelined strip - 16.7 vs 17.5 ms at
    //           the headline shape, 15.1 vs 16.5 with random path counts; with 10 path slots (config 2 x 200k users)
    //           the short chains make it 9.5 vs 8.6 ms, hence MODE 1 there
    auto go16 = [&](bool persistent, int items_per_wg) {
        if (a.rows < 128) return launch_mfma_t<true, 16, 0>(ws, a, blocks, smem, out, stream, persistent, items_per_wg);
        if (ws.P <= 16) return launch_mfma_t<true, 16, 1>(ws, a, blocks, smem, out, stream, persistent, items_per_wg);
        return launch_mfma_t<true, 16, 2>(ws, a, blocks, smem, out, stream, persistent, items_per_wg);
    };
    // The same three bodies in 8-wave workgroups.  Every kernel here is compiled for 4 waves per SIMD (128 VGPRs); a
    // 16-wave workgroup then fills a CU alone, while two 8-wave workgroups share it and overlap each other's phases:
    // 16.0 vs 16.8 ms at the headline shape, 31.7 vs 32.5 at config 5, 8.3 vs 8.8 at config 2 x 200k users.
    auto go8 = [&](bool persistent, int items_per_wg) {
        if (a.rows < 128 || tuning_int("DMX_PLAIN_TILE_MODE", 2) == 0) return launch_mfma_t<true, 8, 0>(ws, a, blocks, smem, out, stream, persistent, items_per_wg);
        if (ws.P <= 16) return launch_mfma_t<true, 8, 1>(ws, a, blocks, smem, out, stream, persistent, items_per_wg);
        return launch_mfma_t<true, 8, 2>(ws, a, blocks, smem, out, stream, persistent, items_per_wg);
    };
    if (gtab || gpack) {                                   // rx_filter path: gains from the table (float or packed f16)
        const bool small = nstrips <= 8 && a.rows < 128;
        if (gpack) {
            // with the fragments coming out of loads the register-lean tile loop wins at every row count: headline shape
            // x 20k users, stage 2 in all: MODE 0 5.43 ms, MODE 1 5.85, MODE 2 (pipelined, 116 B/lane of scratch) 5.61
            const int m1 = tuning_int("DMX_LPF_TILE_MODE", 0);        // tuning build only: 1 / 2 = the grouped / pipelined bodies
            if (dma == 1) return launch_mfma_t<true, 16, 0, 3>(ws, a, blocks, smem, out, stream, true, ITEMS_PER_WG);
            if (small) return launch_mfma_t<true, 4, 0, 2>(ws, a, blocks, smem, out, stream, true, 0);
            if (a.rows >= 128 && m1 == 1) return launch_mfma_t<true, 8, 1, 2>(ws, a, blocks, smem, out, stream, true, ITEMS_PER_WG8);
            if (a.rows >= 128 && m1 == 2) return launch_mfma_t<true, 8, 2, 2>(ws, a, blocks, smem, out, stream, true, ITEMS_PER_WG8);
            return launch_mfma_t<true, 8, 0, 2>(ws, a, blocks, smem, out, stream, true, ITEMS_PER_WG8);
        }
        if (small) return launch_mfma_t<true, 4, 0, 1>(ws, a, blocks, smem, out, stream, true, 0);
        if (a.rows < 128) return launch_mfma_t<true, 8, 0, 1>(ws, a, blocks, smem, out, stream, true, ITEMS_PER_WG8);
        if (ws.P <= 16) return launch_mfma_t<true, 8, 1, 1>(ws, a, blocks, smem, out, stream, true, ITEMS_PER_WG8);
        return launch_mfma_t<true, 8, 2, 1>(ws, a, blocks, smem, out, stream, true, ITEMS_PER_WG8);
    }
    switch (config) {
        case 1: return launch_mfma_t<false, 16>(ws, a, blocks, smem, out, stream);   // plain stores
        case 2: return launch_mfma_t<true, 4>(ws, a, blocks, smem, out, stream);
        case 3: return go8(true, ITEMS_PER_WG8);                                     // 8 waves whatever the strip count
        case 6: return go16(false, 0);                                               // one workgroup per work item
        case 8: return go16(true, ITEMS_PER_WG);                                     // 16 waves whatever the strip count
        case 9: return go16(true, 0);                                                // exactly the resident workgroups
        case 0:
            // 4-wave workgroups where little work shares a set of A' tiles - at most 8 strips (K <= 128) on fewer than
            // 128 rows: 64 pairs x 32 subcarriers 1.12 vs 1.68 ms with 8 waves, 16 x 64: 1.08 vs 1.61.  From 128 rows
            // on the 8-wave form wins at every strip count (256 pairs: K=16 2.10 vs 2.40, K=64 3.00 vs 3.39, K=128
            // 4.68 vs 5.09, K=256 8.38 vs 9.07; a single 16-wave workgroup per CU is behind both everywhere).
            if (nstrips <= 8 && a.rows < 128) return launch_mfma_t<true, 4>(ws, a, blocks, smem, out, stream, true, 0);
            return go8(true, ITEMS_PER_WG8);
        default: set_error("unknown matrix-core kernel configuration %d", config); return DMX_ERR_ARG;
    }
}

}  // namespace dmx
