// Stage 2, frequency domain - "folded" matrix-core kernel for FEW antenna pairs (variant 12; the automatic choice for
// DeepMIMO's default-sized arrays - channel.py:36-46: BS 8x1, UE 1x1 - up to 32 pairs when the selected subcarriers
// are uniformly spaced).
//
// The plain matrix-core kernel (k2_channel_fd_mfma.hip) generates the subcarrier phasors G[l,k] = c_l e^{-j 2pi q_l sc_k}
// (q_l = dn_l / N) once per 16-subcarrier strip and reuses them over the M/32 row tiles of a user: with M <= 32 antenna
// pairs there is ONE half-empty tile per strip, the L x K sin/cos + f16 splits of G are the whole kernel (9-25 % of HBM).
// For a uniformly spaced selection sc_k = sc_0 + d*k the phasor factorises over k = 16a + b:
//     G[l, 16a+b] = c_l e^{-j 2pi q_l (sc_0 + 16 d a)} * e^{-j 2pi q_l d b} = c_l E1[l,a] * E2[l,b]
// so  H[p, 16a+b] = sum_l (A[p,l] c_l E1[l,a]) * E2[l,b]:
// the 16-subcarrier blocks `a` FOLD INTO THE ROW dimension.  Per user the product is a real GEMM
//     C[(a,p)][2b+c] = sum_kk At'[(a,p)][kk] * E2'[kk][2b+c]        kk = 2l + {re, im}, 64 deep (32 path slots)
// with ONE 32-column B operand E2' (16 sin/cos pairs per lane per user, kept in registers) and M*K/16 rows that are
// packed densely into 32-row tiles whatever M is (8 pairs: four subcarrier blocks per tile).  The A operand
// At[(a,p)][l] = Ac[p][l] * E1[a][l] is one complex multiply of two small per-user tables (Ac = c_l a_rx a_tx: M x 32,
// E1: K/16 x 32 entries) - the sin/cos count per user drops from L*K to L*(16 + K/16 + M).
// Precision: as in k2_channel_fd_mfma.hip - operands split x = hi + lo in f16, three MFMAs (hi*hi + hi*lo + lo*hi)
// into one fp32 accumulator, power-of-two operand scales undone exactly in the epilogue; the two extra fp32 complex
// products per operand add ~2e-7 relative.
//
// Mapping: ONE WAVE owns one (user, chunk of <= CH subcarrier blocks) work item from start to end - no workgroup
// barrier after the start-up tables, nothing shared between the four waves of a workgroup but the user-independent
// row tables.  Per item: path records -> per-wave LDS tables (Ac, E1, q) -> E2' fragments in registers -> per 32-row
// tile: At' fragments built in registers straight into the MFMA operand layout (4 complex products + 4 packed f16
// splits per K-step), 3 MFMAs per K-step, 16 non-temporal buffer_store_dword (two 128-B row segments each; the row
// offsets come from a table because row (a,p) lives at (p*K + 16a)*8 bytes).  Persistent grid.
#include "dmx_common.h"
#include <stdlib.h>
// see DMX_MFMA_RESULT_GUARD in k2_mfma_frag.h (this file does not include it)
#define FOLD_TILE_GUARD asm volatile("s_nop 3")

namespace dmx {

typedef _Float16 fh8 __attribute__((ext_vector_type(8)));
typedef _Float16 fh2 __attribute__((ext_vector_type(2)));
typedef __fp16 fhp2 __attribute__((ext_vector_type(2)));
typedef float ff16 __attribute__((ext_vector_type(16)));

static constexpr int FOLD_TROW = 272;            // bytes per table row: 32 complex64 + 16 B pad (conflict-free ds_read_b128)
static constexpr float FOLD_B_SCALE = 64.0f;     // 2^6 on the unit-modulus E2' operand
static constexpr int FOLD_MAX_M = 128;          // u8 element indices, rowsrc packing (128 x 272 < 65536)
static constexpr int FOLD_SHARED_FROM = 33;     // antenna pairs from which the four waves of a workgroup share one set of tables

struct FoldArgs {
    int64_t user_begin;
    int m_tx, ue_mh, bs_mh;
    int M, K;
    int sc_first, sc_stride;
    double inv_n;
    int ch;            // subcarrier blocks per E1 table (inner chunk)
    int sch;           // subcarrier blocks per work item (multiple of ch): E2' is built once per item
    int nblk;          // ceil(K / 16)
    int nchunk;        // ceil(nblk / ch)
    int nsuper;        // ceil(nblk / sch): work items per user
    int nb_last;       // blocks of the last inner chunk
    int tab_rows;      // rows of the row tables (multiple of 32)
    int adaptive;      // 1 = a weak last K-step may take one product term
    int k_tail;        // K - 16*(nblk-1): valid subcarriers of the last block (1..16)
};

typedef float fv2 __attribute__((ext_vector_type(2)));

// x = hi + lo in f16.  The residual x - hi is ONE mixed-precision fma per value (v_fma_mix_f32 reads the f16 half of the
// packed register in place) - selected by the compiler from fma(f16 -> f32, m1, x) with m1 = WsView::neg_one; see k2_mfma_frag.h for why
// this is no longer inline asm.
__device__ __forceinline__ void fold_split2(float x0, float x1, fh2& hi, fh2& lo, float m1) {
    const fhp2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const fh2 hh = __builtin_bit_cast(fh2, h);
    const float r0 = __builtin_fmaf((float)hh[0], m1, x0);
    const float r1 = __builtin_fmaf((float)hh[1], m1, x1);
    const fhp2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi = hh;
    lo = __builtin_bit_cast(fh2, l);
}

// float -> its bits, by VALUE: __builtin_bit_cast applied directly to a vector-element lvalue (`bit_cast(unsigned, v[i])`)
// reads element 0 whatever i is with this compiler; through a by-value parameter it is the element asked for
__device__ __forceinline__ unsigned fold_bits(float x) { return __builtin_bit_cast(unsigned, x); }

// (a + jb)(c + jd) as two packed instructions: [ac, ad] then fma([-b, b], [d, c], .)
__device__ __forceinline__ fv2 fold_cmul(float a, float b, float c, float d) {
    const fv2 t = fv2{a, a} * fv2{c, d};
    return __builtin_elementwise_fma(fv2{-b, b}, fv2{d, c}, t);
}

// a wave's own LDS writes are visible to its later reads (DS operations of one wave execute in order); the compiler
// only has to be kept from moving accesses across this point
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__host__ __device__ inline size_t fold_static_bytes(int tab_rows, int M) { return (size_t)tab_rows * 12 + align_up((size_t)M * 4, 16); }
__host__ __device__ inline size_t fold_wave_bytes(int M, int ch) { return 256 + (size_t)(M + ch) * FOLD_TROW; }

// synchronisation between writers and readers of the per-item tables: the owning wave alone (WS = 1: its DS operations
// execute in order) or the four waves of the workgroup that share one set of tables (WS = 4)
template <int WS>
__device__ __forceinline__ void fold_sync() {
    if constexpr (WS == 4) __syncthreads();
    else wave_lds_sync();
}

// One 32-row tile (a,p) x 32 columns: NS K-steps of A' = Ac[p][l] E1[a][l] (complex products, then the f16 hi / lo split)
// against the item's E2' fragments, and the un-scaling of the accumulator.  NS and LW (weak last K-step: A'hi B'hi only,
// no A'lo built) are template parameters, each K-step's MFMAs are fenced with sched_barrier, the choice of (NS, LW) is made
// OUTSIDE the tile loop (fold_tiles below), and even / odd K-steps accumulate into two different accumulators - all on
// purpose.  Round 2 went through four builds of this loop that were not bit-reproducible: two identical launches differed
// in accumulator registers 8..15 of one tile (the ones an MFMA writes in its last passes; 16 subcarriers x the tile's
// second 16 rows), by 5e-4 ... 1e-1 of the user's peak - for 1 user in 200 (K-steps guarded at run time with the adaptive
// branch inside; or the scheduler threading the next A' split between a K-step's MFMAs), for 1 user-launch in 10^7 on
// every box (a per-tile switch that the compiler merged into shared blocks), and for 1 in 10^7...10^9 on SOME boxes of the
// pool only (one accumulator, everything else as now).  Every parity test on a few hundred users stayed green throughout.
// What the failing builds share: an MFMA that accumulates onto the result of an MFMA issued roughly one MFMA-group
// duration earlier (32 ... 120 cycles), i.e. whose SrcC is being written back right when it is picked up; builds that moved
// dependent MFMAs towards that distance failed more often (four hidden wait states behind every K-step's group: 1,000 x
// more), builds that moved them away less.  Direct probes on boxes of unknown susceptibility show nothing of it
// (tools/mfma_*_probe.hip: sources latched at issue, dependent MFMAs correct at every tested distance, 12 wait states for
// the last accumulator register).  With two accumulators an MFMA only ever accumulates back to back onto its own K-step or
// onto a K-step two steps back: on a box where the one-accumulator build differed in 14 of 600 million user-launches this
// one differed in 0 of 1.4 billion (tools/hot_box_hunt.sh).  tests/test_gpu_parity.py::test_launches_are_bit_reproducible
// and the stress tests in tests/test_gpu_fullsize.py guard it; tools/repro_stress.py is the long form.
template <int NS, bool LW>
__device__ __forceinline__ ff16 fold_tile(const unsigned char* arow, const unsigned char* erow, const fh8 (&Bhi)[4],
                                          const fh8 (&Blo)[4], float m1, float oscale) {
    // TWO accumulators, even and odd K-steps: an MFMA then never accumulates onto the result of the K-step just before it,
    // only onto the one two steps back, long finished.  With one accumulator the first MFMA of K-step s + 1 was issued about
    // one K-step's A' construction (~120 cycles) behind the three MFMAs of K-step s (96 cycles of matrix pipe) - right around
    // the moment their last pass wrote accumulator registers 8..15 - and on some boxes of the pool, once in 10^7...10^9
    // users, those registers went into the next MFMA without the last contribution (every build that
    // moved dependent MFMAs towards that distance failed more often, the ones that moved them away less).
    ff16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float4 x0 = *reinterpret_cast<const float4*>(arow + s * 64);
        const float4 x1 = *reinterpret_cast<const float4*>(arow + s * 64 + 16);
        const float4 y0 = *reinterpret_cast<const float4*>(erow + s * 64);
        const float4 y1 = *reinterpret_cast<const float4*>(erow + s * 64 + 16);
        const fv2 z0 = fold_cmul(x0.x, x0.y, y0.x, y0.y), z1 = fold_cmul(x0.z, x0.w, y0.z, y0.w);
        const fv2 z2 = fold_cmul(x1.x, x1.y, y1.x, y1.y), z3 = fold_cmul(x1.z, x1.w, y1.z, y1.w);
        fh8 Ah;
        if (LW && s == NS - 1) {
            const fh2 p0 = __builtin_bit_cast(fh2, __builtin_amdgcn_cvt_pkrtz(z0[0], z0[1]));
            const fh2 p1 = __builtin_bit_cast(fh2, __builtin_amdgcn_cvt_pkrtz(z1[0], z1[1]));
            const fh2 p2 = __builtin_bit_cast(fh2, __builtin_amdgcn_cvt_pkrtz(z2[0], z2[1]));
            const fh2 p3 = __builtin_bit_cast(fh2, __builtin_amdgcn_cvt_pkrtz(z3[0], z3[1]));
            Ah[0] = p0[0]; Ah[1] = p0[1]; Ah[2] = p1[0]; Ah[3] = p1[1]; Ah[4] = p2[0]; Ah[5] = p2[1]; Ah[6] = p3[0]; Ah[7] = p3[1];
            __builtin_amdgcn_sched_barrier(0);
            acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bhi[s], acc[s & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            fh8 Al;
            fh2 ph, pl;
            fold_split2(z0[0], z0[1], ph, pl, m1);
            Ah[0] = ph[0]; Ah[1] = ph[1]; Al[0] = pl[0]; Al[1] = pl[1];
            fold_split2(z1[0], z1[1], ph, pl, m1);
            Ah[2] = ph[0]; Ah[3] = ph[1]; Al[2] = pl[0]; Al[3] = pl[1];
            fold_split2(z2[0], z2[1], ph, pl, m1);
            Ah[4] = ph[0]; Ah[5] = ph[1]; Al[4] = pl[0]; Al[5] = pl[1];
            fold_split2(z3[0], z3[1], ph, pl, m1);
            Ah[6] = ph[0]; Ah[7] = ph[1]; Al[6] = pl[0]; Al[7] = pl[1];
            // the three MFMAs of a K-step stay together, back to back (dependent MFMAs issued back to back are the path
            // every GEMM kernel exercises), nothing scheduled between or right behind them
            __builtin_amdgcn_sched_barrier(0);
            acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bhi[s], acc[s & 1], 0, 0, 0);
            acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Blo[s], acc[s & 1], 0, 0, 0);
            acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bhi[s], acc[s & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    FOLD_TILE_GUARD;
    ff16 out;
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = (NS > 1 ? acc[0][i] + acc[1][i] : acc[0][i]) * oscale;
    return out;
}

// The row tiles of one inner chunk for a fixed (NS, LW): tile body (fold_tile) and its 16 stores.
template <bool NT, int WS, int NS, bool LW>
__device__ __forceinline__ void fold_tiles(int sub, int ntiles, const uint32_t* rowsrc, const uint32_t* rowoff, int lp, int hh,
                                           const unsigned char* Ac, const unsigned char* E1, const fh8 (&Bhi)[4], const fh8 (&Blo)[4],
                                           float m1, float oscale, bool masked, __amdgpu_buffer_rsrc_t orsrc, uint32_t lane_col,
                                           uint32_t lmask_last) {
    for (int rt = sub; rt < ntiles; rt += WS) {
        const uint32_t src = rowsrc[(rt << 5) + lp];
        const unsigned char* arow = Ac + (src >> 16) + hh * 32;
        const unsigned char* erow = E1 + (src & 0xFFFFu) + hh * 32;
        // the tile, un-scaled: register i is row (i&3) + 8*(i>>2) + 4*(lane>>5), column lane&31
        const ff16 sv = fold_tile<NS, LW>(arow, erow, Bhi, Blo, m1, oscale);
        const uint4* ro4 = reinterpret_cast<const uint4*>(rowoff + (rt << 5) + 4 * hh);
        if (masked) {                                                  // wave-uniform: partial last block in this item
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint4 ro = ro4[2 * g];
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 0]), orsrc, lane_col + (ro.x & lmask_last), 0, NT ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 1]), orsrc, lane_col + (ro.y & lmask_last), 0, NT ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 2]), orsrc, lane_col + (ro.z & lmask_last), 0, NT ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 3]), orsrc, lane_col + (ro.w & lmask_last), 0, NT ? 2 : 0);
            }
        } else {                                                       // rows past the item carry 0xC0000000: out of range as they are
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint4 ro = ro4[2 * g];
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 0]), orsrc, lane_col + ro.x, 0, NT ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 1]), orsrc, lane_col + ro.y, 0, NT ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 2]), orsrc, lane_col + ro.z, 0, NT ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(fold_bits(sv[4 * g + 3]), orsrc, lane_col + ro.w, 0, NT ? 2 : 0);
            }
        }
    }
}

// WS = waves sharing a work item: 1 - every wave owns its items and tables (few antenna pairs: the tables are small, no
// workgroup barrier anywhere); 4 - the workgroup works on one item with ONE set of tables and splits its row tiles over
// the waves (33 ... 128 pairs: per-wave tables would leave one workgroup per CU).
template <bool NT, int WS>
__global__ __launch_bounds__(256, 4) void k2_fd_fold(WsView ws, FoldArgs a, float* __restrict__ out, int64_t items) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* rowoff0 = reinterpret_cast<uint32_t*>(smem);                 // [tab_rows] byte offset of row (a,p), full chunk
    uint32_t* rowoff1 = rowoff0 + a.tab_rows;                              // [tab_rows] same for the last chunk
    uint32_t* rowsrc = rowoff1 + a.tab_rows;                               // [tab_rows] (E1 row | Ac row << 16) byte offsets
    uint32_t* pidx = rowsrc + a.tab_rows;                                  // [M] element indices of pair p, 4 x u8
    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index is uniform by construction; saying so keeps everything derived from it - the work item, its
    // pointers, the output buffer descriptor, the path count - in scalar registers (a descriptor in vector registers
    // makes every buffer_store a readfirstlane waterfall loop)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* wbase = smem + fold_static_bytes(a.tab_rows, a.M) + (WS == 1 ? (size_t)wave * fold_wave_bytes(a.M, a.ch) : (size_t)0);
    float2* qtab = reinterpret_cast<float2*>(wbase);                       // [32] q_l as (multiple of 2^-12, remainder)
    unsigned char* Ac = wbase + 256;                                       // [M][FOLD_TROW]  c_l a_rx a_tx (scaled)
    unsigned char* E1 = Ac + (size_t)a.M * FOLD_TROW;                      // [ch][FOLD_TROW] e^{-j 2pi q_l sc(16 a)}

    // ---- user-independent tables, once per workgroup
    const int M = a.M, K = a.K;
    for (int r = tid; r < a.tab_rows; r += 256) {
        const int ab = r / M, p = r - ab * M;
        const uint32_t off = (uint32_t)(p * K + 16 * ab) * 8u;
        const int nb0 = a.nchunk > 1 ? a.ch : a.nb_last;
        // rows past the item: 0xC0000000 stays beyond num_records (< 2^30) under either lane mask and cannot wrap
        // when the lane's column offset is added
        rowoff0[r] = ab < nb0 ? off : 0xC0000000u;
        uint32_t o1 = ab < a.nb_last ? off : 0xC0000000u;
        if (ab == a.nb_last - 1 && a.k_tail < 16) o1 |= 0x80000000u;      // partial last block: lanes past K keep bit 31
        rowoff1[r] = o1;
        const bool any = ab < (a.ch > a.nb_last ? a.ch : a.nb_last);
        rowsrc[r] = any ? ((uint32_t)(ab * FOLD_TROW) | ((uint32_t)(p * FOLD_TROW) << 16)) : 0u;
    }
    for (int p = tid; p < M; p += 256) {
        const int rx = p / a.m_tx, tx = p - rx * a.m_tx;
        pidx[p] = (uint32_t)(rx % a.ue_mh) | ((uint32_t)(rx / a.ue_mh) << 8) | ((uint32_t)(tx % a.bs_mh) << 16) |
                  ((uint32_t)(tx / a.bs_mh) << 24);
    }
    __syncthreads();

    const int lp = lane & 31, hh = lane >> 5;
    const int col = lane & 31, bsc = col >> 1, cpart = col & 1;
    const int kb = a.sc_stride * bsc;
    const float kbl = (float)(kb & 4095), kbf = (float)kb, cq = 0.25f * (float)cpart;
    const uint32_t lane_col = (uint32_t)col * 4u;
    const uint32_t lmask_last = bsc < a.k_tail ? 0x7FFFFFFFu : 0xFFFFFFFFu;
    const size_t user_floats = (size_t)M * K * 2;

    constexpr int IW = 4 / WS;                                             // work items a workgroup has in flight
    const int sub = WS == 4 ? wave : 0;                                    // this wave's share of table rows / row tiles
    for (int64_t item = (int64_t)blockIdx.x * IW + (WS == 1 ? wave : 0); item < items; item += (int64_t)gridDim.x * IW) {
        const int64_t ul = item / a.nsuper;
        const int sc = (int)(item - ul * a.nsuper);
        const int64_t u = a.user_begin + ul;
        const int b0 = sc * a.sch;                                         // first subcarrier block of this item
        const int bn = (a.nblk - b0) < a.sch ? (a.nblk - b0) : a.sch;      // its blocks
        int n_act = ws.n_keep[u];
        n_act = n_act < 32 ? n_act : 32;
        if (n_act == 0) {                                                  // channel.py:270-271
            float* __restrict__ oz = out + (size_t)ul * user_floats + (size_t)b0 * 32;
            const int k0 = b0 * 16, k1 = (k0 + bn * 16) < K ? (k0 + bn * 16) : K;
            const int nfl = (k1 - k0) * 2;
            for (int p = sub; p < M; p += WS)
                for (int i = lane; i < nfl; i += 64) oz[(size_t)p * K * 2 + i] = 0.f;
            continue;
        }
        // ---- path records of this user: lane (and lane + 32) = path slot
        const size_t rb = (size_t)u * ws.P + lp;
        const bool ok = lp < n_act;
        const float cr = ok ? ws.c_re[rb] : 0.f, ci = ok ? ws.c_im[rb] : 0.f;
        const double q = ok ? (double)ws.dn[rb] * a.inv_n : 0.0;
        const double rxy = ok ? ws.rx_y[rb] : 0.0, rxz = ok ? ws.rx_z[rb] : 0.0;
        const double txy = ok ? ws.tx_y[rb] : 0.0, txz = ok ? ws.tx_z[rb] : 0.0;
        float m = fmaxf(fabsf(cr), fabsf(ci));
        for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        // adaptive precision as in k2_channel_fd_mfma.hip (stage_item): a last K-step whose paths are all <= 2^-11 of the
        // strongest one in amplitude is multiplied out in one f16 term (here: A'hi only is built for it)
        const int l0w = ((n_act - 1) >> 3) << 3;
        const float a2 = fmaf(cr, cr, ci * ci);
        float m2 = a2, mw2 = lp >= l0w ? a2 : 0.f;
        for (int off = 16; off > 0; off >>= 1) { m2 = fmaxf(m2, __shfl_xor(m2, off)); mw2 = fmaxf(mw2, __shfl_xor(mw2, off)); }
        const bool last_weak = a.adaptive && l0w >= 8 && mw2 * 4194304.0f <= m2;
        int e;
        (void)frexpf(m, &e);                                               // m = f * 2^e, f in [0.5, 1)
        const float gs = ldexpf(1.0f, 10 - e);                             // max |c| component -> [512, 1024)
        const float oscale = ldexpf(1.0f, e - 10 - 6);                     // 1 / (gs * FOLD_B_SCALE)
        const double qh = rint(q * 4096.0) * (1.0 / 4096.0);
        const float qhf = (float)qh, qlf = (float)(q - qh);
        const float cgr = cr * gs, cgi = ci * gs;
        fold_sync<WS>();                                                   // the previous item's table reads are done
        if (lane < 32 && sub == 0) qtab[lane] = make_float2(qhf, qlf);
        // Ac[p][l] = c_l a_rx[rx,l] a_tx[tx,l]   (geometry.py:85-102; phases in float64 revolutions)
        for (int p = 2 * sub + hh; p < M; p += 2 * WS) {
            const uint32_t ix = pidx[p];
            const double ph = (double)(ix & 255u) * rxy + (double)((ix >> 8) & 255u) * rxz +
                              (double)((ix >> 16) & 255u) * txy + (double)(ix >> 24) * txz;
            float s, c;
            sincos_rev(frac_rev(ph), s, c);
            *reinterpret_cast<float2*>(Ac + (size_t)p * FOLD_TROW + lp * 8) = make_float2(cgr * c - cgi * s, cgr * s + cgi * c);
        }
        fold_sync<WS>();

        // ---- E2' fragments (B operand, 32 columns = 16 subcarrier offsets x {re, im}); element j of K-step s is row
        // kk = 16s + 8h + j, i.e. path 8s + 4h + (j>>1), component j&1.  Column c = 0 holds (Re G, -Im G) = (cos, sin)
        // of the phase, column c = 1 holds (Im G, Re G) = the same pair a quarter revolution later.
        const int nsteps = (n_act + 7) >> 3;
        // wave-uniform by construction, and made scalar for the compiler: a divergent switch around MFMAs (which ignore the
        // EXEC mask) is not something to leave to the structurizer
        const int tile_kind = __builtin_amdgcn_readfirstlane(2 * nsteps + (last_weak ? 1 : 0));
        fh8 Bhi[4], Blo[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            Bhi[s] = fh8{0, 0, 0, 0, 0, 0, 0, 0};
            Blo[s] = Bhi[s];
            if (s < nsteps) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float2 qq = qtab[8 * s + 4 * hh + jj];
                    const float p1 = fmaf(qq.x, kbl, cq);
                    float sn, cs;
                    sincos_rev(fmaf(qq.y, kbf, p1 - rintf(p1)), sn, cs);
                    fh2 ph, pl;
                    fold_split2(cs * FOLD_B_SCALE, sn * FOLD_B_SCALE, ph, pl, ws.neg_one);
                    Bhi[s][2 * jj] = ph[0]; Bhi[s][2 * jj + 1] = ph[1];
                    Blo[s][2 * jj] = pl[0]; Blo[s][2 * jj + 1] = pl[1];
                }
            }
        }

        // ---- inner chunks of <= ch subcarrier blocks: one E1 table each, the same E2' fragments for all of them
        for (int c0 = 0; c0 < bn; c0 += a.ch) {
        const int a0 = b0 + c0;
        const int nb = (bn - c0) < a.ch ? (bn - c0) : a.ch;
        const bool last = a0 + nb == a.nblk;
        float* __restrict__ o = out + (size_t)ul * user_floats + (size_t)a0 * 32;
        fold_sync<WS>();                                                   // the previous chunk's tiles have read E1
        // E1[a][l] = e^{-j 2pi q_l sc(16 (a0 + a))}; q = qh + ql with qh a multiple of 2^-12, so qh * (sc mod 4096) is
        // exact in float32 and qh * (sc - sc mod 4096) is an integer (k2_channel_fd_mfma.hip gen_b_step)
        for (int ab = 2 * sub + hh; ab < nb; ab += 2 * WS) {
            const int sca = a.sc_first + a.sc_stride * 16 * (a0 + ab);
            const float p1 = qhf * (float)(sca & 4095);
            float s, c;
            sincos_rev(fmaf(qlf, (float)sca, p1 - rintf(p1)), s, c);
            *reinterpret_cast<float2*>(E1 + (size_t)ab * FOLD_TROW + lp * 8) = make_float2(c, -s);
        }
        fold_sync<WS>();

        // ---- row tiles: 32 rows (a,p) each
        const uint32_t* rowoff = last ? rowoff1 : rowoff0;
        const bool masked = last && a.k_tail < 16;
        const int rows = nb * M;
        const int ntiles = (rows + 31) >> 5;
        const __amdgpu_buffer_rsrc_t orsrc =
            __builtin_amdgcn_make_buffer_rsrc(o, 0, (int)((unsigned)(user_floats - (size_t)a0 * 32) * 4u), 0x00020000);
        // one loop nest per (K-steps, weak last step): the choice is made out here, so that inside a loop every tile is
        // the same straight-line code from its first LDS read to its last store
        switch (tile_kind) {
            case 2: fold_tiles<NT, WS, 1, false>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, oscale, masked, orsrc, lane_col, lmask_last); break;
            case 4: fold_tiles<NT, WS, 2, false>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, oscale, masked, orsrc, lane_col, lmask_last); break;
            case 5: fold_tiles<NT, WS, 2, true>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, oscale, masked, orsrc, lane_col, lmask_last); break;
            case 6: fold_tiles<NT, WS, 3, false>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, oscale, masked, orsrc, lane_col, lmask_last); break;
            case 7: fold_tiles<NT, WS, 3, true>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, oscale, masked, orsrc, lane_col, lmask_last); break;
            case 8: fold_tiles<NT, WS, 4, false>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, oscale, masked, orsrc, lane_col, lmask_last); break;
            default: fold_tiles<NT, WS, 4, true>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, oscale, masked, orsrc, lane_col, lmask_last); break;
        }
        }                                                                  // inner chunks
    }
}

// ---------------------------------------------------------------------------------------------------------------------
bool fd_fold_supported(const dmx_params& prm, const WsView& ws) {
    (void)ws;
    const int64_t M = (int64_t)prm.ue_shape[0] * prm.ue_shape[1] * prm.bs_shape[0] * prm.bs_shape[1];
    return prm.sc_stride > 0 && prm.n_selected >= 1 && M <= FOLD_MAX_M;
}

// Subcarrier blocks per E1 table: the per-wave LDS tables ((M + ch) rows of 272 B) decide how many workgroups share a
// CU, and this kernel's waves wait on memory and LDS for a good part of their time - so the largest of 32 / 16 / 8
// that still lets FOUR workgroups (16 waves) share the 160 KiB, else three.  E2' is built once per work item whatever
// ch is, so a smaller ch costs only its loop overhead.
static constexpr int FOLD_SUPER = 64;            // blocks per work item (1024 subcarriers)
static int fold_chunk_blocks(int M, int nblk, int forced, int table_sets) {
    const int cand[3] = {32, 16, 8};
    if (forced > 0) return forced < nblk ? forced : nblk;
    for (int want = 4; want >= 2; --want) {
        for (int ch : cand) {
            const int c = ch < nblk ? ch : nblk;
            const int tab = (M * c + 31) / 32 * 32;
            const size_t smem = fold_static_bytes(tab, M) + table_sets * fold_wave_bytes(M, c);
            if (smem * want <= 160 * 1024) return c;
        }
    }
    return 8 < nblk ? 8 : nblk;
}

// Automatic choice (variant 0), from profiles/r2_fold_sweep.txt (25 paths; ms for this kernel | the best of the others:
// 9 small-output, 2 matrix cores, 1 fp32 vector).  200k users:
//    8 pairs: K=4 0.17 | 0.19 (9)   K=8 0.20 | 0.30 (9)   K=16 0.24 | 0.52 (9)   K=64 0.28 | 1.31 (1)   K=512 1.35 | 6.6 (2)
//   32 pairs: K=8 0.35 | 0.47 (9)   K=16 0.41 | 1.16 (9)   K=64 0.76 | 2.29 (2)   K=256 2.79 | 5.2 (2)   K=512 5.7 | 7.9 (2)
// 100k users, one table set per workgroup (33+ pairs):
//   48 pairs: K=64 0.74 | 1.11   K=256 1.98 | 3.20   K=512 4.15 | 4.79        64 pairs: K=64 0.86 | 1.23   K=128 1.38 | 1.88
//   64 pairs: K=256 2.83 | 3.47  K=512 5.34 | 5.07   K=1024 10.6 | 9.1        96 pairs: K=64 1.28 | 1.59   K=256 4.17 | 3.96
//  128 pairs: K=64 1.64 | 2.57   K=256 5.48 | 4.78   K=512 10.8 | 8.6
// This kernel levels off near 4.8-5.0 TB/s; the plain matrix-core kernel, whose B' generation is shared by M/32 full row
// tiles, passes it from the many-subcarrier end as the panel grows.
bool fd_fold_preferred(const dmx_params& prm, const WsView& ws) {
    const int64_t M = (int64_t)prm.ue_shape[0] * prm.ue_shape[1] * prm.bs_shape[0] * prm.bs_shape[1];
    const int K = prm.n_selected;
    // below one 16-subcarrier block this kernel's time is flat in K while the small-output kernel's grows with M*K:
    // 8 pairs K=1 0.15 | 0.12 (9), K=3 0.175 | 0.170, K=4 0.17 | 0.19; 64 pairs K=1 0.64 | 0.18, K=4 0.69 | 0.54, K=8 0.76 | 1.07
    if (!fd_fold_supported(prm, ws) || K < (M <= 16 ? 4 : 6)) return false;
    if (M <= 32) return true;
    if (M <= 48) return K <= 512;
    if (M <= 64) return K <= 256;
    if (M <= 96) return K <= 128;
    return K <= 64;
}

int launch_channels_fd_fold(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count, float2* out,
                            int chunk_blocks, hipStream_t stream) {
    if (user_count == 0 || prm.n_selected == 0) return DMX_OK;
    if (!fd_fold_supported(prm, ws)) {
        set_error("the folded kernel needs a uniformly spaced subcarrier selection (dmx_params.sc_stride > 0) and at most %d antenna pairs", FOLD_MAX_M);
        return DMX_ERR_SHAPE;
    }
    FoldArgs a;
    a.user_begin = user_begin;
    a.m_tx = prm.bs_shape[0] * prm.bs_shape[1];
    a.ue_mh = prm.ue_shape[0];
    a.bs_mh = prm.bs_shape[0];
    a.M = prm.ue_shape[0] * prm.ue_shape[1] * a.m_tx;
    a.K = prm.n_selected;
    a.sc_first = prm.sc_first;
    a.sc_stride = prm.sc_stride;
    a.inv_n = 1.0 / (double)prm.n_subcarriers;
    a.nblk = (a.K + 15) / 16;
    if (chunk_blocks <= 0) {                       // measurement hook: DMX_FOLD_CHUNK=8|16|32 overrides the LDS-occupancy rule
        const char* env = getenv("DMX_FOLD_CHUNK");
        if (env) chunk_blocks = atoi(env);
        if (chunk_blocks != 8 && chunk_blocks != 16 && chunk_blocks != 32) chunk_blocks = 0;
    }
    // table sets per workgroup: one per wave up to 32 pairs, one for the whole workgroup above (measurement hook
    // DMX_FOLD_SHARED=0|1 forces either)
    bool shared = a.M >= FOLD_SHARED_FROM;
    if (const char* env = getenv("DMX_FOLD_SHARED")) shared = env[0] == '1';
    const int sets = shared ? 1 : 4;
    a.ch = fold_chunk_blocks(a.M, a.nblk, chunk_blocks, sets);
    a.sch = (FOLD_SUPER / a.ch) * a.ch;
    if (a.sch > a.nblk) a.sch = (a.nblk + a.ch - 1) / a.ch * a.ch;
    a.nsuper = (a.nblk + a.sch - 1) / a.sch;
    a.nchunk = (a.nblk + a.ch - 1) / a.ch;
    a.nb_last = a.nblk - a.ch * (a.nchunk - 1);
    a.tab_rows = (a.M * a.ch + 31) / 32 * 32;
    a.k_tail = a.K - 16 * (a.nblk - 1);
    a.adaptive = getenv("DMX_NO_ADAPTIVE") == nullptr;              // env = measurement hook: always three terms
    if ((size_t)a.M * (size_t)a.K * 8 >= (size_t)1 << 30) { set_error("%d x %d outputs per user are too many for the folded kernel", a.M, a.K); return DMX_ERR_SHAPE; }
    const size_t smem = fold_static_bytes(a.tab_rows, a.M) + sets * fold_wave_bytes(a.M, a.ch);
    if (smem > 160 * 1024) { set_error("folded kernel tables of %zu bytes exceed LDS", smem); return DMX_ERR_SHAPE; }
    const int64_t items = user_count * a.nsuper;
    const void* kfn = shared ? reinterpret_cast<const void*>(k2_fd_fold<true, 4>) : reinterpret_cast<const void*>(k2_fd_fold<true, 1>);
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
    const int64_t need = shared ? items : (items + 3) / 4;
    if (grid > need) grid = need;
    if (shared) hipLaunchKernelGGL((k2_fd_fold<true, 4>), dim3((unsigned)grid), dim3(256), smem, stream, ws, a, reinterpret_cast<float*>(out), items);
    else hipLaunchKernelGGL((k2_fd_fold<true, 1>), dim3((unsigned)grid), dim3(256), smem, stream, ws, a, reinterpret_cast<float*>(out), items);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k2_fd_fold launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

}  // namespace dmx
