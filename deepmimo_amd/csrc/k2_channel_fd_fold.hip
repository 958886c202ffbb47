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
// row tables (33+ pairs: the four waves share one set of tables and split the row tiles).  Per item: path records ->
// per-wave LDS tables (Ac, E1, q) -> E2' fragments in registers -> per 32-row tile and 8-path K-step ONE asm statement
// that multiplies and splits the lane's table entries in place and loads the next K-step's (24 vector instructions, the
// entries in v[112:127] by name), 3 MFMAs per K-step into two accumulators (even / odd K-steps), the result guard, a
// packed add + scale, 16 non-temporal buffer_store_dword (two 128-B row segments each).  Row (a,p) lives at
// (p*K + 16a)*8 bytes: scalar arithmetic in the store's soffset for a power-of-two pair count, a table otherwise.
// Persistent grid, at most four workgroups per CU.  The shape of the tile loop is dictated by the round-2 incident:
// see the comment in front of it and DESIGN.md section 4.
#include "dmx_common.h"
#include "dmx_tuning.h"

namespace dmx {

typedef _Float16 fh8 __attribute__((ext_vector_type(8)));
typedef _Float16 fh2 __attribute__((ext_vector_type(2)));
typedef __fp16 fhp2 __attribute__((ext_vector_type(2)));
typedef float ff16 __attribute__((ext_vector_type(16)));
typedef unsigned fu4 __attribute__((ext_vector_type(4)));

static constexpr int FOLD_TROW = 272;            // bytes per table row: 32 complex64 + 16 B pad (conflict-free ds_read_b128)
static constexpr float FOLD_B_SCALE = 64.0f;     // 2^6 on the unit-modulus E2' operand
static constexpr int FOLD_MAX_M = 128;          // u8 element indices, rowsrc packing (128 x 272 < 65536)
static constexpr int FOLD_SHARED_FROM = 33;     // antenna pairs from which the four waves of a workgroup share one set of tables
// LDS a workgroup asks for at least: 5 x this exceeds the 160 KiB of a CU, so at most FOUR workgroups (four waves per
// SIMD) are ever resident.  Every non-reproducible build of round 2 ran at five; the
// register count alone (tests/test_isa_lint.py) must not be what keeps the kernel out of that regime.
static constexpr size_t FOLD_MIN_LDS = 160 * 1024 / 5 + 64;

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
    int adaptive;      // 1 = a weak last K-step may take one product term (dmx_params.flags & DMX_FLAG_ADAPTIVE_TERMS)
    int k_tail;        // K - 16*(nblk-1): valid subcarriers of the last block (1..16)
    int mlog;          // log2(M) when M is a power of two (row offsets of the stores are then scalar arithmetic), else -1
};

typedef float fv2 __attribute__((ext_vector_type(2)));

// x = hi + lo in f16 (B' fragments, once per user).  The residual x - hi is ONE mixed-precision fma per value
// (v_fma_mix_f32 reads the f16 half of the packed register in place), selected by the compiler from
// fma(f16 -> f32, m1, x) with m1 = WsView::neg_one (k2_mfma_frag.h).
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

// max over the 32 path lanes (both half-waves hold the same values): four DPP steps give every lane its 16-lane row's
// maximum, two lane reads combine the rows - 7 vector instructions and no LDS round trip (five ds_bpermute steps before)
__device__ __forceinline__ float fold_max_paths(float v) {
    int x = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false)));    // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)));    // quad_perm [2,3,0,1]
    x = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false)));   // row_half_mirror
    x = __builtin_bit_cast(int, v);
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false)));   // row_mirror
    x = __builtin_bit_cast(int, v);
    return fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16)));
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

// ---------------------------------------------------------------------------------------------------------------------
// The tile loop.  Round 2 went through four builds of it that were not bit-reproducible (two identical launches differed
// in rows 16..31 of one 32-row tile, 5e-4 ... 1e-1 of the user's peak, once per 200 ... 10^9 users; DESIGN.md section 4 has
// the table of those builds' machine code, tools/isa_incident_table.py regenerates it).  The cause was never pinned to one
// instruction pair, so this loop is written so that NONE of the candidate mechanisms can occur, and
// tests/test_isa_lint.py checks the built code object for each of them:
//   * result read: the accumulators pass through `s_nop 15` (16 wait states; the hazard table's - and the probe's - 12
//     plus 4) as operands of the asm statement, so no read can be scheduled in front of it, and everything that reads
//     them is ordered behind it by data dependence;
//   * operands of an issued MFMA: the loads of the NEXT K-step (or the next tile's first) are issued BEFORE a K-step's
//     MFMAs, into registers of their own (v[112:127], by name), and the first vector instruction that can write a
//     register of a K-step's A' operands is the tenth of the next K-step's statement (a wait and eight products first);
//     the row-offset table is read only behind the result guard;
//   * accumulation chains: even and odd K-steps run into two accumulators, so an MFMA group accumulates onto the group
//     two K-steps back (>= 50 wait states), never onto the one just issued (the failing builds: dependent groups 27-43
//     wait states apart on one accumulator; this arrangement did not differ in 1.4e9 user-launches on a box where the
//     one-accumulator build differed 14 times in 6e8);
//   * residency: at most four waves per SIMD (FOLD_MIN_LDS): all failing builds ran at five.
// The vector work of a K-step and of the epilogue is `asm volatile`: the compiler allocates the registers, but the order
// is the one written here - no dependent pair is adjacent and nothing of a later K-step is threaded between the MFMAs.

// One K-step of a lane's table rows - 4 entries of Ac[p][.] (x0, x1) and of E1[a][.] (y0, y1), 16 registers - lives in
// v[112:127] BY NAME: the asm statements below load, multiply and split it in place (the halves of a 64-bit product
// cannot be named through an operand), and the variables are tied to those registers at every statement so that the
// compiler keeps them free in between.  v[108:111] are scratch inside a statement (clobbers).  The kernel is compiled
// for 128 registers (__launch_bounds__(256, 4)).
typedef float ff4 __attribute__((ext_vector_type(4)));
struct FoldX { ff4 x0, x1, y0, y1; };
#define FOLD_X_IO(X) "+{v[112:115]}"((X).x0), "+{v[116:119]}"((X).x1), "+{v[120:123]}"((X).y0), "+{v[124:127]}"((X).y1)

// LDS byte address (the DS instructions' operand) of a pointer into the dynamic shared memory
__device__ __forceinline__ uint32_t fold_lds_addr(const void* p) {
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// the first K-step of a tile sequence: loads only
__device__ __forceinline__ void fold_load0(FoldX& X, uint32_t ar, uint32_t er) {
    asm volatile(
        "ds_read_b128 v[112:115], %4\n\t"
        "ds_read_b128 v[116:119], %4 offset:16\n\t"
        "ds_read_b128 v[120:123], %5\n\t"
        "ds_read_b128 v[124:127], %5 offset:16"
        : "={v[112:115]}"(X.x0), "={v[116:119]}"(X.x1), "={v[120:123]}"(X.y0), "={v[124:127]}"(X.y1)
        : "v"(ar), "v"(er));
}

// the loads of the last statement have landed (the compiler does not count a statement's memory operations)
__device__ __forceinline__ void fold_drain(FoldX& X) { asm volatile("s_waitcnt lgkmcnt(0)" : FOLD_X_IO(X)); }

// One K-step: wait for its table entries, A'[(a,p)][l] = Ac[p][l] E1[a][l] for the lane's four paths in place, the f16
// hi / lo split, and the loads of the NEXT K-step (rows at ar / er, byte offset OFF) - 24 vector instructions (8 packed
// complex-product halves, 4 + 4 conversions, 8 residuals), no moves, no padding, in ONE statement.  (One instruction per
// statement: the hazard recognizer pads every statement that reads what the statement right in front of it wrote, 4
// s_nop per K-step.)  Inside the statement no instruction reads the result of the one before it.
// (a + jb)(c + jd): u = (b d, b c), then z = (a c - u.x, a d + u.y).
#define FOLD_CMUL_TEXT                                                                                               \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                       \
    "v_pk_mul_f32 v[108:109], v[112:113], v[120:121] op_sel:[1,1] op_sel_hi:[1,0]\n\t"                                \
    "v_pk_mul_f32 v[110:111], v[114:115], v[122:123] op_sel:[1,1] op_sel_hi:[1,0]\n\t"                                \
    "v_pk_fma_f32 v[112:113], v[112:113], v[120:121], v[108:109] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"                \
    "v_pk_fma_f32 v[114:115], v[114:115], v[122:123], v[110:111] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"                \
    "v_pk_mul_f32 v[108:109], v[116:117], v[124:125] op_sel:[1,1] op_sel_hi:[1,0]\n\t"                                \
    "v_pk_mul_f32 v[110:111], v[118:119], v[126:127] op_sel:[1,1] op_sel_hi:[1,0]\n\t"                                \
    "v_pk_fma_f32 v[116:117], v[116:117], v[124:125], v[108:109] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"                \
    "v_pk_fma_f32 v[118:119], v[118:119], v[126:127], v[110:111] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
#define FOLD_CVT_HI_TEXT                                                                                             \
    "v_cvt_pkrtz_f16_f32 %0, v112, v113\n\t"                                                                         \
    "v_cvt_pkrtz_f16_f32 %1, v114, v115\n\t"                                                                         \
    "v_cvt_pkrtz_f16_f32 %2, v116, v117\n\t"                                                                         \
    "v_cvt_pkrtz_f16_f32 %3, v118, v119\n\t"
template <bool LO, int OFF>
__device__ __forceinline__ void fold_kstep(FoldX& X, float m1, uint32_t ar, uint32_t er, fh8& Ah, fh8& Al) {
    unsigned h0, h1, h2, h3;
    if constexpr (LO) {
        unsigned l0, l1, l2, l3;
        asm volatile(
            FOLD_CMUL_TEXT
            FOLD_CVT_HI_TEXT
            "v_fma_mix_f32 v112, %0, %14, v112 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 v113, %0, %14, v113 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 v114, %1, %14, v114 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 v115, %1, %14, v115 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 v116, %2, %14, v116 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 v117, %2, %14, v117 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 v118, %3, %14, v118 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mix_f32 v119, %3, %14, v119 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_cvt_pkrtz_f16_f32 %4, v112, v113\n\t"
            "v_cvt_pkrtz_f16_f32 %5, v114, v115\n\t"
            "v_cvt_pkrtz_f16_f32 %6, v116, v117\n\t"
            "v_cvt_pkrtz_f16_f32 %7, v118, v119\n\t"
            "ds_read_b128 v[112:115], %12 offset:%15\n\t"
            "ds_read_b128 v[116:119], %12 offset:%16\n\t"
            "ds_read_b128 v[120:123], %13 offset:%15\n\t"
            "ds_read_b128 v[124:127], %13 offset:%16"
            : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), FOLD_X_IO(X)
            : "v"(ar), "v"(er), "s"(m1), "i"(OFF), "i"(OFF + 16)
            : "v108", "v109", "v110", "v111");
        Al = __builtin_bit_cast(fh8, fu4{l0, l1, l2, l3});
    } else {
        asm volatile(
            FOLD_CMUL_TEXT
            FOLD_CVT_HI_TEXT
            "ds_read_b128 v[112:115], %8 offset:%10\n\t"
            "ds_read_b128 v[116:119], %8 offset:%11\n\t"
            "ds_read_b128 v[120:123], %9 offset:%10\n\t"
            "ds_read_b128 v[124:127], %9 offset:%11"
            : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), FOLD_X_IO(X)
            : "v"(ar), "v"(er), "i"(OFF), "i"(OFF + 16)
            : "v108", "v109", "v110", "v111");
    }
    Ah = __builtin_bit_cast(fh8, fu4{h0, h1, h2, h3});
    if constexpr (!LO) Al = Ah;                              // never used as an operand (weak last K-step)
}
#undef FOLD_CMUL_TEXT
#undef FOLD_CVT_HI_TEXT

// Byte offset, inside the chunk's output window, of the row whose index bits are `r` (r = a * M + p for M = 2^mlog):
// (p K + 16 a) * 8.  The row index of accumulator register i in tile rt is 32 rt + (i & 3) + 8 (i >> 2) + 4 (lane >> 5):
// the four terms occupy disjoint bits, so the offset is the SUM of four terms' offsets - a per-tile scalar, two sets of
// four kernel constants, and a per-lane constant.
__device__ __forceinline__ uint32_t fold_rowbits_off(uint32_t r, int mlog, uint32_t K8) {
    return (r & ((1u << mlog) - 1u)) * K8 + ((r >> mlog) << 7);
}

// Un-scaled tile -> 16 stores.  The accumulators are read by asm statements only (behind the guard): the hazard
// recognizer does not look into them, and it does not have to.
template <int NS>
__device__ __forceinline__ void fold_finish(ff16& acc0, ff16& acc1, fv2 osc, fv2 (&w)[8]) {
    if constexpr (NS > 1) {
        asm volatile("s_nop 15" : "+v"(acc0), "+v"(acc1));
        fv2 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const fv2 a = {acc0[2 * i], acc0[2 * i + 1]}, b = {acc1[2 * i], acc1[2 * i + 1]};
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(w[i]) : "v"(t[i]), "v"(osc));
    } else {
        asm volatile("s_nop 15" : "+v"(acc0));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const fv2 a = {acc0[2 * i], acc0[2 * i + 1]};
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(w[i]) : "v"(a), "v"(osc));
        }
    }
}

// K-steps S ... NS-1 of a tile (compile-time recursion: the byte offset of the next K-step's loads is an immediate of
// the asm statement).  Even K-steps accumulate into acc0, odd ones into acc1.  Operands of K-step S, and the loads of
// K-step S + 1 - of the next tile's first K-step after the last one (rows at arn / ern) - issued in front of its MFMAs.
// (K-steps 0, 1 and 2, 3 as two back-to-back runs of six MFMAs - no dependent MFMA anywhere but right behind its
// producer - was built too: two live operand sets need 16 registers more than the 128 of four waves per SIMD leave,
// and the compiler then reloads a B' fragment from scratch between the MFMAs.)
template <int S, int NS, bool LW>
__device__ __forceinline__ void fold_ksteps(FoldX& X, float m1, uint32_t ar, uint32_t er, uint32_t arn, uint32_t ern,
                                            ff16& acc0, ff16& acc1, const fh8 (&Bhi)[4], const fh8 (&Blo)[4]) {
    constexpr bool weak = LW && S == NS - 1;
    fh8 Ah, Al;
    if constexpr (S == NS - 1) fold_kstep<!weak, 0>(X, m1, arn, ern, Ah, Al);
    else fold_kstep<true, (S + 1) * 64>(X, m1, ar, er, Ah, Al);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bhi[S], acc1, 0, 0, 0);
        if constexpr (!weak) {
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Blo[S], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bhi[S], acc1, 0, 0, 0);
        }
    } else {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bhi[S], acc0, 0, 0, 0);
        if constexpr (!weak) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Blo[S], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bhi[S], acc0, 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S + 1 < NS) fold_ksteps<S + 1, NS, LW>(X, m1, ar, er, arn, ern, acc0, acc1, Bhi, Blo);
}

struct FoldStoreCtx {
    __amdgpu_buffer_rsrc_t orsrc;
    uint32_t lane_col;         // byte offset of the lane's column inside a row segment
    uint32_t lane_row;         // SROW: lane_col + the offset of the lane's half-wave row term (4 (lane >> 5))
    uint32_t lmask_last;       // table path: lanes past K in a partial last block keep bit 31 of the row offset
    uint32_t K8;               // K * 8
    int mlog;
    int nb;                    // blocks of this chunk
    int lane_hh;               // lane >> 5
    int tail_lane_dead;        // this lane's subcarrier is past K in the partial last block
    bool masked;               // the chunk ends with a partial last block
};

// The row tiles of one inner chunk for a fixed (NS, LW).  SROW: M is a power of two and the 16 row offsets of a tile are
// scalar arithmetic (no table read, no vector add per store); tiles that hold rows past the chunk or the partial last
// block take the checked form at the end.  Otherwise the offsets come from the row table, read behind the result guard.
template <bool NT, int WS, int NS, bool LW, bool SROW>
__device__ __forceinline__ void fold_tiles(int sub, int ntiles, const uint32_t* rowsrc, const uint32_t* rowoff, int lp, int hh,
                                           const unsigned char* Ac, const unsigned char* E1, const fh8 (&Bhi)[4], const fh8 (&Blo)[4],
                                           float m1, fv2 osc, const FoldStoreCtx& c) {
    if (sub >= ntiles) return;
    constexpr int AUX = NT ? 2 : 0;
    // first tile that needs per-row checks (SROW): the one holding the last block when that block is partial, else the
    // last tile when the chunk's rows do not fill it
    const int rows = c.nb << (SROW ? c.mlog : 0);
    const int rt_chk = !SROW ? 0 : (c.masked ? (((c.nb - 1) << c.mlog) >> 5) : ((rows & 31) ? ntiles - 1 : ntiles));
    uint32_t P[4], D[4];
    if constexpr (SROW) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { P[q] = fold_rowbits_off((uint32_t)q, c.mlog, c.K8); D[q] = fold_rowbits_off((uint32_t)(8 * q), c.mlog, c.K8); }
    }
    const uint32_t ac_lds = fold_lds_addr(Ac) + (uint32_t)hh * 32u, e1_lds = fold_lds_addr(E1) + (uint32_t)hh * 32u;
    uint32_t src = rowsrc[(sub << 5) + lp];
    uint32_t ar = ac_lds + (src >> 16), er = e1_lds + (src & 0xFFFFu);
    FoldX X;
    fold_load0(X, ar, er);
    for (int rt = sub; rt < ntiles; rt += WS) {
        const int rtn = rt + WS < ntiles ? rt + WS : rt;                // past the end: the same rows again (never used)
        const uint32_t srcn = rowsrc[(rtn << 5) + lp];
        const uint32_t arn = ac_lds + (srcn >> 16), ern = e1_lds + (srcn & 0xFFFFu);
        ff16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        fold_ksteps<0, NS, LW>(X, m1, ar, er, arn, ern, acc0, acc1, Bhi, Blo);
        ar = arn; er = ern;
        fv2 w[8];
        fold_finish<NS>(acc0, acc1, osc, w);
        // register i of the tile is row (i & 3) + 8 (i >> 2) + 4 (lane >> 5), column lane & 31
        if constexpr (SROW) {
            const uint32_t T = fold_rowbits_off((uint32_t)rt << 5, c.mlog, c.K8);
            if (rt < rt_chk) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[i >> 1][i & 1]), c.orsrc, c.lane_row, T + P[i & 3] + D[i >> 2], AUX);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int a_blk = ((rt << 5) + (i & 3) + 8 * (i >> 2) + 4 * c.lane_hh) >> c.mlog;
                    const bool dead = a_blk >= c.nb || (c.masked && a_blk == c.nb - 1 && c.tail_lane_dead);
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[i >> 1][i & 1]), c.orsrc, dead ? 0x80000000u : c.lane_row,
                                                          T + P[i & 3] + D[i >> 2], AUX);
                }
            }
        } else {
            const uint4* ro4 = reinterpret_cast<const uint4*>(rowoff + (rt << 5) + 4 * hh);
            if (c.masked) {                                            // wave-uniform: partial last block in this chunk
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint4 ro = ro4[2 * g];
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g][0]), c.orsrc, c.lane_col + (ro.x & c.lmask_last), 0, AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g][1]), c.orsrc, c.lane_col + (ro.y & c.lmask_last), 0, AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g + 1][0]), c.orsrc, c.lane_col + (ro.z & c.lmask_last), 0, AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g + 1][1]), c.orsrc, c.lane_col + (ro.w & c.lmask_last), 0, AUX);
                }
            } else {                                                   // rows past the item carry 0xC0000000: out of range as they are
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint4 ro = ro4[2 * g];
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g][0]), c.orsrc, c.lane_col + ro.x, 0, AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g][1]), c.orsrc, c.lane_col + ro.y, 0, AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g + 1][0]), c.orsrc, c.lane_col + ro.z, 0, AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(fold_bits(w[2 * g + 1][1]), c.orsrc, c.lane_col + ro.w, 0, AUX);
                }
            }
        }
    }
    fold_drain(X);                                                      // the last statement's loads (never used) have landed
}

// one loop nest per (K-steps, weak last step): the choice is made out here, so that inside a loop every tile is the same
// straight-line code from its first LDS read to its last store
template <bool NT, int WS, bool SROW>
__device__ __forceinline__ void fold_tiles_kind(int tile_kind, int sub, int ntiles, const uint32_t* rowsrc, const uint32_t* rowoff, int lp, int hh,
                                                const unsigned char* Ac, const unsigned char* E1, const fh8 (&Bhi)[4], const fh8 (&Blo)[4],
                                                float m1, fv2 osc, const FoldStoreCtx& c) {
    switch (tile_kind) {
        case 2: fold_tiles<NT, WS, 1, false, SROW>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, m1, osc, c); break;
        case 4: fold_tiles<NT, WS, 2, false, SROW>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, m1, osc, c); break;
        case 5: fold_tiles<NT, WS, 2, true, SROW>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, m1, osc, c); break;
        case 6: fold_tiles<NT, WS, 3, false, SROW>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, m1, osc, c); break;
        case 7: fold_tiles<NT, WS, 3, true, SROW>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, m1, osc, c); break;
        case 8: fold_tiles<NT, WS, 4, false, SROW>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, m1, osc, c); break;
        default: fold_tiles<NT, WS, 4, true, SROW>(sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, m1, osc, c); break;
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
        // one work item per user up to 1024 subcarriers: no 64-bit division (~80 instructions) per item
        const int64_t ul = a.nsuper == 1 ? item : item / a.nsuper;
        const int sc = a.nsuper == 1 ? 0 : (int)(item - ul * a.nsuper);
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
        const float m = fold_max_paths(fmaxf(fabsf(cr), fabsf(ci)));
        // opt-in adaptive precision as in k2_channel_fd_mfma.hip (stage_item): a last K-step whose paths are all <= 2^-11 of
        // the strongest one in amplitude is multiplied out in one f16 term (here: A'hi only is built for it)
        bool last_weak = false;
        if (a.adaptive) {                                                  // kernel-uniform
            const int l0w = ((n_act - 1) >> 3) << 3;
            const float a2 = fmaf(cr, cr, ci * ci);
            const float m2 = fold_max_paths(a2), mw2 = fold_max_paths(lp >= l0w ? a2 : 0.f);
            last_weak = l0w >= 8 && mw2 * 4194304.0f <= m2;
        }
        int e;
        (void)frexpf(m, &e);                                               // m = f * 2^e, f in [0.5, 1)
        const float gs = ldexpf(1.0f, 10 - e);                             // max |c| component -> [512, 1024)
        const float oscale = ldexpf(1.0f, e - 10 - 6);                     // 1 / (gs * FOLD_B_SCALE)
        const fv2 osc = {oscale, oscale};
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
        // exact in float32 and qh * (sc - sc mod 4096) is an integer (k2_channel_fd_mfma.hip gen_b_step).  A lane walks the
        // blocks ab, ab + 2 WS, ...: every eighth entry is evaluated like that, the seven behind it by ONE complex
        // multiplication with the step phasor W = e^{-j 2pi q_l 32 WS d} (4 instead of ~12 vector instructions per entry;
        // <= 7 float32 roundings of ~6e-8 on top of the hardware sin / cos's 1.25e-7).
        {
            const int dsc = a.sc_stride * 32 * WS;
            const float pw = qhf * (float)(dsc & 4095);
            float ws_, wc_;
            sincos_rev(fmaf(qlf, (float)dsc, pw - rintf(pw)), ws_, wc_);
            const float wr = wc_, wi = -ws_;
            float er_ = 1.f, ei_ = 0.f;
            int it = 0;
            for (int ab = 2 * sub + hh; ab < nb; ab += 2 * WS, ++it) {
                if ((it & 7) == 0) {                                       // wave-uniform
                    const int sca = a.sc_first + a.sc_stride * 16 * (a0 + ab);
                    const float p1 = qhf * (float)(sca & 4095);
                    float s, c;
                    sincos_rev(fmaf(qlf, (float)sca, p1 - rintf(p1)), s, c);
                    er_ = c; ei_ = -s;
                } else {
                    const float t = fmaf(-ei_, wi, er_ * wr);
                    ei_ = fmaf(ei_, wr, er_ * wi);
                    er_ = t;
                }
                *reinterpret_cast<float2*>(E1 + (size_t)ab * FOLD_TROW + lp * 8) = make_float2(er_, ei_);
            }
        }
        fold_sync<WS>();

        // ---- row tiles: 32 rows (a,p) each
        const uint32_t* rowoff = last ? rowoff1 : rowoff0;
        const int rows = nb * M;
        const int ntiles = (rows + 31) >> 5;
        FoldStoreCtx sc_;
        sc_.orsrc = __builtin_amdgcn_make_buffer_rsrc(o, 0, (int)((unsigned)(user_floats - (size_t)a0 * 32) * 4u), 0x00020000);
        sc_.lane_col = lane_col;
        sc_.K8 = (uint32_t)K * 8u;
        sc_.mlog = a.mlog;
        sc_.lane_row = lane_col + (a.mlog >= 0 ? (uint32_t)hh * fold_rowbits_off(4u, a.mlog, (uint32_t)K * 8u) : 0u);
        sc_.lmask_last = lmask_last;
        sc_.nb = nb;
        sc_.lane_hh = hh;
        sc_.tail_lane_dead = bsc >= a.k_tail;
        sc_.masked = last && a.k_tail < 16;
        if (a.mlog >= 0) fold_tiles_kind<NT, WS, true>(tile_kind, sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, osc, sc_);
        else fold_tiles_kind<NT, WS, false>(tile_kind, sub, ntiles, rowsrc, rowoff, lp, hh, Ac, E1, Bhi, Blo, ws.neg_one, osc, sc_);
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
    if (chunk_blocks <= 0) {                       // tuning build only: DMX_FOLD_CHUNK=8|16|32 overrides the LDS-occupancy rule
        chunk_blocks = tuning_int("DMX_FOLD_CHUNK", 0);
        if (chunk_blocks != 8 && chunk_blocks != 16 && chunk_blocks != 32) chunk_blocks = 0;
    }
    // table sets per workgroup: one per wave up to 32 pairs, one for the whole workgroup above (tuning build only:
    // DMX_FOLD_SHARED=0|1 forces either)
    bool shared = a.M >= FOLD_SHARED_FROM;
    if (const int f = tuning_int("DMX_FOLD_SHARED", -1); f >= 0) shared = f == 1;
    const int sets = shared ? 1 : 4;
    a.ch = fold_chunk_blocks(a.M, a.nblk, chunk_blocks, sets);
    a.sch = (FOLD_SUPER / a.ch) * a.ch;
    if (a.sch > a.nblk) a.sch = (a.nblk + a.ch - 1) / a.ch * a.ch;
    a.nsuper = (a.nblk + a.sch - 1) / a.sch;
    a.nchunk = (a.nblk + a.ch - 1) / a.ch;
    a.nb_last = a.nblk - a.ch * (a.nchunk - 1);
    a.tab_rows = (a.M * a.ch + 31) / 32 * 32;
    a.k_tail = a.K - 16 * (a.nblk - 1);
    a.adaptive = (prm.flags & DMX_FLAG_ADAPTIVE_TERMS) ? 1 : 0;     // default: three product terms everywhere
    a.mlog = -1;
    for (int b = 0; b < 8; ++b) if (a.M == (1 << b)) a.mlog = b;
    if (tuning_int("DMX_FOLD_ROWTABLE", 0) == 1) a.mlog = -1;       // tuning build only: force the row-table stores
    if ((size_t)a.M * (size_t)a.K * 8 >= (size_t)1 << 30) { set_error("%d x %d outputs per user are too many for the folded kernel", a.M, a.K); return DMX_ERR_SHAPE; }
    size_t smem = fold_static_bytes(a.tab_rows, a.M) + sets * fold_wave_bytes(a.M, a.ch);
    if (smem > 160 * 1024) { set_error("folded kernel tables of %zu bytes exceed LDS", smem); return DMX_ERR_SHAPE; }
    if (smem < FOLD_MIN_LDS) smem = FOLD_MIN_LDS;                   // at most four workgroups per CU (see FOLD_MIN_LDS)
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
