// Fragment helpers shared by the matrix-core kernels of stage 2 (k2_channel_fd_mfma.hip, k2c_beam_power.hip): the f16
// hi/lo split and the per-lane generation of the B' operand (subcarrier phasors) in MFMA register layout.
#pragma once
#include "dmx_common.h"

namespace dmx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

static constexpr int ROW_BYTES = 144;        // 64 f16 + 16 B pad
static constexpr int MAX_ROWS = 256;         // antenna pairs per workgroup
static constexpr int MFMA_LDS_MAX = 80 * 1024;   // two workgroups per CU
static constexpr int LPAD = 32;              // path slots (kk = 64)
static constexpr float A_SCALE = 64.0f;      // 2^6

// Wait states the compiler does not know about, right behind the last MFMA of a chain, on top of the 12 the hazard
// recognizer puts in front of the first vector read of its accumulator (it counts this statement as ONE of its 12, so
// `s_nop 4` = five leaves 16 in all) (12 is also what a direct probe needs for the registers
// written in the last pass: profiles/r2_mfma_hazard_probes.txt item 3, i.e. the recognizer's figure has no margin).
// The accumulator IS AN OPERAND of the statement ("+v"): every read of it is data-dependent on the asm and cannot be
// scheduled in front of it.  (Round 2 had `asm volatile("s_nop 3")` without operands; the compiler moved 15 of 16
// accumulator reads across it - VERDICT r2.)  tests/test_isa_lint.py measures the result in the built code object:
// >= 16 wait states in front of the first read of any MFMA result in every kernel of this library.
#define DMX_MFMA_RESULT_GUARD(acc) asm volatile("s_nop 4" : "+v"(acc))
#define DMX_MFMA_RESULT_GUARD2(acc_a, acc_b) asm volatile("s_nop 4" : "+v"(acc_a), "+v"(acc_b))
// The same for accumulators the register allocator may keep in AGPRs (a "+v" operand would make it copy them out IN
// FRONT of the statement): no operands, fenced on both sides against the schedulers instead.
#define DMX_MFMA_RESULT_GUARD_FENCED() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 4"); __builtin_amdgcn_sched_barrier(0); } while (0)

// (x0, x1) -> packed hi pair and packed lo pair, x = hi + lo.  v_cvt_pkrtz_f16_f32 converts two floats per
// instruction; with round-toward-zero x - hi is exact in fp32 and lo keeps 11 more bits of it.  The residual is ONE
// v_fma_mix_f32 per value (it reads the f16 half of the packed register directly: op_sel_hi marks the operand as f16,
// op_sel picks the half); the compiler's own form is v_cvt_f32_f16 + v_sub_f32.  Same-box A/B at the headline shape
// (tools/ab_two_libs.sh): 15.80 vs 16.21 ms.
typedef __fp16 hp2 __attribute__((ext_vector_type(2)));
// m1 = WsView::neg_one, i.e. -1.0f arriving as a kernel argument: fma(f16 -> f32, m1, x) is selected as ONE
// v_fma_mix_f32 reading the f16 half in place, while with a literal -1.0f (or a constant the compiler can see) the
// expression is canonicalised to x - h and becomes v_cvt_f32_f16 + v_sub_f32.  (Until late in round 2 this was inline
// asm; compiler-selected instructions are visible to the hazard recognizer, asm operands are not - no inline asm next
// to MFMAs.)
__device__ __forceinline__ void split2_f16(float x0, float x1, h2& hi, h2& lo, float m1) {
    const hp2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const h2 hh = __builtin_bit_cast(h2, h);
    const float r0 = __builtin_fmaf((float)hh[0], m1, x0);
    const float r1 = __builtin_fmaf((float)hh[1], m1, x1);
    const hp2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi = hh;
    lo = __builtin_bit_cast(h2, l);
}

// B' fragments of one lane for strip `strip` (32 columns = 16 subcarriers, re/im interleaved): element j of
// K-step s is row kk = 16s + 8h + j of B', i.e. path l = 8s + 4h + (j>>1), component j&1.
struct BLane {                      // what a lane knows about its column of the strip
    int kidx, c;                    // subcarrier index in the selection, re/im column
    bool kok;                       // column inside the selection
    unsigned lane_off;              // byte offset of (row 4h, this column) inside a 32-row tile of the output
    float kl, kf;                   // selected subcarrier number: its low 12 bits (exact) and the whole (both as float)
};

__device__ __forceinline__ BLane b_lane(int strip, int col, int hh, size_t twoK, const int32_t* __restrict__ sc) {
    BLane b;
    const int ncol = (strip << 5) + col;                                // column of C = 2*kidx + c
    b.kidx = ncol >> 1; b.c = ncol & 1;
    b.kok = (size_t)ncol < twoK;
    b.lane_off = ((unsigned)(4 * hh) * (unsigned)twoK + (unsigned)ncol) * 4u;
    const int kki = b.kok ? sc[b.kidx] : 0;
    b.kl = (float)(kki & 4095);
    b.kf = (float)kki;
    return b;
}

// one K-step (8 paths) of the strip's B' fragments
__device__ __forceinline__ void gen_b_step(float m1, int s, const BLane& bl, int hh, int n_act, const float2* qtab, const float* crtab,
                                           const float* citab, const float2* __restrict__ grow, int K, float gs,
                                           h8& Bhi, h8& Blo) {
    Bhi = h8{0, 0, 0, 0, 0, 0, 0, 0};
    Blo = Bhi;
    if (8 * s >= n_act) return;
    const int c = bl.c;
    // this lane evaluates paths jj = 2c, 2c+1 of the step, its pair lane (same subcarrier, other re/im column)
    // the other two; swap through a lane-pair shuffle
    float mr[2], mi[2], orr[2], oi[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int pl = 8 * s + 4 * hh + 2 * c + t;
        if (grow) {                                                     // rx_filter: G[l,k] from the k3 table
            float2 g = make_float2(0.f, 0.f);
            if (bl.kok && pl < n_act) g = grow[(size_t)pl * K + bl.kidx];
            mr[t] = g.x * gs; mi[t] = g.y * gs;
        } else {
            float sn, cs;
            // Phase dn_l sc_k / N in revolutions without float64: q = dn/N is held as qh + ql with qh a multiple of
            // 2^-12 in [0, 1], so qh * (k mod 4096) is exact in float32 and qh * (k - k mod 4096) is an integer
            // (drops out); ql <= 2^-13 carries the rest (its product is rounded at 2^-24 of a value below one).
            const float2 q = qtab[pl];
            const float p1 = q.x * bl.kl;
            sincos_rev(fmaf(q.y, bl.kf, p1 - rintf(p1)), sn, cs);
            const float cr = crtab[pl], ci = citab[pl];
            mr[t] = cr * cs + ci * sn;                                  // Re c*exp(-j x)
            mi[t] = ci * cs - cr * sn;                                  // Im
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) { orr[t] = __shfl_xor(mr[t], 1); oi[t] = __shfl_xor(mi[t], 1); }
    float gr[4], gi[4];
    gr[0] = c ? orr[0] : mr[0]; gi[0] = c ? oi[0] : mi[0];
    gr[1] = c ? orr[1] : mr[1]; gi[1] = c ? oi[1] : mi[1];
    gr[2] = c ? mr[0] : orr[0]; gi[2] = c ? mi[0] : oi[0];
    gr[3] = c ? mr[1] : orr[1]; gi[3] = c ? mi[1] : oi[1];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const float e0 = c ? gi[jj] : gr[jj];                           // row 2l   : Re G (re col) / Im G (im col)
        const float e1 = c ? gr[jj] : -gi[jj];                          // row 2l+1 : -Im G        / Re G
        h2 ph, pl2;
        split2_f16(e0, e1, ph, pl2, m1);
        Bhi[2 * jj] = ph[0]; Bhi[2 * jj + 1] = ph[1];
        Blo[2 * jj] = pl2[0]; Blo[2 * jj + 1] = pl2[1];
    }
}

// One K-step of the strip's B' fragments from the PACKED gains table the rx_filter path's FFT kernel writes
// (k3_lpf_fft_wave, pack = 1): entry (l, k) is {f16 pair (re, im) of hi, f16 pair (re, im) of lo} of G[l,k] already scaled
// by the user's power of two - no scaling, no split, no lane-pair exchange here: a column c = 0 lane wants
// (Re, -Im) = the pair with the upper sign bit flipped, a c = 1 lane wants (Im, Re) = the pair rotated by 16 bits.
__device__ __forceinline__ void load_b_step_packed(int s, const BLane& bl, int hh, int n_act, const uint2* __restrict__ prow, int K,
                                                   h8& Bhi, h8& Blo) {
    Bhi = h8{0, 0, 0, 0, 0, 0, 0, 0};
    Blo = Bhi;
    if (8 * s >= n_act) return;
    const unsigned rot = bl.c ? 16u : 0u, sgn = bl.c ? 0u : 0x80000000u;
    uint2 g[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int pl = 8 * s + 4 * hh + jj;
        g[jj] = make_uint2(0u, 0u);
        if (bl.kok && pl < n_act) g[jj] = prow[(size_t)pl * K + bl.kidx];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const unsigned xh = __builtin_amdgcn_alignbit(g[jj].x, g[jj].x, rot) ^ sgn;
        const unsigned xl = __builtin_amdgcn_alignbit(g[jj].y, g[jj].y, rot) ^ sgn;
        const h2 ph = __builtin_bit_cast(h2, xh), pl2 = __builtin_bit_cast(h2, xl);
        Bhi[2 * jj] = ph[0]; Bhi[2 * jj + 1] = ph[1];
        Blo[2 * jj] = pl2[0]; Blo[2 * jj + 1] = pl2[1];
    }
}

// k2b_beam_project's output inside the beam workspace
struct BeamTabs {
    const float2* ftab;      // [user_count, n_beams, P]  f[b,l] = sum_tx F[b,tx] a_tx[tx,l]
    const int32_t* fexp;     // [user_count]              exponent of max |f| per user
};

}  // namespace dmx
