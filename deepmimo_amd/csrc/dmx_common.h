// Shared definitions of the HIP side: the per-path record layout that stage 1 (k1_path_prep)
// writes and stage 2 (k2_channel_fd / k4_channel_td) reads, and small device helpers.
//
// Record layout in the caller-provided workspace (SoA, one array per field, [n_ue, P] row-major,
// P = min(params.num_paths, loaded paths)).  Paths that take part in a user's sum are COMPACTED to
// the front of the user's row; n_keep[u] says how many.  Phase steps are kept in REVOLUTIONS
// (phase / 2pi) in float64 so that element-index multiples stay exact before range reduction:
//   a_tx[m = y + Mh*z, l] = exp(j 2pi (y*tx_y[l] + z*tx_z[l]))      geometry.py:85-102
// (kd = 2pi*spacing, so kd*sin(theta)sin(phi)/2pi = spacing*sin(theta)sin(phi)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/deepmimo_amd.h"

namespace dmx {

struct WsView {
    float*   c_re;    // [n, P]  path coefficient: FD sqrt(p/N) e^{j phase} (x Doppler), TD sqrt(p) e^{j phase}
    float*   c_im;    // [n, P]
    float*   dn;      // [n, P]  normalised delay tau/Ts after the ">= N" clip (channel.py:183-189)
    double*  tx_y;    // [n, P]  revolutions per y-step of the BS panel
    double*  tx_z;    // [n, P]
    double*  rx_y;    // [n, P]
    double*  rx_z;    // [n, P]
    float*   dop_v;   // [n, P]  Doppler velocity / acceleration of the kept path (used by the rx_filter
    float*   dop_a;   //         variant only, where the Doppler phase depends on the tap index)
    int32_t* n_keep;  // [n]     compacted paths per user
    int64_t  n;
    int32_t  P;
    float    neg_one; // -1.0f as a kernel-argument SGPR the compiler cannot fold: the multiplier of the f16 split's
                      // residual fma, so that it is selected as v_fma_mix_f32 (k2_mfma_frag.h, split2_f16)
};

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Carve the workspace.  Returns total bytes; fills `v` when base != nullptr.
__host__ inline size_t ws_carve(void* base, int64_t n, int32_t P, WsView* v) {
    size_t off = 0;
    const size_t np = (size_t)n * (size_t)P;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_cre = take(np * 4), o_cim = take(np * 4), o_dn = take(np * 4);
    size_t o_txy = take(np * 8), o_txz = take(np * 8), o_rxy = take(np * 8), o_rxz = take(np * 8);
    size_t o_dv = take(np * 4), o_da = take(np * 4);
    size_t o_keep = take((size_t)n * 4);
    if (v) {
        char* b = (char*)base;
        v->c_re = (float*)(b + o_cre); v->c_im = (float*)(b + o_cim); v->dn = (float*)(b + o_dn);
        v->tx_y = (double*)(b + o_txy); v->tx_z = (double*)(b + o_txz);
        v->rx_y = (double*)(b + o_rxy); v->rx_z = (double*)(b + o_rxz);
        v->dop_v = (float*)(b + o_dv); v->dop_a = (float*)(b + o_da);
        v->n_keep = (int32_t*)(b + o_keep);
        v->n = n; v->P = P;
        v->neg_one = -1.0f;
    }
    return off;
}

// sin/cos of 2*pi*r for r already reduced to [-0.5, 0.5] revolutions.  v_sin_f32 / v_cos_f32 take
// their argument in revolutions, i.e. exactly this representation; measured max abs error on the
// reduced range (tools/sincos_acc.hip, MI355X): 1.25e-7, against 5.2e-8 for sincospif at ~15x the
// instructions.  All phase accuracy in this library lives in the float64 range reduction before.
__device__ __forceinline__ void sincos_rev(float r, float& s, float& c) {
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
}

// fractional part in [-0.5, 0.5] of a float64 phase given in revolutions
__device__ __forceinline__ float frac_rev(double t) {
    return (float)(t - rint(t));
}

void set_error(const char* fmt, ...);

// compute units of the current device (persistent grids are sized from it)
int device_cu_count();

// stage launchers (defined next to their kernels)
int launch_path_prep(const dmx_rays& rays, const dmx_params& prm, const WsView& ws, const dmx_side& side,
                     hipStream_t stream);
int launch_channels_fd(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                       float2* out, int variant, hipStream_t stream);
int fd_auto_choice(const dmx_params& prm, const WsView& ws);
int launch_channels_fd_lpf(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                           float2* gtab, float2* out, hipStream_t stream);
int launch_channels_fd_beams(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                             const float2* codebook, int n_beams, void* beam_ws, float2* out, hipStream_t stream);
size_t beam_workspace_bytes(int64_t user_count, int n_beams, int P);
int launch_beam_power(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                      const float2* codebook, int n_beams, void* beam_ws, float* out_amp, int32_t* out_best, hipStream_t stream);
bool fd_mfma_supported(const dmx_params& prm, const WsView& ws);
int launch_channels_td(const dmx_params& prm, const WsView& ws, int64_t user_begin, int64_t user_count,
                       float2* out, hipStream_t stream);

}  // namespace dmx
