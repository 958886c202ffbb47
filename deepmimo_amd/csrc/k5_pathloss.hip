// Fused consumer of the ray records: Dataset.compute_pathloss
// (deepmimo/generator/dataset.py:541-566) - per user the (coherent or amplitude) sum of the path gains
//   g_l = sqrt(10^(p_l/10)) [* exp(j deg2rad(phase_l))],   PL = -10 log10 |sum_l g_l|^2   (NaN where the sum is 0)
// over ALL loaded paths, NaN paths skipped (np.nansum), float32 / complex64 arithmetic as the reference.
// One 32-lane group per user (lane = path, same mapping as k1_path_prep), butterfly reduction.
#include "dmx_common.h"

namespace dmx {

__global__ __launch_bounds__(256) void k5_pathloss(dmx_rays r, int coherent, float* __restrict__ out) {
    const int lane = threadIdx.x & 31;
    const int64_t u = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    const bool u_ok = u < r.n_ue;
    const size_t row = (size_t)(u_ok ? u : 0) * (size_t)r.ld;
    float sr = 0.f, si = 0.f;
    for (int j = lane; j < r.n_paths; j += 32) {
        const float p = u_ok ? r.power[row + j] : __int_as_float(0x7fc00000);
        const float amp = sqrtf(exp10f(p / 10.0f));
        float gr = amp, gi = 0.f;
        if (coherent) {
            float s, c;
            sincosf(r.phase[row + j] * 0.017453292519943295f, &s, &c);
            gr = amp * c; gi = amp * s;
        }
        if (!(isnan(gr) || isnan(gi))) { sr += gr; si += gi; }           // np.nansum on complex64
    }
    for (int off = 16; off > 0; off >>= 1) { sr += __shfl_xor(sr, off, 32); si += __shfl_xor(si, off, 32); }
    if (lane == 0 && u_ok) {
        const float a = hypotf(sr, si);
        const float tp = a * a;
        out[u] = tp > 0.f ? -10.0f * log10f(tp) : __int_as_float(0x7fc00000);
    }
}

}  // namespace dmx

using namespace dmx;

extern "C" int dmx_pathloss(const dmx_rays* rays, int32_t coherent, float* out, void* stream) {
    if (!rays || rays->n_ue < 0 || rays->n_paths < 0 || rays->ld < rays->n_paths) { set_error("bad ray matrix shape"); return DMX_ERR_ARG; }
    if (rays->n_ue == 0) return DMX_OK;
    if (!out || (rays->n_paths > 0 && (!rays->power || !rays->phase))) { set_error("power/phase/out is NULL"); return DMX_ERR_ARG; }
    const int64_t blocks = (rays->n_ue + 7) / 8;
    if (blocks > 0x7fffffffLL) { set_error("too many users for one call"); return DMX_ERR_SHAPE; }
    hipLaunchKernelGGL(k5_pathloss, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *rays, (int)coherent, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k5_pathloss launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}
