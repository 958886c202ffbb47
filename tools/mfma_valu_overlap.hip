// Do matrix-core (MFMA) and vector (VALU) instructions overlap on a gfx950 SIMD - inside one wave, and between waves?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
// Kernels: M = MFMA only (4 independent accumulator chains of v_mfma_f32_32x32x16_f16), V = VALU only (v_fma_f32, 8
// chains), T = transcendental only (v_sqrt_f32), MV = both in every wave (VPM vector ops per MFMA).  Waves per SIMD: 1, 2, 4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define ITER 2000

template <int MODE, int VPM>   // MODE 0 M, 1 V, 2 MV, 4 T (sqrt), 5 M + T
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f16v acc[4];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.01f * (lane + i);
    const float k1 = 0.999f, k2 = 0.001f;
    constexpr bool do_m = MODE == 0 || MODE == 2 || MODE == 5;
    constexpr bool do_v = MODE == 1 || MODE == 2;
    constexpr bool do_t = MODE == 4 || MODE == 5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (do_m) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
            if (do_v) {
#pragma unroll
                for (int j = 0; j < VPM; ++j) v[(c * VPM + j) & 7] = __builtin_fmaf(v[(c * VPM + j) & 7], k1, k2);
            }
            if (do_t) {
#pragma unroll
                for (int j = 0; j < VPM; ++j) v[(c * VPM + j) & 7] = __builtin_amdgcn_sqrtf(v[(c * VPM + j) & 7]);
            }
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int VPM>
static float run(int wps, const char* name, float* d) {
    // wps waves per SIMD: 256 CUs x wps workgroups of 4 waves
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE, VPM>), dim3(256 * wps), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, VPM>), dim3(256 * wps), dim3(256), 0, 0, d, ITER);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per loop iteration per SIMD assuming 2.4 GHz: informational only
    printf("%-34s waves/SIMD %d: %8.3f ms  (%.1f ns per iteration)\n", name, wps, ms, ms * 1e6 / ITER);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    for (int wps : {1, 2, 4}) {
        run<0, 0>(wps, "MFMA only (4 per iter)", d);
        run<1, 4>(wps, "VALU only (16 fma per iter)", d);
        run<2, 4>(wps, "MFMA + 4 fma each, same wave", d);
        run<1, 7>(wps, "VALU only (28 fma per iter)", d);
        run<2, 7>(wps, "MFMA + 7 fma each, same wave", d);
        run<4, 1>(wps, "sqrt only (4 per iter)", d);
        run<5, 1>(wps, "MFMA + 1 sqrt each, same wave", d);
        run<4, 2>(wps, "sqrt only (8 per iter)", d);
        run<5, 2>(wps, "MFMA + 2 sqrt each, same wave", d);
        printf("\n");
    }
    return 0;
}
