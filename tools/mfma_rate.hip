// Issue rate of v_mfma_f32_32x32x16_f16 vs the legacy v_mfma_f32_32x32x8_f16 on one wave per SIMD (gfx950):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate && tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    f16v acc0 = {}, acc1 = {};
    h8 a8, b8; h4 a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(threadIdx.x * 0.001f + j); b8[j] = (_Float16)(j * 0.5f); }
    for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc1, 0, 0, 0); }
            else { acc0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc1, 0, 0, 0); }
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMallocManaged(&cyc, 16);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
    }
    printf("32x32x16_f16: %.1f clock64 ticks per MFMA   32x32x8_f16: %.1f ticks per MFMA  (one wave per SIMD, 16 MFMAs per loop)\n",
           (double)cyc[0] / (iters * 16.0), (double)cyc[1] / (iters * 16.0));
    return 0;
}
