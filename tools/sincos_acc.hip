// Accuracy probe (GPU box only): hardware v_sin_f32 / v_cos_f32 (argument in revolutions) and
// sincospif against float64, on the reduced range [-0.5, 0.5] revolutions the kernels use.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const float* r, float* hs, float* hc, float* ps, float* pc, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hs[i] = __builtin_amdgcn_sinf(r[i]);
    hc[i] = __builtin_amdgcn_cosf(r[i]);
    float s, c;
    sincospif(2.0f * r[i], &s, &c);
    ps[i] = s; pc[i] = c;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> r(n), hs(n), hc(n), ps(n), pc(n);
    for (int i = 0; i < n; ++i) r[i] = -0.5f + (float)i / (float)(n - 1);
    // plus tiny values around 0 and the quadrant boundaries
    for (int i = 0; i < 4096; ++i) { r[i] = ldexpf(1.0f, -30 + (i % 28)) * ((i & 1) ? 1.f : -1.f); r[4096 + i] = 0.25f + (i - 2048) * 1e-8f; }
    float *d[5];
    for (auto& p : d) hipMalloc(&p, n * 4);
    hipMemcpy(d[0], r.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(d[0], d[1], d[2], d[3], d[4], n);
    hipMemcpy(hs.data(), d[1], n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), d[2], n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ps.data(), d[3], n * 4, hipMemcpyDeviceToHost); hipMemcpy(pc.data(), d[4], n * 4, hipMemcpyDeviceToHost);
    double e_hs = 0, e_hc = 0, e_ps = 0, e_pc = 0;
    for (int i = 0; i < n; ++i) {
        double x = 2.0 * M_PI * (double)r[i];
        e_hs = fmax(e_hs, fabs(hs[i] - sin(x))); e_hc = fmax(e_hc, fabs(hc[i] - cos(x)));
        e_ps = fmax(e_ps, fabs(ps[i] - sin(x))); e_pc = fmax(e_pc, fabs(pc[i] - cos(x)));
    }
    printf("max abs err  v_sin_f32 %.3e  v_cos_f32 %.3e   sincospif: sin %.3e cos %.3e\n", e_hs, e_hc, e_ps, e_pc);
    return 0;
}
