// Stand-alone client of the C-ABI (include/deepmimo_amd.h): no Python, no PyTorch - hipMalloc'ed buffers,
// plain structs, two calls.  Generates the channels of N synthetic users (a fixed LCG, so any other client can
// reproduce the inputs bit for bit - tests/test_gpu_parity.py::test_c_abi_demo does, through the Python host)
// and prints sizes, timing and float64 fingerprints of the result as one JSON line.
//
//   hipcc --offload-arch=gfx950 -O2 examples/c_abi_demo.cpp -Iinclude -Ldeepmimo_amd/lib -ldeepmimo_amd \
//         -Wl,-rpath,'$ORIGIN' -o deepmimo_amd/lib/dmx_demo
//   ./deepmimo_amd/lib/dmx_demo [n_users]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "deepmimo_amd.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define DMX_OK_(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "dmx error %d: %s\n", rc_, dmx_last_error()); return 1; } } while (0)

static uint32_t lcg_state = 12345u;
static float uni(float lo, float hi) {                       // 24-bit uniform from a 32-bit LCG
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return lo + (hi - lo) * (float)(lcg_state >> 8) * (1.0f / 16777216.0f);
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 2000;
    const int L = 10, K = 64, M_RX = 2, M_TX = 16;
    std::vector<float> field[8];
    const float lo[8] = {-140, -180, 1e-8f, -180, 0, -180, 0, 0}, hi[8] = {-60, 180, 2e-6f, 180, 180, 180, 180, 4.999f};
    for (int f = 0; f < 8; ++f) {
        field[f].resize((size_t)n * L);
        for (auto& v : field[f]) v = uni(lo[f], hi[f]);
    }
    for (auto& v : field[7]) v = floorf(v);                  // interaction codes 0..4
    for (int64_t u = 0; u < n; u += 7)                        // every 7th user: only 3 valid paths (trailing NaN padding)
        for (int f = 0; f < 8; ++f)
            for (int l = 3; l < L; ++l) field[f][(size_t)u * L + l] = NAN;

    float* d_field[8];
    for (int f = 0; f < 8; ++f) {
        HIP_OK(hipMalloc(&d_field[f], field[f].size() * 4));
        HIP_OK(hipMemcpy(d_field[f], field[f].data(), field[f].size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<int32_t> sc(K);
    for (int k = 0; k < K; ++k) sc[k] = k;
    int32_t* d_sc;
    HIP_OK(hipMalloc(&d_sc, K * 4));
    HIP_OK(hipMemcpy(d_sc, sc.data(), K * 4, hipMemcpyHostToDevice));

    dmx_rays rays = {};
    rays.n_ue = n; rays.n_paths = L; rays.ld = L;
    rays.power = d_field[0]; rays.phase = d_field[1]; rays.delay = d_field[2]; rays.aoa_az = d_field[3];
    rays.aoa_el = d_field[4]; rays.aod_az = d_field[5]; rays.aod_el = d_field[6]; rays.inter = d_field[7];

    dmx_params prm = {};
    prm.bs_shape[0] = 8; prm.bs_shape[1] = 2; prm.ue_shape[0] = 2; prm.ue_shape[1] = 1;
    prm.bs_spacing = prm.ue_spacing = 0.5;
    prm.bs_rotation[2] = 30.0 * M_PI / 180.0;                 // radians, as np.deg2rad gives them
    prm.num_paths = 25; prm.freq_domain = 1; prm.n_subcarriers = 512; prm.n_selected = K;
    prm.selected_subcarriers = d_sc; prm.bandwidth = 10e6;
    // host-side promise about d_sc (ABI 2): sc[k] = sc_first + k * sc_stride.  It lets variant 0 take the folded
    // matrix-core kernel for this 32-pair panel; leave sc_stride = 0 for an arbitrary selection.
    prm.sc_first = 0; prm.sc_stride = 1;

    void *ws, *out;
    int32_t *d_los, *d_np;
    const size_t ws_bytes = dmx_workspace_bytes(&prm, n, L);
    HIP_OK(hipMalloc(&ws, ws_bytes));
    HIP_OK(hipMalloc(&out, (size_t)n * M_RX * M_TX * K * 8));
    HIP_OK(hipMalloc(&d_los, n * 4));
    HIP_OK(hipMalloc(&d_np, n * 4));
    dmx_side side = {};
    side.los = d_los; side.num_paths = d_np;

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    DMX_OK_(dmx_path_prep(&rays, &prm, ws, ws_bytes, &side, stream));              // warm-up pass
    DMX_OK_(dmx_channels_fd(&prm, ws, n, L, 0, n, out, 0, stream));
    HIP_OK(hipEventRecord(e0, stream));
    DMX_OK_(dmx_path_prep(&rays, &prm, ws, ws_bytes, &side, stream));
    DMX_OK_(dmx_channels_fd(&prm, ws, n, L, 0, n, out, 0, stream));
    HIP_OK(hipEventRecord(e1, stream));
    HIP_OK(hipStreamSynchronize(stream));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));

    std::vector<float> h((size_t)n * M_RX * M_TX * K * 2);
    std::vector<int32_t> los(n), np_(n);
    HIP_OK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(los.data(), d_los, n * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(np_.data(), d_np, n * 4, hipMemcpyDeviceToHost));
    double energy = 0, wsum = 0;
    for (size_t i = 0; i < h.size(); ++i) { energy += (double)h[i] * h[i]; wsum += (double)h[i] * cos(0.37 * (double)i); }
    long long los_sum = 0, np_sum = 0;
    for (int64_t u = 0; u < n; ++u) { los_sum += los[u]; np_sum += np_[u]; }
    printf("{\"abi\": %d, \"users\": %lld, \"shape\": [%lld, %d, %d, %d], \"ms\": %.4f, \"energy\": %.17g, \"wsum\": %.17g, "
           "\"los_sum\": %lld, \"num_paths_sum\": %lld}\n",
           dmx_version(), (long long)n, (long long)n, M_RX, M_TX, K, ms, energy, wsum, los_sum, np_sum);
    return 0;
}
