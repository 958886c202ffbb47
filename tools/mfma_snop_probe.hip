// The same question as mfma_raw_probe.hip with ONE `s_nop N` instruction (what the compiler emits) instead of N separate
// `s_nop 0`, at 1 and 4 waves per SIMD: is `s_nop N` worth N + 1 of the wait states an MFMA result needs?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define PROBE(NAME, IDLE, NOPS, REG) \
__global__ __launch_bounds__(256) void NAME(const unsigned* in, float* out, int iters) { \
    const int t = threadIdx.x & 63; \
    unsigned a0 = in[t], a1 = in[64 + t], a2 = in[128 + t], a3 = in[192 + t]; \
    unsigned b0 = in[256 + t], b1 = in[320 + t], b2 = in[384 + t], b3 = in[448 + t]; \
    float bad = 0.f, ref = 0.f; \
    for (int it = 0; it < iters; ++it) { \
        float s; \
        asm volatile( \
            "v_mov_b32 v100, %1\n v_mov_b32 v101, %2\n v_mov_b32 v102, %3\n v_mov_b32 v103, %4\n" \
            "v_mov_b32 v104, %5\n v_mov_b32 v105, %6\n v_mov_b32 v106, %7\n v_mov_b32 v107, %8\n" \
            "v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n" \
            "v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n" \
            IDLE \
            "v_mfma_f32_32x32x16_f16 v[110:125], v[100:103], v[104:107], v[110:125]\n" \
            NOPS \
            "v_mov_b32 %0, " REG "\n" \
            "s_nop 15\n s_nop 15\n s_nop 15\n" \
            : "=v"(s) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3) \
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", \
              "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125"); \
        if (it == 0) ref = s; \
        bad += (s != ref) ? 1.f : 0.f; \
    } \
    out[blockIdx.x * 256 + threadIdx.x] = bad; \
    out[gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = ref; \
}
PROBE(probe_snop8, "s_nop 7\n", "s_nop 8\n", "v125")
PROBE(probe_snop9, "s_nop 7\n", "s_nop 9\n", "v125")
PROBE(probe_snop10, "s_nop 7\n", "s_nop 10\n", "v125")
PROBE(probe_snop11, "s_nop 7\n", "s_nop 11\n", "v125")
PROBE(probe_snop12, "s_nop 7\n", "s_nop 12\n", "v125")
PROBE(probe_snop13, "s_nop 7\n", "s_nop 13\n", "v125")
PROBE(probe_snop14, "s_nop 7\n", "s_nop 14\n", "v125")
PROBE(probe_snop15, "s_nop 7\n", "s_nop 15\n", "v125")
PROBE(probe_snop15_0, "s_nop 7\n", "s_nop 15\n s_nop 0\n", "v125")
PROBE(probe_snop15_1, "s_nop 7\n", "s_nop 15\n s_nop 1\n", "v125")
PROBE(probe_snop15_2, "s_nop 7\n", "s_nop 15\n s_nop 2\n", "v125")
PROBE(probe_snop15_3, "s_nop 7\n", "s_nop 15\n s_nop 3\n", "v125")
PROBE(probe_snop15_4, "s_nop 7\n", "s_nop 15\n s_nop 4\n", "v125")
PROBE(probe_snop15_6, "s_nop 7\n", "s_nop 15\n s_nop 6\n", "v125")
PROBE(probe_snop15_8, "s_nop 7\n", "s_nop 15\n s_nop 8\n", "v125")
PROBE(probe_ref, "s_nop 7\n", "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n", "v125")

typedef void (*kfn)(const unsigned*, float*, int);
int main() {
    std::vector<unsigned> h(512);
    for (int i = 0; i < 512; ++i) { unsigned short x = 0x3c00 + (i * 37 % 512), y = 0x3800 + (i * 91 % 700); h[i] = x | (y << 16); }
    unsigned* din; float* dout;
    hipMalloc(&din, 2048); hipMalloc(&dout, 256 * 4 * 256 * 2 * sizeof(float));
    hipMemcpy(din, h.data(), 2048, hipMemcpyHostToDevice);
    struct { const char* n; kfn f; } ks[] = {
        {"probe_snop8", probe_snop8},
        {"probe_snop9", probe_snop9},
        {"probe_snop10", probe_snop10},
        {"probe_snop11", probe_snop11},
        {"probe_snop12", probe_snop12},
        {"probe_snop13", probe_snop13},
        {"probe_snop14", probe_snop14},
        {"probe_snop15", probe_snop15},
        {"probe_snop15_0", probe_snop15_0},
        {"probe_snop15_1", probe_snop15_1},
        {"probe_snop15_2", probe_snop15_2},
        {"probe_snop15_3", probe_snop15_3},
        {"probe_snop15_4", probe_snop15_4},
        {"probe_snop15_6", probe_snop15_6},
        {"probe_snop15_8", probe_snop15_8},
    };
    for (int wps : {1, 4}) {
        const int grid = 256 * wps;
        std::vector<float> o(grid * 256 * 2), base(grid * 256);
        hipLaunchKernelGGL(probe_ref, dim3(grid), dim3(256), 0, 0, din, dout, 50);
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < grid * 256; ++i) base[i] = o[grid * 256 + i];
        for (auto& k : ks) {
            hipLaunchKernelGGL(k.f, dim3(grid), dim3(256), 0, 0, din, dout, 50);
            hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
            long wrong = 0, unstable = 0;
            for (int i = 0; i < grid * 256; ++i) { wrong += o[grid * 256 + i] != base[i]; unstable += o[i] != 0.f; }
            printf("%d waves/SIMD  %-16s wrong lanes %7ld of %d; unstable %ld\n", wps, k.n, wrong, grid * 256, unstable);
        }
    }
    return 0;
}
