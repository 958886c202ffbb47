// Store-pattern microbenchmark (GPU box only): what write rate does the stage-2 output ordering reach with
// NO compute?  One 1024-thread workgroup per 1 MiB "user block" [256 rows x 1024 floats]; 16 waves, each
// owns 2 strips of 32 columns and walks the 8 row tiles, exactly like k2_fd_mfma.
//   mode 0: dword per lane, one store = rows (r, r+4) x 128 B         (the accumulator layout as it is)
//   mode 1: dwordx4 per lane, one store = 8 rows x 128 B               (after an in-register 4x4 transpose)
//   mode 2: dwordx4 per lane, one store = 1 KiB contiguous of one row  (what an LDS-staged epilogue could do)
//   mode 3: dword per lane as mode 0 but row tiles outermost across strips (row-band order)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(1024) void pattern(float* __restrict__ out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* o = out + (size_t)blockIdx.x * 256 * 1024;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(o, 0, 256 * 1024 * 4, 0x00020000);
    if (MODE == 0 || MODE == 3) {
        const int col = lane & 31, hh = lane >> 5;
        if (MODE == 0) {
            for (int strip = wave; strip < 32; strip += 16) {
                const unsigned lane_off = ((unsigned)(4 * hh) * 1024u + (unsigned)(strip * 32 + col)) * 4u;
                for (int pt = 0; pt < 8; ++pt)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + i), rs, lane_off,
                                                              (unsigned)(pt * 32 + (i & 3) + 8 * (i >> 2)) * 4096u, 2);
            }
        } else {
            for (int pt = 0; pt < 8; ++pt)
                for (int strip = wave; strip < 32; strip += 16) {
                    const unsigned lane_off = ((unsigned)(4 * hh) * 1024u + (unsigned)(strip * 32 + col)) * 4u;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + i), rs, lane_off,
                                                              (unsigned)(pt * 32 + (i & 3) + 8 * (i >> 2)) * 4096u, 2);
                }
        }
    } else if (MODE == 1) {
        // lane = (row-in-8 = lane>>3, 16-B chunk = lane&7): one store covers 8 rows x 128 B
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 d = {__builtin_bit_cast(unsigned, v), 1u, 2u, 3u};
        for (int strip = wave; strip < 32; strip += 16) {
            const unsigned lane_off = ((unsigned)(lane >> 3) * 1024u + (unsigned)(strip * 32 + (lane & 7) * 4)) * 4u;
            for (int pt = 0; pt < 8; ++pt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane_off, (unsigned)(pt * 32 + g * 8) * 4096u, 2);
        }
    } else {
        // wave writes whole 1-KiB quarter rows: rows wave*16 .. wave*16+15, 4 stores per row
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const u4 d = {__builtin_bit_cast(unsigned, v), 1u, 2u, 3u};
        for (int r = wave * 16; r < wave * 16 + 16; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(d, rs, (unsigned)(q * 256 + lane * 4) * 4u, (unsigned)r * 4096u, 2);
    }
}

// mode 4: like mode 2 (1 KiB per store instruction, 1 MiB block per workgroup) with plain stores
// mode 5: like torch fill_: 256-thread workgroups, consecutive workgroups write consecutive 16 KiB chunks
// mode 6: mode 0 with plain stores
// mode 7: mode 5 with nt stores
template <int AUX>
__global__ __launch_bounds__(1024) void pattern_rows(float* __restrict__ out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* o = out + (size_t)blockIdx.x * 256 * 1024;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(o, 0, 256 * 1024 * 4, 0x00020000);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 d = {__builtin_bit_cast(unsigned, v), 1u, 2u, 3u};
    for (int r = wave * 16; r < wave * 16 + 16; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(d, rs, (unsigned)(q * 256 + lane * 4) * 4u, (unsigned)r * 4096u, AUX);
}

template <bool NT>
__global__ __launch_bounds__(256) void pattern_linear(float4* __restrict__ out4, float v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* o = reinterpret_cast<f4*>(out4) + (size_t)blockIdx.x * 1024 + threadIdx.x;       // 16 KiB per workgroup
    const f4 d = {v, 1.f, 2.f, 3.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (NT) __builtin_nontemporal_store(d, o + i * 256); else o[i * 256] = d;
    }
}

template <int AUX>
__global__ __launch_bounds__(1024) void pattern_tiles(float* __restrict__ out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* o = out + (size_t)blockIdx.x * 256 * 1024;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(o, 0, 256 * 1024 * 4, 0x00020000);
    const int col = lane & 31, hh = lane >> 5;
    for (int strip = wave; strip < 32; strip += 16) {
        const unsigned lane_off = ((unsigned)(4 * hh) * 1024u + (unsigned)(strip * 32 + col)) * 4u;
        for (int pt = 0; pt < 8; ++pt)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + i), rs, lane_off,
                                                      (unsigned)(pt * 32 + (i & 3) + 8 * (i >> 2)) * 4096u, AUX);
    }
}

// mode 8: dwordx2 per lane, one store = 4 rows x 128 B (what a lane-pair exchange of the accumulator would give)
__global__ __launch_bounds__(1024) void pattern_pairs(float* __restrict__ out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* o = out + (size_t)blockIdx.x * 256 * 1024;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(o, 0, 256 * 1024 * 4, 0x00020000);
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 d = {__builtin_bit_cast(unsigned, v), 7u};
    const int hh = lane >> 5, odd = lane & 1, c2 = (lane & 31) >> 1;      // 16 column pairs
    for (int strip = wave; strip < 32; strip += 16) {
        const unsigned lane_off = ((unsigned)(4 * hh + odd) * 1024u + (unsigned)(strip * 32 + c2 * 2)) * 4u;
        for (int pt = 0; pt < 8; ++pt)
#pragma unroll
            for (int i = 0; i < 8; ++i)                                     // row pairs (0,1),(2,3),(8,9),...
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane_off,
                                                      (unsigned)(pt * 32 + ((2 * i) & 3) + 8 * ((2 * i) >> 2)) * 4096u, 2);
    }
}

// mode 9: dword per lane, one store = ONE row x 256 B (two adjacent strips merged with v_permlane32_swap)
__global__ __launch_bounds__(1024) void pattern_row256(float* __restrict__ out, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* o = out + (size_t)blockIdx.x * 256 * 1024;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(o, 0, 256 * 1024 * 4, 0x00020000);
    const unsigned lane_off = (unsigned)(wave * 64 + lane) * 4u;            // 16 waves x 64 columns = the 1024-float row
    for (int pt = 0; pt < 8; ++pt)
#pragma unroll
        for (int i = 0; i < 32; ++i)                                         // 32 rows of the tile, one per store
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + i), rs, lane_off, (unsigned)(pt * 32 + i) * 4096u, 2);
}

int main() {
    const int users = 100000;
    float* out;
    if (hipMalloc(&out, (size_t)users * 256 * 1024 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const double gb = (double)users * 256 * 1024 * 4 / 1e9;
    for (int mode = 0; mode < 10; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            if (mode == 0) pattern<0><<<users, 1024>>>(out, 1.f);
            if (mode == 1) pattern<1><<<users, 1024>>>(out, 1.f);
            if (mode == 2) pattern<2><<<users, 1024>>>(out, 1.f);
            if (mode == 3) pattern<3><<<users, 1024>>>(out, 1.f);
            if (mode == 4) pattern_rows<0><<<users, 1024>>>(out, 1.f);
            if (mode == 5) pattern_linear<false><<<users * 64, 256>>>((float4*)out, 1.f);
            if (mode == 6) pattern_tiles<0><<<users, 1024>>>(out, 1.f);
            if (mode == 7) pattern_linear<true><<<users * 64, 256>>>((float4*)out, 1.f);
            if (mode == 8) pattern_pairs<<<users, 1024>>>(out, 1.f);
            if (mode == 9) pattern_row256<<<users, 1024>>>(out, 1.f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("mode %d: %.2f ms  %.0f GB/s\n", mode, best, gb / best * 1e3);
    }
    return 0;
}
