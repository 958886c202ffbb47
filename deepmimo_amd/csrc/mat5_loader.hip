// Loader step before the path: ray matrices from the reference's on-disk format straight into the
// device SoA the kernels read (replaces scipy.io.loadmat + NumPy slicing in
// deepmimo/generator/core.py:186-258).
//
// The DeepMIMO converter writes each matrix with scipy.io.savemat (converter_utils.py:59-85): a MATLAB
// level-5 MAT file holding ONE numeric array, stored column-major (MATLAB order).  dmx_mat5_find is a
// small host-side parser (mat5_parser.cpp, plain C++ so it can be built under AddressSanitizer) that locates
// that array's payload in a file image (mmap'ed or read by the caller); dmx_mat_to_rowmajor_f32 is the device kernel that turns the raw column-major payload -
// copied to HBM as is - into the row-major float32 [n_sel, cols_keep] matrix the kernels want, fusing
// the receiver-index gather (core.py:250) and the max_paths trim (core.py:254) into the same pass.
// HBM-bound byte shuffling: 32x32 tiles through LDS so that both the reads (down a column of the file
// layout) and the writes (along a row of the output) are coalesced.
#include "dmx_common.h"
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <atomic>
#include <thread>
#include <vector>

namespace dmx {

// MAT-file data types (MAT-File Format, table 1-1)
enum { miINT8 = 1, miUINT8 = 2, miINT16 = 3, miUINT16 = 4, miINT32 = 5, miUINT32 = 6, miSINGLE = 7, miDOUBLE = 9,
       miINT64 = 12, miUINT64 = 13, miMATRIX = 14, miCOMPRESSED = 15 };

static int elem_size(uint32_t type) {
    switch (type) {
        case miINT8: case miUINT8: return 1;
        case miINT16: case miUINT16: return 2;
        case miINT32: case miUINT32: case miSINGLE: return 4;
        case miDOUBLE: case miINT64: case miUINT64: return 8;
        default: return 0;
    }
}

template <typename T>
__device__ __forceinline__ float to_f32(const void* p, size_t i) { return (float)reinterpret_cast<const T*>(p)[i]; }

__device__ __forceinline__ float load_as_f32(const void* p, int type, size_t i) {
    switch (type) {
        case miSINGLE: return reinterpret_cast<const float*>(p)[i];
        case miDOUBLE: return to_f32<double>(p, i);
        case miINT32: return to_f32<int32_t>(p, i);
        case miUINT32: return to_f32<uint32_t>(p, i);
        case miINT16: return to_f32<int16_t>(p, i);
        case miUINT16: return to_f32<uint16_t>(p, i);
        case miINT8: return to_f32<int8_t>(p, i);
        case miUINT8: return to_f32<uint8_t>(p, i);
        case miINT64: return to_f32<long long>(p, i);
        default: return to_f32<unsigned long long>(p, i);
    }
}

// in: column-major [rows, cols] (element (r, c) at c*rows + r).  out: row-major [n_sel, keep].
__global__ __launch_bounds__(256) void k_mat_to_rowmajor(const void* __restrict__ in, int type, int64_t rows, int64_t cols,
                                                         const int64_t* __restrict__ row_idx, int64_t n_sel, int keep,
                                                         float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8 threads
    const int64_t r0 = (int64_t)blockIdx.x * 32;                          // output rows of this tile
    const int c0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {                                    // read: tx walks the file's fast axis (rows)
        const int64_t orow = r0 + tx;
        const int c = c0 + j;
        float v = 0.f;
        if (orow < n_sel && c < keep) {
            const int64_t src = row_idx ? row_idx[orow] : orow;
            v = load_as_f32(in, type, (size_t)c * (size_t)rows + (size_t)src);
        }
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {                                    // write: tx walks the output's fast axis (cols)
        const int64_t orow = r0 + j;
        const int c = c0 + tx;
        if (orow < n_sel && c < keep) out[(size_t)orow * keep + c] = tile[tx][j];
    }
}

}  // namespace dmx

using namespace dmx;

extern "C" {

int dmx_mat_to_rowmajor_f32(const void* d_payload, int32_t data_type, int64_t rows, int64_t cols, const int64_t* d_row_idx,
                            int64_t n_sel, int32_t cols_keep, float* d_out, void* stream) {
    if (elem_size((uint32_t)data_type) == 0) { set_error("unsupported MAT data type %d", data_type); return DMX_ERR_ARG; }
    if (rows < 0 || cols < 0 || n_sel < 0 || cols_keep < 0 || cols_keep > cols) { set_error("bad matrix shape"); return DMX_ERR_ARG; }
    if (!d_row_idx && n_sel > rows) { set_error("n_sel exceeds the stored rows"); return DMX_ERR_ARG; }
    if (n_sel == 0 || cols_keep == 0) return DMX_OK;
    if (!d_payload || !d_out) { set_error("payload/out is NULL"); return DMX_ERR_ARG; }
    const int64_t gx = (n_sel + 31) / 32;
    if (gx > 0x7fffffffLL) { set_error("too many rows for one call"); return DMX_ERR_SHAPE; }
    hipLaunchKernelGGL(k_mat_to_rowmajor, dim3((unsigned)gx, (unsigned)((cols_keep + 31) / 32)), dim3(256), 0,
                       (hipStream_t)stream, d_payload, (int)data_type, rows, cols, d_row_idx, n_sel, (int)cols_keep, d_out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("k_mat_to_rowmajor launch failed: %s", hipGetErrorString(e)); return DMX_ERR_LAUNCH; }
    return DMX_OK;
}

// include/deepmimo_amd.h: dmx_mats_to_device.  Tasks are (job, slice) in job-major order behind one atomic counter, so
// the jobs complete in order, the first after 1 / n_jobs of the whole read time; the calling thread issues the copy and
// the layout kernel of a job the moment its last slice has landed.
int dmx_mats_to_device(const dmx_mat_job* jobs, int32_t n_jobs, void* staging, const int64_t* d_row_idx, int64_t n_sel,
                       int32_t n_threads, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && (!jobs || !staging))) { set_error("jobs / staging is NULL"); return DMX_ERR_ARG; }
    if (n_jobs == 0) return DMX_OK;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 32) n_threads = 32;
    struct Task { int job; uint64_t lo, hi; };
    std::vector<Task> tasks;
    std::vector<int> fds((size_t)n_jobs, -1);
    std::vector<std::atomic<int>> remaining((size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        int cnt = 0;
        if (jobs[j].path && jobs[j].nbytes) {
            fds[j] = open(jobs[j].path, O_RDONLY);
            if (fds[j] < 0) {
                for (int k = 0; k < j; ++k) if (fds[k] >= 0) close(fds[k]);
                set_error("cannot open %s", jobs[j].path);
                return DMX_ERR_ARG;
            }
            uint64_t step = (jobs[j].nbytes + (uint64_t)n_threads - 1) / (uint64_t)n_threads;
            step = (step + 4095) & ~(uint64_t)4095;
            if (step < 65536) step = 65536;
            for (uint64_t lo = 0; lo < jobs[j].nbytes; lo += step, ++cnt)
                tasks.push_back({j, lo, lo + step < jobs[j].nbytes ? lo + step : jobs[j].nbytes});
        }
        remaining[j].store(cnt);
    }
    std::atomic<size_t> next(0);
    std::atomic<int> failed(0);
    auto worker = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= tasks.size()) return;
            const Task& tk = tasks[t];
            const dmx_mat_job& jb = jobs[tk.job];
            char* dst = static_cast<char*>(staging) + jb.stage_offset;
            uint64_t got = tk.lo;
            while (got < tk.hi && !failed.load(std::memory_order_relaxed)) {
                const ssize_t n = pread(fds[tk.job], dst + got, (size_t)(tk.hi - got), (off_t)(jb.file_offset + got));
                if (n <= 0) { failed.store(1); break; }
                got += (uint64_t)n;
            }
            remaining[tk.job].fetch_sub(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    const int nthr = (int)tasks.size() < n_threads ? (int)tasks.size() : n_threads;
    for (int i = 0; i < nthr; ++i) pool.emplace_back(worker);
    int rc = DMX_OK;
    for (int j = 0; j < n_jobs && rc == DMX_OK; ++j) {
        while (remaining[j].load(std::memory_order_acquire) > 0) std::this_thread::yield();
        if (failed.load()) { set_error("short read of %s", jobs[j].path ? jobs[j].path : "(staged)"); rc = DMX_ERR_ARG; break; }
        const dmx_mat_job& jb = jobs[j];
        if (jb.nbytes) {
            hipError_t e = hipMemcpyAsync(jb.d_payload, static_cast<char*>(staging) + jb.stage_offset, (size_t)jb.nbytes,
                                          hipMemcpyHostToDevice, (hipStream_t)stream);
            if (e != hipSuccess) { set_error("hipMemcpyAsync failed: %s", hipGetErrorString(e)); rc = DMX_ERR_LAUNCH; break; }
        }
        rc = dmx_mat_to_rowmajor_f32(jb.d_payload, jb.data_type, jb.rows, jb.cols, d_row_idx, d_row_idx ? n_sel : jb.rows,
                                     jb.cols_keep, jb.d_out, stream);
    }
    if (rc != DMX_OK) failed.store(1);
    for (auto& t : pool) t.join();
    for (int fd : fds) if (fd >= 0) close(fd);
    return rc;
}

}  // extern "C"
