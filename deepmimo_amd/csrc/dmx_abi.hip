// C-ABI entry points (include/deepmimo_amd.h): argument validation, workspace carving, error
// string.  No allocation, no synchronisation, no global state besides the thread-local message.
#include "dmx_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace dmx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int device_cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) {
        (void)hipGetLastError();
        cus = 256;                                        // MI355X
    }
    return cus;
}

static int check_params(const dmx_params* p) {
    if (!p) { set_error("params is NULL"); return DMX_ERR_ARG; }
    for (int i = 0; i < 2; ++i) {
        if (p->bs_shape[i] < 1 || p->ue_shape[i] < 1) { set_error("antenna shape entries must be >= 1"); return DMX_ERR_SHAPE; }
    }
    if ((int64_t)p->bs_shape[0] * p->bs_shape[1] > 65536 || (int64_t)p->ue_shape[0] * p->ue_shape[1] > 65536) {
        set_error("antenna panel larger than 65536 elements"); return DMX_ERR_SHAPE;
    }
    if (p->bs_pattern < 0 || p->bs_pattern > 1 || p->ue_pattern < 0 || p->ue_pattern > 1) {
        set_error("unknown radiation pattern id"); return DMX_ERR_ARG;
    }
    if (p->num_paths < 0) { set_error("num_paths must be >= 0"); return DMX_ERR_ARG; }
    if ((p->flags & ~DMX_FLAG_ADAPTIVE_TERMS) != 0 || p->reserved0 != 0) { set_error("unknown bits in dmx_params.flags / reserved0"); return DMX_ERR_ARG; }
    if (p->freq_domain) {
        if (p->n_subcarriers < 1) { set_error("ofdm.subcarriers must be >= 1"); return DMX_ERR_ARG; }
        if (p->n_selected < 0 || (p->n_selected > 0 && !p->selected_subcarriers)) {
            set_error("selected_subcarriers missing"); return DMX_ERR_ARG;
        }
        if (!(p->bandwidth > 0)) { set_error("ofdm.bandwidth must be > 0"); return DMX_ERR_ARG; }
        if (p->sc_stride < 0) { set_error("sc_stride must be >= 0"); return DMX_ERR_ARG; }
    }
    return DMX_OK;
}

static inline int used_paths(const dmx_params* p, int32_t loaded) { return p->num_paths < loaded ? p->num_paths : loaded; }

}  // namespace dmx

using namespace dmx;

extern "C" {

int dmx_version(void) { return DMX_ABI_VERSION; }

const char* dmx_last_error(void) { return g_err; }

size_t dmx_workspace_bytes(const dmx_params* prm, int64_t n_ue, int32_t n_paths_loaded) {
    if (!prm || n_ue < 0 || n_paths_loaded < 0) return 0;
    return ws_carve(nullptr, n_ue, used_paths(prm, n_paths_loaded), nullptr);
}

float dmx_decode_max_delay(uint32_t key) {
    if (key == 0) { uint32_t q = 0x7fc00000u; float f; memcpy(&f, &q, 4); return f; }
    uint32_t b = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

int dmx_path_prep(const dmx_rays* rays, const dmx_params* prm, void* workspace, size_t workspace_bytes,
                  const dmx_side* side, void* stream) {
    int rc = check_params(prm);
    if (rc) return rc;
    if (!rays) { set_error("rays is NULL"); return DMX_ERR_ARG; }
    if (rays->n_ue < 0 || rays->n_paths < 0 || rays->ld < rays->n_paths) { set_error("bad ray matrix shape"); return DMX_ERR_ARG; }
    if (rays->n_ue > 0 && rays->n_paths > 0 &&
        (!rays->power || !rays->phase || !rays->delay || !rays->aoa_az || !rays->aoa_el || !rays->aod_az ||
         !rays->aod_el || !rays->inter)) {
        set_error("a required ray field pointer is NULL"); return DMX_ERR_ARG;
    }
    if (rays->n_ue > 0x7fffffffLL * 4) { set_error("too many users for one call"); return DMX_ERR_SHAPE; }
    const int P = used_paths(prm, rays->n_paths);
    const size_t need = ws_carve(nullptr, rays->n_ue, P, nullptr);
    if (need > 0 && (!workspace || workspace_bytes < need)) { set_error("workspace too small: need %zu bytes", need); return DMX_ERR_WORKSPACE; }
    if (((uintptr_t)workspace & 255u) != 0) { set_error("workspace must be 256-byte aligned"); return DMX_ERR_WORKSPACE; }
    WsView ws;
    ws_carve(workspace, rays->n_ue, P, &ws);
    dmx_side s;
    if (side) s = *side; else memset(&s, 0, sizeof(s));
    return launch_path_prep(*rays, *prm, ws, s, (hipStream_t)stream);
}

static int stage2_common(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                         int64_t user_begin, int64_t user_count, void* out, WsView* ws) {
    int rc = check_params(prm);
    if (rc) return rc;
    if (n_ue < 0 || user_begin < 0 || user_count < 0 || user_begin + user_count > n_ue) {
        set_error("user range [%lld, %lld) outside [0, %lld)", (long long)user_begin, (long long)(user_begin + user_count), (long long)n_ue);
        return DMX_ERR_ARG;
    }
    if (user_count > 0x7fffffffLL) { set_error("too many users for one call"); return DMX_ERR_SHAPE; }
    if (user_count > 0 && (!workspace || !out)) { set_error("workspace/out is NULL"); return DMX_ERR_ARG; }
    if (((uintptr_t)workspace & 255u) != 0) { set_error("workspace must be 256-byte aligned"); return DMX_ERR_WORKSPACE; }
    if (((uintptr_t)out & 7u) != 0) { set_error("out must be 8-byte aligned"); return DMX_ERR_ARG; }
    ws_carve(const_cast<void*>(workspace), n_ue, used_paths(prm, n_paths_loaded), ws);
    return DMX_OK;
}

int dmx_channels_fd(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                    int64_t user_begin, int64_t user_count, void* out_c64, int32_t variant, void* stream) {
    WsView ws;
    int rc = stage2_common(prm, workspace, n_ue, n_paths_loaded, user_begin, user_count, out_c64, &ws);
    if (rc) return rc;
    if (!prm->freq_domain) { set_error("dmx_channels_fd called with freq_domain = 0"); return DMX_ERR_ARG; }
    if (prm->rx_filter) { set_error("rx_filter = 1 is handled by dmx_channels_fd_lpf"); return DMX_ERR_ARG; }
    if (variant < 0 || variant > 12) { set_error("unknown variant %d", variant); return DMX_ERR_ARG; }
    if (prm->n_selected == 0) return DMX_OK;
    return launch_channels_fd(*prm, ws, user_begin, user_count, (float2*)out_c64, variant, (hipStream_t)stream);
}

int dmx_fd_kernel_choice(const dmx_params* prm, int32_t n_paths_loaded) {
    if (!prm) { set_error("params is NULL"); return DMX_ERR_ARG; }
    if (n_paths_loaded < 0 || prm->n_selected < 0 || prm->bs_shape[0] < 1 || prm->bs_shape[1] < 1 || prm->ue_shape[0] < 1 ||
        prm->ue_shape[1] < 1) { set_error("bad shape"); return DMX_ERR_SHAPE; }
    WsView ws{};
    ws.P = used_paths(prm, n_paths_loaded);
    return fd_auto_choice(*prm, ws);
}

size_t dmx_lpf_workspace_bytes(const dmx_params* prm, int64_t user_count, int32_t n_paths_loaded) {
    if (!prm || user_count < 0 || n_paths_loaded < 0 || prm->n_selected < 0) return 0;
    return align_up((size_t)user_count * (size_t)used_paths(prm, n_paths_loaded) * (size_t)prm->n_selected * 8, 256);
}

int dmx_channels_fd_lpf(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                        int64_t user_begin, int64_t user_count, void* lpf_workspace, size_t lpf_workspace_bytes,
                        void* out_c64, void* stream) {
    WsView ws;
    int rc = stage2_common(prm, workspace, n_ue, n_paths_loaded, user_begin, user_count, out_c64, &ws);
    if (rc) return rc;
    if (!prm->freq_domain || !prm->rx_filter) { set_error("dmx_channels_fd_lpf needs freq_domain = 1 and rx_filter = 1"); return DMX_ERR_ARG; }
    if (prm->n_selected == 0) return DMX_OK;
    const size_t need = dmx_lpf_workspace_bytes(prm, user_count, n_paths_loaded);
    if (need > 0 && (!lpf_workspace || lpf_workspace_bytes < need || ((uintptr_t)lpf_workspace & 255u))) {
        set_error("lpf workspace too small or misaligned: need %zu bytes, 256-byte aligned", need);
        return DMX_ERR_WORKSPACE;
    }
    return launch_channels_fd_lpf(*prm, ws, user_begin, user_count, (float2*)lpf_workspace, (float2*)out_c64, (hipStream_t)stream);
}

size_t dmx_beam_workspace_bytes(const dmx_params* prm, int64_t user_count, int32_t n_paths_loaded, int32_t n_beams) {
    if (!prm || user_count < 0 || n_paths_loaded < 0 || n_beams < 0) return 0;
    return beam_workspace_bytes(user_count, n_beams, used_paths(prm, n_paths_loaded));
}

int dmx_channels_fd_beams(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                          int64_t user_begin, int64_t user_count, const void* codebook_c64, int32_t n_beams,
                          void* beam_workspace, size_t beam_workspace_bytes_, void* out_c64, void* stream) {
    WsView ws;
    int rc = stage2_common(prm, workspace, n_ue, n_paths_loaded, user_begin, user_count, out_c64, &ws);
    if (rc) return rc;
    if (!prm->freq_domain || prm->rx_filter) { set_error("dmx_channels_fd_beams needs freq_domain = 1 and rx_filter = 0"); return DMX_ERR_ARG; }
    if (n_beams < 0 || (n_beams > 0 && !codebook_c64)) { set_error("codebook missing"); return DMX_ERR_ARG; }
    if (prm->n_selected == 0 || n_beams == 0) return DMX_OK;
    if (ws.P > 32) { set_error("num_paths = %d exceeds the 32 paths the beam-space kernel supports", ws.P); return DMX_ERR_SHAPE; }
    const size_t need = beam_workspace_bytes(user_count, n_beams, ws.P);
    if (!beam_workspace || beam_workspace_bytes_ < need || ((uintptr_t)beam_workspace & 255u)) {
        set_error("beam workspace too small or misaligned: need %zu bytes, 256-byte aligned", need);
        return DMX_ERR_WORKSPACE;
    }
    return launch_channels_fd_beams(*prm, ws, user_begin, user_count, (const float2*)codebook_c64, n_beams, beam_workspace,
                                    (float2*)out_c64, (hipStream_t)stream);
}

int dmx_beam_power(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                   int64_t user_begin, int64_t user_count, const void* codebook_c64, int32_t n_beams,
                   void* beam_workspace, size_t beam_workspace_bytes_, float* out_mean_amp, int32_t* out_best_beam,
                   void* stream) {
    WsView ws;
    int rc = stage2_common(prm, workspace, n_ue, n_paths_loaded, user_begin, user_count, out_mean_amp, &ws);
    if (rc) return rc;
    if (!prm->freq_domain || prm->rx_filter) { set_error("dmx_beam_power needs freq_domain = 1 and rx_filter = 0"); return DMX_ERR_ARG; }
    if (n_beams < 1 || !codebook_c64) { set_error("codebook missing"); return DMX_ERR_ARG; }
    if (prm->n_selected < 1) { set_error("dmx_beam_power needs at least one selected subcarrier"); return DMX_ERR_ARG; }
    if (ws.P > 32) { set_error("num_paths = %d exceeds the 32 paths the beam-space kernels support", ws.P); return DMX_ERR_SHAPE; }
    const size_t need = beam_workspace_bytes(user_count, n_beams, ws.P);
    if (!beam_workspace || beam_workspace_bytes_ < need || ((uintptr_t)beam_workspace & 255u)) {
        set_error("beam workspace too small or misaligned: need %zu bytes, 256-byte aligned", need);
        return DMX_ERR_WORKSPACE;
    }
    return launch_beam_power(*prm, ws, user_begin, user_count, (const float2*)codebook_c64, n_beams, beam_workspace,
                             out_mean_amp, out_best_beam, (hipStream_t)stream);
}

int dmx_channels_td(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                    int64_t user_begin, int64_t user_count, void* out_c64, void* stream) {
    WsView ws;
    int rc = stage2_common(prm, workspace, n_ue, n_paths_loaded, user_begin, user_count, out_c64, &ws);
    if (rc) return rc;
    if (prm->freq_domain) { set_error("dmx_channels_td called with freq_domain = 1"); return DMX_ERR_ARG; }
    return launch_channels_td(*prm, ws, user_begin, user_count, (float2*)out_c64, (hipStream_t)stream);
}

}  // extern "C"
