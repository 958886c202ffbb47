/*
 * deepmimo_amd.h - C-ABI of the MI355X-native DeepMIMO channel-generation path.
 *
 * The reference (jmoraispk/DeepMIMO v4.0.0a3) is pure Python and has no FFI: the seam this
 * library sits behind is the method  Dataset.compute_channels(params)
 * (deepmimo/generator/dataset.py:224-268) and the lazy attributes it feeds (`channel`, `los`,
 * `num_paths`, `_fov_mask`, rotated angles, powers; dataset.py:831-869).  A maintainer binds
 * these entry points with ctypes (INTEGRATION.md shows the stub).  Everything is plain C:
 * borrowed DEVICE pointers + sizes in, caller-allocated DEVICE buffers out, a hipStream_t passed
 * as void*.  The library allocates nothing, keeps no global state except a thread-local error
 * string (in particular it reads no environment variable), launches asynchronously on the given stream and never
 * synchronises.  One entry point does I/O: dmx_mats_to_device (the loader) opens the files it is given and runs reader
 * threads for the duration of the call; nothing of either outlives it.
 *
 * Layouts: ray fields are float32 row-major [n_ue, ld] (ld >= n_paths), NaN = "no path", exactly
 * the arrays Dataset holds after core.py:209-219.  The channel tensor is complex64 interleaved
 * (re, im), C-contiguous [n_ue, M_rx, M_tx, K] (frequency domain) or [n_ue, M_rx, M_tx, P]
 * (time domain), last index fastest, as channel.py:257 allocates it.
 */
#ifndef DEEPMIMO_AMD_H
#define DEEPMIMO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMX_ABI_VERSION 3

/* status codes (0 = ok).  dmx_last_error() holds the message of the last failure on this thread. */
#define DMX_OK               0
#define DMX_ERR_ARG         -1   /* NULL / inconsistent argument */
#define DMX_ERR_SHAPE       -2   /* shape outside what the kernels support */
#define DMX_ERR_LAUNCH      -3   /* hipLaunch / runtime failure */
#define DMX_ERR_WORKSPACE   -4   /* workspace too small or misaligned */

/* radiation patterns: deepmimo/consts.py:254, deepmimo/generator/ant_patterns.py:21-71 */
#define DMX_PATTERN_ISOTROPIC        0
#define DMX_PATTERN_HALFWAVE_DIPOLE  1

/* Ray-path records of one (TX, RX-set) pair: the float32 matrices of core.py:209-219. */
typedef struct dmx_rays {
    int64_t n_ue;          /* users (rows) */
    int32_t n_paths;       /* loaded paths per user (columns) */
    int32_t ld;            /* row stride in elements, >= n_paths */
    const float* power;    /* dBW                        consts.py:188 */
    const float* phase;    /* degrees                    consts.py:189 */
    const float* delay;    /* seconds                    consts.py:190 */
    const float* aoa_az;   /* degrees                    consts.py:191 */
    const float* aoa_el;   /* degrees, zenith            consts.py:192 */
    const float* aod_az;   /* degrees                    consts.py:193 */
    const float* aod_el;   /* degrees, zenith            consts.py:194 */
    const float* inter;    /* interaction code, 0 = LoS  consts.py:197 */
    const float* doppler_vel;  /* m/s, optional (NULL)   deepmimo_v3/consts.py:63 */
    const float* doppler_acc;  /* m/s^2, optional (NULL) deepmimo_v3/consts.py:64 */
} dmx_rays;

/* Channel parameters: ChannelGenParameters (channel.py:33-63) after validate() (:78-139), plus
 * the FoV pair Dataset.apply_fov stored (dataset.py:423-448).  Angles in RADIANS as the host's
 * np.deg2rad produced them (float64), so the device sees the reference's exact doubles. */
typedef struct dmx_params {
    int32_t bs_shape[2];          /* [Mh, Mv]; element m = y + Mh*z   geometry.py:105-120 */
    int32_t ue_shape[2];
    double  bs_spacing;           /* wavelengths */
    double  ue_spacing;
    double  bs_rotation[3];       /* radians about x, y, z            geometry.py:286-291 */
    double  ue_rotation[3];       /* radians; used when ue_rotation_per_user == NULL */
    const double* ue_rotation_per_user; /* device [n_ue, 3] DEGREES (dataset.py:329-338) or NULL */
    int32_t bs_pattern;           /* DMX_PATTERN_* */
    int32_t ue_pattern;
    int32_t fov_enabled;          /* 0: Dataset._compute_fov returns mask None (dataset.py:484) */
    int32_t bs_fov_restricted;    /* dataset.py:497 */
    int32_t ue_fov_restricted;    /* dataset.py:502 */
    double  bs_fov[2];            /* radians [horizontal, vertical]   geometry.py:184 */
    double  ue_fov[2];
    int32_t num_paths;            /* params.num_paths; min(num_paths, rays.n_paths) paths are used */
    int32_t freq_domain;          /* 1: OFDM channel, 0: time-domain taps   channel.py:54 */
    int32_t n_subcarriers;        /* ofdm.subcarriers (N) */
    int32_t n_selected;           /* K = len(ofdm.selected_subcarriers) */
    const int32_t* selected_subcarriers; /* device [K] */
    double  bandwidth;            /* Hz; Ts = 1/bandwidth             channel.py:223 */
    int32_t rx_filter;            /* ofdm.rx_filter (LPF / sinc interpolation)  channel.py:193-194 */
    int32_t enable_doppler;       /* apply the v3 Doppler term (construct_deepmimo.py:267-280) */
    double  carrier_freq;         /* Hz, for Doppler */
    /* Host-side promise about the DEVICE array above (ABI 2): when sc_stride > 0 the caller guarantees
     * selected_subcarriers[k] == sc_first + k * sc_stride for every k (np.arange(N), np.arange(0, N, s), [0] ...:
     * what channel.py:57 and the reference's own scripts select).  It lets dmx_channels_fd pick the folded kernel for
     * few antenna pairs without reading the device array back.  sc_stride = 0 makes no promise (any selection). */
    int32_t sc_first;
    int32_t sc_stride;
    /* Arithmetic mode of the matrix-core kernels (ABI 3).  The reference multiplies and sums every path in complex128
     * (channel.py:283-284); the kernels form every product from three f16 x f16 terms with fp32 accumulation, which
     * keeps |error| <= ~2e-6 of a user's strongest path.  0 = that, for every path (default).
     * DMX_FLAG_ADAPTIVE_TERMS: opt-in - stage 1 writes a user's kept paths in order of falling amplitude and a last
     * 8-path group whose paths are all >= 66 dB below the strongest one is multiplied out in ONE term (worst case 7.6e-6
     * of the strongest path; 3-5 % less time at 25 paths).  The flag must be the same in dmx_path_prep and in the
     * stage-2 call that reads its workspace. */
    uint32_t flags;
    uint32_t reserved0;          /* must be 0 */
} dmx_params;

#define DMX_FLAG_ADAPTIVE_TERMS 1u

/* Optional side products of the path-prep stage (any pointer may be NULL = not wanted).
 * All device pointers; [n_ue, n_paths] arrays are dense row-major with row stride n_paths. */
typedef struct dmx_side {
    uint8_t*  fov_mask;              /* [n_ue, n_paths] 0/1                 dataset.py:494-504 */
    int32_t*  num_paths;             /* [n_ue]                              dataset.py:613-619 */
    int32_t*  los;                   /* [n_ue] in {-1, 0, 1}                dataset.py:569-611 */
    double*   aod_el_rot;            /* [n_ue, n_paths] radians, before FoV masking  dataset.py:341-349 */
    double*   aod_az_rot;
    double*   aoa_el_rot;
    double*   aoa_az_rot;
    float*    power_linear;          /* [n_ue, n_paths] W                   dataset.py:694-696 */
    double*   power_linear_ant_gain; /* [n_ue, n_paths] W                   dataset.py:665-691 */
    uint32_t* max_delay_key;         /* [1], caller zeroes it; order-preserving key of
                                        nanmax(delay[:, :P]) (channel.py:231), see dmx_decode_max_delay */
} dmx_side;

/* ABI version of the loaded library (== DMX_ABI_VERSION of the header it was built from). */
int dmx_version(void);

/* Message of the last failing call on the calling thread ("" if none). */
const char* dmx_last_error(void);

/* Bytes of device workspace dmx_path_prep needs for n_ue users (256-byte aligned base required). */
size_t dmx_workspace_bytes(const dmx_params* prm, int64_t n_ue, int32_t n_paths_loaded);

/* Decode *max_delay_key (copied back to the host) into seconds; NaN when no finite delay was seen. */
float dmx_decode_max_delay(uint32_t key);

/*
 * Stage 1 (replaces dataset.py:310-356 rotate, :461-512 FoV, :665-696 powers/patterns, :569-619
 * LoS/path counts, and the per-path part of channel.py:170-198): one pass over the ray matrices
 * that fills the compact per-path records in `workspace` and the requested side products.
 */
int dmx_path_prep(const dmx_rays* rays, const dmx_params* prm, void* workspace, size_t workspace_bytes,
                  const dmx_side* side, void* stream);

/*
 * Stage 2, frequency domain (replaces dataset.py:398-417 array-response product and the user loop
 * channel.py:264-284): out[u, rx, tx, k] = sum_l a_rx[rx,l] a_tx[tx,l] c_l exp(-j 2pi dn_l sc_k / N)
 * for users [user_begin, user_begin + user_count) of the prepared workspace; `out` points at the
 * first of those users (complex64 [user_count, M_rx, M_tx, K]).
 * variant: 0 = automatic; 1 = fp32 vector kernel; 2 = split-precision MFMA kernel (persistent workgroups of 8
 *          waves, two per CU, or of 4 waves up to 128 subcarriers; non-temporal output stores); 9 = small-output
 *          kernel (one wave per user; automatic when few subcarriers are selected); 12 = folded matrix-core kernel
 *          for at most 128 antenna pairs (needs prm->sc_stride > 0; automatic for every such selection up to 32 pairs and
 *          up to 128 pairs while few subcarriers are selected).
 *          Tuning knobs kept for A/B measurements (all parity-tested): 3 = MFMA with plain stores (16 waves),
 *          4 / 5 / 10 = 4- / 8- / 16-wave workgroups whatever the subcarrier count, 8 = one 16-wave workgroup per
 *          (user, row block) instead of persistent workgroups, 11 = exactly the resident number of persistent
 *          16-wave workgroups.
 */
int dmx_channels_fd(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                    int64_t user_begin, int64_t user_count, void* out_c64, int32_t variant, void* stream);

/* Host-only: the kernel `variant = 0` selects for this shape (1, 2, 9 or 12 above), from measured crossovers; negative on
 * a bad argument.  No GPU involved. */
int dmx_fd_kernel_choice(const dmx_params* prm, int32_t n_paths_loaded);

/*
 * Stage 2, frequency domain with the receive low-pass filter (ofdm.rx_filter = 1; replaces
 * channel.py:166-168, 193-194): g[l,k] = sum_d c_l sinc(d - dn_l) exp(-j 2pi d sc_k / N) is first
 * written to `lpf_workspace` (dmx_lpf_workspace_bytes(prm, user_count, n_paths_loaded) bytes,
 * 256-byte aligned, device), then contracted as in dmx_channels_fd.
 */
size_t dmx_lpf_workspace_bytes(const dmx_params* prm, int64_t user_count, int32_t n_paths_loaded);
int dmx_channels_fd_lpf(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                        int64_t user_begin, int64_t user_count, void* lpf_workspace, size_t lpf_workspace_bytes,
                        void* out_c64, void* stream);

/*
 * Stage 2 with a fused consumer: beam-space channel for a TX codebook F [n_beams, M_tx]
 * (complex64, row-major, device), out[u, rx, b, k] = sum_tx F[b,tx] H[u, rx, tx, k]  - what
 * docs/manual.ipynb cell 105 computes as `F1 @ dataset.channel` after materialising H.  H itself is never
 * written: the projection is folded into the transmit array response before the contraction, so the output
 * (complex64 [user_count, M_rx, n_beams, K]) and the HBM traffic shrink by M_tx / n_beams.
 */
size_t dmx_beam_workspace_bytes(const dmx_params* prm, int64_t user_count, int32_t n_paths_loaded, int32_t n_beams);
int dmx_channels_fd_beams(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                          int64_t user_begin, int64_t user_count, const void* codebook_c64, int32_t n_beams,
                          void* beam_workspace, size_t beam_workspace_bytes, void* out_c64, void* stream);

/*
 * Fused consumer that writes no [N, ., K] tensor at all: the beam-sweep reduction of
 * docs/manual.ipynb cell 105, `np.abs(F1 @ dataset.channel).mean(axis=1).mean(axis=-1)`:
 *   out_mean_amp[u, b] = 1 / (M_rx K) * sum_rx sum_k | sum_tx F[b,tx] H[u, rx, tx, k] |        float32 [user_count, n_beams]
 *   out_best_beam[u]   = argmax_b out_mean_amp[u, b] (first maximum; -1 for a user without paths)   int32 [user_count], may be NULL
 * (cells 110 / 112 take argmax / max of the rounded dBm values 20 log10(.) + 30: the host does that on the small
 * [N, n_beams] result).  Frequency domain without rx_filter; same workspace as dmx_channels_fd_beams.
 */
int dmx_beam_power(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                   int64_t user_begin, int64_t user_count, const void* codebook_c64, int32_t n_beams,
                   void* beam_workspace, size_t beam_workspace_bytes, float* out_mean_amp, int32_t* out_best_beam,
                   void* stream);

/*
 * Stage 2, time domain (replaces channel.py:285-287): out[u, rx, tx, s] = a_rx a_tx sqrt(p) e^{j phase}
 * of the s-th valid path (valid paths compacted to the front, remaining slots zero),
 * complex64 [user_count, M_rx, M_tx, P], P = min(num_paths, n_paths_loaded).
 */
int dmx_channels_td(const dmx_params* prm, const void* workspace, int64_t n_ue, int32_t n_paths_loaded,
                    int64_t user_begin, int64_t user_count, void* out_c64, void* stream);

/*
 * Consumer of the ray records: Dataset.compute_pathloss (dataset.py:541-566).  out: device float32 [n_ue], dB;
 * coherent != 0 sums complex gains, 0 sums amplitudes; NaN where the summed power is not positive.
 */
int dmx_pathloss(const dmx_rays* rays, int32_t coherent, float* out, void* stream);

/* ---- loader step before the path: reference .mat files -> device SoA ------------ */

/* Where one numeric array lives inside a MATLAB level-5 MAT-file image, as written by scipy.io.savemat
 * in the DeepMIMO converter (deepmimo/converter/converter_utils.py:59-85). */
typedef struct dmx_mat_info {
    int32_t class_id;      /* mxCLASS of the array (7 = single, 6 = double, 12 = int32, ...) */
    int32_t data_type;     /* miTYPE of the stored payload (7 = miSINGLE, 9 = miDOUBLE, 5 = miINT32, ...) */
    int32_t elem_bytes;
    int32_t ndim;
    int64_t dims[4];       /* MATLAB order; payload is column-major */
    int64_t data_offset;   /* byte offset of the payload in the file image */
    int64_t data_bytes;
    int32_t compressed;    /* 1: the element is zlib-compressed (miCOMPRESSED) at [comp_offset, +comp_bytes):   */
    int32_t reserved;      /*    inflate it and call dmx_mat5_find again on the inflated element              */
    int64_t comp_offset;
    int64_t comp_bytes;
} dmx_mat_info;

/* Host-side: locate variable `var_name` (NULL = first array) in a MAT-file image.  Replaces the parsing
 * half of scipy.io.loadmat at deepmimo/generator/core.py:241.  No GPU involved. */
int dmx_mat5_find(const void* file_image, size_t len, const char* var_name, dmx_mat_info* info);

/* Device: raw column-major [rows, cols] payload (copied to HBM as stored) -> row-major float32
 * [n_sel, cols_keep], gathering rows d_row_idx (device int64 [n_sel], NULL = the first n_sel rows) and
 * keeping the first cols_keep columns.  Replaces core.py:250 (rx_idxs select) and :254 (max_paths trim). */
int dmx_mat_to_rowmajor_f32(const void* d_payload, int32_t data_type, int64_t rows, int64_t cols,
                            const int64_t* d_row_idx, int64_t n_sel, int32_t cols_keep, float* d_out, void* stream);

/* All ray matrices of a TX/RX pair in one pipeline (core.py:241-254 loads them one `scipy.io.loadmat` at a time): reader
 * threads of the library `pread` the payloads - every file in slices by all threads, file after file - into `staging`
 * (host memory the caller page-locked, e.g. hipHostMalloc; job i's bytes at stage_offset), while the calling thread, as
 * each file completes, queues its asynchronous H2D copy into d_payload and dmx_mat_to_rowmajor_f32 on `stream`.  Returns
 * when everything is QUEUED: `staging`, every d_payload and d_row_idx must stay untouched until the stream has been
 * synchronised.  path == NULL: the job's bytes are in `staging` already (an inflated miCOMPRESSED element). */
typedef struct dmx_mat_job {
    const char* path;        /* file holding the payload, or NULL */
    uint64_t file_offset;    /* dmx_mat_info.data_offset */
    uint64_t nbytes;         /* dmx_mat_info.data_bytes */
    uint64_t stage_offset;   /* where the payload goes inside `staging` */
    void*    d_payload;      /* device, nbytes */
    int32_t  data_type;      /* dmx_mat_info.data_type */
    int32_t  cols_keep;
    int64_t  rows, cols;
    float*   d_out;          /* device float32 [n_sel, cols_keep] */
} dmx_mat_job;
int dmx_mats_to_device(const dmx_mat_job* jobs, int32_t n_jobs, void* staging, const int64_t* d_row_idx, int64_t n_sel,
                       int32_t n_threads, void* stream);

/* ---- two steps before the path: Wireless InSite paths.p2m text -> ray matrices ----- */

/* Receiver count announced on line 22 of a `*.paths.*.p2m` file image (p2m_parser.py:36, 80); -1 on error. */
int64_t dmx_p2m_count_rx(const char* text, size_t len);

/* Host-side parse of a `*.paths.*.p2m` file image into NaN-padded float32 matrices [n_rx, max_paths]
 * (inter_pos: [n_rx, max_paths, max_inter, 3], may be NULL).  Replaces paths_parser,
 * deepmimo/converter/wireless_insite/p2m_parser.py:48-145, before its compress_path_data step. */
int dmx_p2m_parse_paths(const char* text, size_t len, int32_t max_paths, int32_t max_inter, int64_t n_rx,
                        float* aoa_az, float* aoa_el, float* aod_az, float* aod_el, float* delay, float* power,
                        float* phase, float* inter, float* inter_pos);

#ifdef __cplusplus
}
#endif
#endif /* DEEPMIMO_AMD_H */
