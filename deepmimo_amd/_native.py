"""ctypes binding of the C-ABI in include/deepmimo_amd.h (libdeepmimo_amd.so, built in-tree by
__graft_entry__.build() / deepmimo_amd/csrc/Makefile).

There is NO fallback: if the shared library is missing, or a call returns an error, this module
raises.  Structures mirror the header field by field."""
from __future__ import annotations

import ctypes as C
import os

# DMX_LIB_PATH lets a measurement load another build of the same ABI (A/B of two libraries); default = in-tree
LIB_PATH = os.environ.get("DMX_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                          "libdeepmimo_amd.so")
ABI_VERSION = 3
FLAG_ADAPTIVE_TERMS = 1

EXPORTED_SYMBOLS = ("dmx_version", "dmx_last_error", "dmx_workspace_bytes", "dmx_decode_max_delay",
                    "dmx_path_prep", "dmx_channels_fd", "dmx_channels_td", "dmx_channels_fd_lpf",
                    "dmx_lpf_workspace_bytes", "dmx_mat5_find", "dmx_mat_to_rowmajor_f32", "dmx_mats_to_device",
                    "dmx_beam_workspace_bytes", "dmx_channels_fd_beams", "dmx_pathloss",
                    "dmx_p2m_count_rx", "dmx_p2m_parse_paths", "dmx_fd_kernel_choice", "dmx_beam_power")

PATTERN_IDS = {"isotropic": 0, "halfwave-dipole": 1}

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


class DmxRays(C.Structure):
    _fields_ = [("n_ue", C.c_int64), ("n_paths", C.c_int32), ("ld", C.c_int32),
                ("power", C.c_void_p), ("phase", C.c_void_p), ("delay", C.c_void_p),
                ("aoa_az", C.c_void_p), ("aoa_el", C.c_void_p), ("aod_az", C.c_void_p), ("aod_el", C.c_void_p),
                ("inter", C.c_void_p), ("doppler_vel", C.c_void_p), ("doppler_acc", C.c_void_p)]


class DmxParams(C.Structure):
    _fields_ = [("bs_shape", C.c_int32 * 2), ("ue_shape", C.c_int32 * 2),
                ("bs_spacing", C.c_double), ("ue_spacing", C.c_double),
                ("bs_rotation", C.c_double * 3), ("ue_rotation", C.c_double * 3),
                ("ue_rotation_per_user", C.c_void_p),
                ("bs_pattern", C.c_int32), ("ue_pattern", C.c_int32),
                ("fov_enabled", C.c_int32), ("bs_fov_restricted", C.c_int32), ("ue_fov_restricted", C.c_int32),
                ("bs_fov", C.c_double * 2), ("ue_fov", C.c_double * 2),
                ("num_paths", C.c_int32), ("freq_domain", C.c_int32),
                ("n_subcarriers", C.c_int32), ("n_selected", C.c_int32),
                ("selected_subcarriers", C.c_void_p),
                ("bandwidth", C.c_double), ("rx_filter", C.c_int32), ("enable_doppler", C.c_int32),
                ("carrier_freq", C.c_double), ("sc_first", C.c_int32), ("sc_stride", C.c_int32),
                ("flags", C.c_uint32), ("reserved0", C.c_uint32)]


class DmxSide(C.Structure):
    _fields_ = [("fov_mask", C.c_void_p), ("num_paths", C.c_void_p), ("los", C.c_void_p),
                ("aod_el_rot", C.c_void_p), ("aod_az_rot", C.c_void_p),
                ("aoa_el_rot", C.c_void_p), ("aoa_az_rot", C.c_void_p),
                ("power_linear", C.c_void_p), ("power_linear_ant_gain", C.c_void_p),
                ("max_delay_key", C.c_void_p)]


class DmxMatInfo(C.Structure):
    _fields_ = [("class_id", C.c_int32), ("data_type", C.c_int32), ("elem_bytes", C.c_int32), ("ndim", C.c_int32),
                ("dims", C.c_int64 * 4), ("data_offset", C.c_int64), ("data_bytes", C.c_int64),
                ("compressed", C.c_int32), ("reserved", C.c_int32), ("comp_offset", C.c_int64), ("comp_bytes", C.c_int64)]


class DmxMatJob(C.Structure):
    _fields_ = [("path", C.c_char_p), ("file_offset", C.c_uint64), ("nbytes", C.c_uint64), ("stage_offset", C.c_uint64),
                ("d_payload", C.c_void_p), ("data_type", C.c_int32), ("cols_keep", C.c_int32),
                ("rows", C.c_int64), ("cols", C.c_int64), ("d_out", C.c_void_p)]


class NativeError(RuntimeError):
    """A dmx_* call returned a non-zero status."""


_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"deepmimo_amd: native library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C deepmimo_amd/csrc`. "
            "There is no CPU fallback for the channel-generation path.")
    lib = C.CDLL(LIB_PATH)
    lib.dmx_version.restype = C.c_int
    lib.dmx_last_error.restype = C.c_char_p
    lib.dmx_workspace_bytes.restype = C.c_size_t
    lib.dmx_workspace_bytes.argtypes = [C.POINTER(DmxParams), C.c_int64, C.c_int32]
    lib.dmx_fd_kernel_choice.restype = C.c_int
    lib.dmx_fd_kernel_choice.argtypes = [C.POINTER(DmxParams), C.c_int32]
    lib.dmx_decode_max_delay.restype = C.c_float
    lib.dmx_decode_max_delay.argtypes = [C.c_uint32]
    lib.dmx_path_prep.restype = C.c_int
    lib.dmx_path_prep.argtypes = [C.POINTER(DmxRays), C.POINTER(DmxParams), C.c_void_p, C.c_size_t,
                                  C.POINTER(DmxSide), C.c_void_p]
    lib.dmx_channels_fd.restype = C.c_int
    lib.dmx_channels_fd.argtypes = [C.POINTER(DmxParams), C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_int32, C.c_void_p]
    lib.dmx_channels_td.restype = C.c_int
    lib.dmx_channels_td.argtypes = [C.POINTER(DmxParams), C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p]
    lib.dmx_lpf_workspace_bytes.restype = C.c_size_t
    lib.dmx_lpf_workspace_bytes.argtypes = [C.POINTER(DmxParams), C.c_int64, C.c_int32]
    lib.dmx_channels_fd_lpf.restype = C.c_int
    lib.dmx_channels_fd_lpf.argtypes = [C.POINTER(DmxParams), C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.dmx_beam_workspace_bytes.restype = C.c_size_t
    lib.dmx_beam_workspace_bytes.argtypes = [C.POINTER(DmxParams), C.c_int64, C.c_int32, C.c_int32]
    lib.dmx_channels_fd_beams.restype = C.c_int
    lib.dmx_channels_fd_beams.argtypes = [C.POINTER(DmxParams), C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.dmx_beam_power.restype = C.c_int
    lib.dmx_beam_power.argtypes = [C.POINTER(DmxParams), C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dmx_p2m_count_rx.restype = C.c_int64
    lib.dmx_p2m_count_rx.argtypes = [C.c_void_p, C.c_size_t]
    lib.dmx_p2m_parse_paths.restype = C.c_int
    lib.dmx_p2m_parse_paths.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int64] + [C.c_void_p] * 9
    lib.dmx_pathloss.restype = C.c_int
    lib.dmx_pathloss.argtypes = [C.POINTER(DmxRays), C.c_int32, C.c_void_p, C.c_void_p]
    lib.dmx_mat5_find.restype = C.c_int
    lib.dmx_mat5_find.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(DmxMatInfo)]
    lib.dmx_mat_to_rowmajor_f32.restype = C.c_int
    lib.dmx_mat_to_rowmajor_f32.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                            C.c_int32, C.c_void_p, C.c_void_p]
    lib.dmx_mats_to_device.restype = C.c_int
    lib.dmx_mats_to_device.argtypes = [C.POINTER(DmxMatJob), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    v = lib.dmx_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"deepmimo_amd: ABI mismatch, library reports {v}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().dmx_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what} failed (status {rc}): {msg}")
