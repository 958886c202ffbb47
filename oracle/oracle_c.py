"""ctypes front end of the oracle's C restatement (oracle/oracle_c.c -> oracle/lib/liboracle_c.so).

TEST INFRASTRUCTURE ONLY (same rule as oracle_np.py): used by tests/ and bench.py's cpu_baseline leg as a checker /
timed CPU baseline, never by deepmimo_amd/.  ``compute_channels`` has oracle_np.compute_channels' signature and result
keys, so the golden tests run both restatements through the same assertions.  The struct mirrors below are written
out independently of deepmimo_amd/_native.py on purpose (tests/test_host_cpu.py compares their sizes): the twin
entry points dmx_cpu_* take include/deepmimo_amd.h's structs with host pointers.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import oracle_np as onp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liboracle_c.so")
SYMBOLS = ("dmx_cpu_version", "dmx_cpu_workspace_bytes", "dmx_cpu_path_prep", "dmx_cpu_channels_fd",
           "dmx_cpu_channels_td", "dmx_cpu_decode_max_delay", "dmx_cpu_set_threads")

_F = C.POINTER(C.c_float)
_D = C.POINTER(C.c_double)


class Rays(C.Structure):
    _fields_ = [("n_ue", C.c_int64), ("n_paths", C.c_int32), ("ld", C.c_int32)] + \
               [(k, _F) for k in ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter",
                                  "doppler_vel", "doppler_acc")]


class Params(C.Structure):
    _fields_ = [("bs_shape", C.c_int32 * 2), ("ue_shape", C.c_int32 * 2),
                ("bs_spacing", C.c_double), ("ue_spacing", C.c_double),
                ("bs_rotation", C.c_double * 3), ("ue_rotation", C.c_double * 3),
                ("ue_rotation_per_user", _D),
                ("bs_pattern", C.c_int32), ("ue_pattern", C.c_int32),
                ("fov_enabled", C.c_int32), ("bs_fov_restricted", C.c_int32), ("ue_fov_restricted", C.c_int32),
                ("bs_fov", C.c_double * 2), ("ue_fov", C.c_double * 2),
                ("num_paths", C.c_int32), ("freq_domain", C.c_int32),
                ("n_subcarriers", C.c_int32), ("n_selected", C.c_int32),
                ("selected_subcarriers", C.POINTER(C.c_int32)),
                ("bandwidth", C.c_double), ("rx_filter", C.c_int32), ("enable_doppler", C.c_int32),
                ("carrier_freq", C.c_double),
                ("sc_first", C.c_int32), ("sc_stride", C.c_int32),     # ABI 2 hint; the twins read the array itself
                ("flags", C.c_uint32), ("reserved0", C.c_uint32)]      # ABI 3 arithmetic mode; the twins sum in complex128


class Side(C.Structure):
    _fields_ = [("fov_mask", C.POINTER(C.c_uint8)), ("num_paths", C.POINTER(C.c_int32)),
                ("los", C.POINTER(C.c_int32)),
                ("aod_el_rot", _D), ("aod_az_rot", _D), ("aoa_el_rot", _D), ("aoa_az_rot", _D),
                ("power_linear", _F), ("power_linear_ant_gain", _D),
                ("max_delay_key", C.POINTER(C.c_uint32))]


_lib = None


def build() -> None:
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def load():
    """The C oracle library; built on first use if the snapshot does not carry it (gcc only, ~1 s)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.dmx_cpu_workspace_bytes.restype = C.c_size_t
        lib.dmx_cpu_workspace_bytes.argtypes = [C.POINTER(Params), C.c_int64, C.c_int32]
        lib.dmx_cpu_path_prep.argtypes = [C.POINTER(Rays), C.POINTER(Params), C.c_void_p, C.c_size_t,
                                          C.POINTER(Side), C.c_void_p]
        lib.dmx_cpu_channels_fd.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                            C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
        lib.dmx_cpu_channels_td.argtypes = [C.POINTER(Params), C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                            C.c_int64, C.c_void_p, C.c_void_p]
        lib.dmx_cpu_decode_max_delay.restype = C.c_float
        lib.dmx_cpu_decode_max_delay.argtypes = [C.c_uint32]
        _lib = lib
    return _lib


_PATTERNS = {"isotropic": 0, "halfwave-dipole": 1}


def _pattern(name):
    if name not in _PATTERNS:
        raise NotImplementedError(f"The given '{name}' antenna radiation pattern is not applicable.")
    return _PATTERNS[name]


def _ptr(a, t):
    return a.ctypes.data_as(t)


def compute_channels(rays: dict, params: dict, bs_fov=None, ue_fov=None, doppler=None, users=None,
                     threads: int | None = None) -> dict:
    """Same contract as oracle_np.compute_channels; the arithmetic runs in oracle_c.c.  `threads` = OpenMP threads of the user loop (default 1: the reference is single-threaded)."""
    lib = load()
    lib.dmx_cpu_set_threads(int(threads) if threads else 1)
    keep = {}                                                    # arrays the structs point into
    n_ue, L = rays["power"].shape
    r = Rays(n_ue=n_ue, n_paths=L, ld=L)
    for k in onp.RAY_KEYS:
        keep[k] = np.ascontiguousarray(rays[k], np.float32)
        setattr(r, k, _ptr(keep[k], _F))
    use_dop = bool(params["enable_doppler"]) and doppler is not None
    if use_dop:
        keep["vel"] = np.ascontiguousarray(doppler["vel"], np.float32)
        keep["acc"] = np.ascontiguousarray(doppler["acc"], np.float32)
        r.doppler_vel, r.doppler_acc = _ptr(keep["vel"], _F), _ptr(keep["acc"], _F)

    bs, ue, ofdm = params["bs_antenna"], params["ue_antenna"], params["ofdm"]
    np.random.seed(1001)                                         # dataset.py:250
    ue_rot = onp.resolve_ue_rotation(ue["rotation"], n_ue)       # host-side RNG order, dataset.py:327-338
    p = Params()
    p.bs_shape[:] = [int(v) for v in bs["shape"]]
    p.ue_shape[:] = [int(v) for v in ue["shape"]]
    p.bs_spacing, p.ue_spacing = float(bs["spacing"]), float(ue["spacing"])
    p.bs_rotation[:] = np.deg2rad(np.asarray(bs["rotation"])).astype(np.float64).tolist()
    keep["ue_rot"] = np.ascontiguousarray(ue_rot, np.float64)
    p.ue_rotation_per_user = _ptr(keep["ue_rot"], _D)
    p.bs_pattern, p.ue_pattern = _pattern(bs["radiation_pattern"]), _pattern(ue["radiation_pattern"])
    if bs_fov is not None and ue_fov is None:                    # Dataset.apply_fov defaults, dataset.py:423-448
        ue_fov = np.array([360, 180])
    if ue_fov is not None and bs_fov is None:
        bs_fov = np.array([360, 180])
    bs_full = bs_fov is not None and onp.is_full_fov(bs_fov)
    ue_full = ue_fov is not None and onp.is_full_fov(ue_fov)
    fov_on = not ((bs_fov is None and ue_fov is None) or (bs_full and ue_full))
    p.fov_enabled = int(fov_on)
    if fov_on:
        p.bs_fov_restricted, p.ue_fov_restricted = int(not bs_full), int(not ue_full)
        p.bs_fov[:] = np.deg2rad(np.asarray(bs_fov)).astype(np.float64).tolist()
        p.ue_fov[:] = np.deg2rad(np.asarray(ue_fov)).astype(np.float64).tolist()
    P = min(int(params["num_paths"]), L)
    p.num_paths, p.freq_domain = int(params["num_paths"]), int(bool(params["freq_domain"]))
    keep["sc"] = np.ascontiguousarray(ofdm["selected_subcarriers"], np.int32)
    p.n_subcarriers, p.n_selected = int(ofdm["subcarriers"]), len(keep["sc"])
    p.selected_subcarriers = _ptr(keep["sc"], C.POINTER(C.c_int32))
    p.bandwidth, p.rx_filter = float(ofdm["bandwidth"]), int(bool(ofdm["rx_filter"]))
    p.enable_doppler = int(use_dop)
    p.carrier_freq = float(doppler["carrier_freq"]) if use_dop else 0.0

    out = {"ue_rotation": ue_rot,
           "_fov_mask": np.zeros((n_ue, L), np.uint8) if fov_on else None,
           "num_paths": np.zeros(n_ue, np.int32), "los": np.zeros(n_ue, np.int32),
           "power_linear": np.zeros((n_ue, L), np.float32), "_power_linear_ant_gain": np.zeros((n_ue, L), np.float64)}
    for k in ("_aod_el_rot", "_aod_az_rot", "_aoa_el_rot", "_aoa_az_rot"):
        out[k] = np.zeros((n_ue, L), np.float64)
    key = np.zeros(1, np.uint32)
    s = Side(num_paths=_ptr(out["num_paths"], C.POINTER(C.c_int32)), los=_ptr(out["los"], C.POINTER(C.c_int32)),
             aod_el_rot=_ptr(out["_aod_el_rot"], _D), aod_az_rot=_ptr(out["_aod_az_rot"], _D),
             aoa_el_rot=_ptr(out["_aoa_el_rot"], _D), aoa_az_rot=_ptr(out["_aoa_az_rot"], _D),
             power_linear=_ptr(out["power_linear"], _F),
             power_linear_ant_gain=_ptr(out["_power_linear_ant_gain"], _D),
             max_delay_key=_ptr(key, C.POINTER(C.c_uint32)))
    if fov_on:
        s.fov_mask = _ptr(out["_fov_mask"], C.POINTER(C.c_uint8))
    ws = np.zeros(max(lib.dmx_cpu_workspace_bytes(C.byref(p), n_ue, L), 8), np.uint8)
    rc = lib.dmx_cpu_path_prep(C.byref(r), C.byref(p), ws.ctypes.data, ws.size, C.byref(s), None)
    assert rc == 0, f"dmx_cpu_path_prep -> {rc}"
    if fov_on:
        out["_fov_mask"] = out["_fov_mask"].astype(bool)
    iso = p.bs_pattern == 0 and p.ue_pattern == 0
    if iso:                                                      # the reference keeps float32 here (ant_patterns.py:167)
        out["_power_linear_ant_gain"] = out["_power_linear_ant_gain"].astype(np.float32)

    if users is None:
        u0, cnt = 0, n_ue
    else:
        idx = np.arange(n_ue)[users]
        assert idx.size == 0 or np.array_equal(idx, np.arange(idx[0], idx[0] + idx.size)), "contiguous user range only"
        u0, cnt = (int(idx[0]), int(idx.size)) if idx.size else (0, 0)
    m_rx, m_tx = int(np.prod(ue["shape"])), int(np.prod(bs["shape"]))
    fd = bool(params["freq_domain"])
    H = np.zeros((cnt, m_rx, m_tx, p.n_selected if fd else P), np.complex64)
    if fd:
        rc = lib.dmx_cpu_channels_fd(C.byref(p), ws.ctypes.data, n_ue, L, u0, cnt, H.ctypes.data, 0, None)
        md = float(lib.dmx_cpu_decode_max_delay(int(key[0])))
        out["max_delay"] = md
        out["delay_exceeds_symbol"] = bool(md > ofdm["subcarriers"] / ofdm["bandwidth"])
    else:
        rc = lib.dmx_cpu_channels_td(C.byref(p), ws.ctypes.data, n_ue, L, u0, cnt, H.ctypes.data, None)
    assert rc == 0, f"dmx_cpu_channels -> {rc}"
    out["channel"] = H
    return out
