"""Wireless InSite ``*.paths.*.p2m`` -> ray matrices.

``paths_parser(file)`` returns the same dictionary as the reference's
``deepmimo.converter.wireless_insite.p2m_parser.paths_parser`` (p2m_parser.py:48-145): float32
``aoa_az, aoa_el, aod_az, aod_el, delay, power, phase, inter`` of shape [n_rx, n_paths_max] and ``inter_pos``
[n_rx, n_paths_max, n_bounces_max, 3], NaN-padded, already trimmed like ``compress_path_data``
(converter_utils.py:167-197).  The text is parsed by the C-ABI's host-side C++ parser over a memory-mapped
file (dmx_p2m_parse_paths); only the final trim is NumPy slicing."""
from __future__ import annotations

import ctypes as C
import mmap
from typing import Dict

import numpy as np

from . import _native as nat
from . import consts as c

MAX_INTER_PER_PATH = 10          # deepmimo/consts.py:181
_KEYS = (c.AOA_AZ_PARAM_NAME, c.AOA_EL_PARAM_NAME, c.AOD_AZ_PARAM_NAME, c.AOD_EL_PARAM_NAME, c.DELAY_PARAM_NAME,
         c.POWER_PARAM_NAME, c.PHASE_PARAM_NAME, c.INTERACTIONS_PARAM_NAME)


def paths_parser(file: str, max_paths: int = c.MAX_PATHS, max_inter: int = MAX_INTER_PER_PATH) -> Dict[str, np.ndarray]:
    lib = nat.load()
    with open(file, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            view = np.frombuffer(mm, dtype=np.uint8)
            ptr, n = C.c_void_p(view.ctypes.data), view.size
            n_rx = int(lib.dmx_p2m_count_rx(ptr, n))
            if n_rx < 0:
                nat.check(-1, "dmx_p2m_count_rx")
            data = {k: np.empty((n_rx, max_paths), dtype=np.float32) for k in _KEYS}
            data[c.INTERACTIONS_POS_PARAM_NAME] = np.empty((n_rx, max_paths, max_inter, 3), dtype=np.float32)
            args = [C.c_void_p(data[k].ctypes.data) for k in _KEYS] + [C.c_void_p(data[c.INTERACTIONS_POS_PARAM_NAME].ctypes.data)]
            rc = lib.dmx_p2m_parse_paths(ptr, n, int(max_paths), int(max_inter), n_rx, *args)
            nat.check(rc, "dmx_p2m_parse_paths")
        finally:
            del view
            mm.close()
    return compress_path_data(data, max_paths)


def compress_path_data(data: Dict[str, np.ndarray], max_paths_cap: int = c.MAX_PATHS) -> Dict[str, np.ndarray]:
    """converter_utils.py:167-238: keep paths up to the first index that is NaN for every receiver, and as many
    interaction slots as the longest interaction code has digits."""
    all_nan = np.all(np.isnan(data[c.AOA_AZ_PARAM_NAME]), axis=0)
    first = np.where(all_nan)[0]
    n_paths = int(first[0]) if len(first) else max_paths_cap
    inter = data[c.INTERACTIONS_PARAM_NAME]
    bounces = np.zeros_like(inter, dtype=int)
    with np.errstate(invalid="ignore"):
        nz = inter > 0
    bounces[nz] = np.floor(np.log10(inter[nz])).astype(int) + 1
    max_b = int(bounces.max()) if bounces.size else 0
    out = {}
    for k, v in data.items():
        if v.ndim >= 2:
            v = v[:, :n_paths, ...]
        if v.ndim >= 3:
            v = v[:, :n_paths, :max_b]
        out[k] = v
    return out
