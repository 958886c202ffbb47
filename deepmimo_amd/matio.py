"""Device-direct reader of the reference's ray-matrix files.

The reference loads every matrix with ``scipy.io.loadmat`` and slices it in NumPy
(deepmimo/generator/core.py:241-254).  Here the file is memory-mapped, the array payload is located by
the C-ABI's MAT-v5 parser (``dmx_mat5_find``), copied to HBM exactly as stored (column-major) and
turned into the row-major float32 ``[n_sel, max_paths]`` matrix the kernels read by one device pass
(``dmx_mat_to_rowmajor_f32``) that also applies the receiver selection and the ``max_paths`` trim.
No NumPy copy of the matrix is made on the host (the H2D copy reads the mapped file pages).
"""
from __future__ import annotations

import ctypes as C
import mmap
import zlib
from typing import Optional, Tuple

import numpy as np
import torch

from . import _native as nat

_FAKE_HEADER = bytes(126) + b"IM"      # level-5 header stub (little-endian mark) for inflated elements


def find_array(buf, name: Optional[str]) -> Tuple[nat.DmxMatInfo, object]:
    """Locate variable `name` in a MAT-v5 file image.  Returns (info, image the offsets refer to);
    a zlib-compressed element (scipy's do_compression=True) is inflated first."""
    lib = nat.load()
    info = nat.DmxMatInfo()
    arr = np.frombuffer(buf, dtype=np.uint8)
    key = None if name is None else name.encode()
    rc = lib.dmx_mat5_find(C.c_void_p(arr.ctypes.data), arr.size, key, C.byref(info))
    if rc != 0 and info.compressed:
        raw = zlib.decompress(bytes(arr[info.comp_offset: info.comp_offset + info.comp_bytes]))
        buf = _FAKE_HEADER + raw
        arr = np.frombuffer(buf, dtype=np.uint8)
        info = nat.DmxMatInfo()
        rc = lib.dmx_mat5_find(C.c_void_p(arr.ctypes.data), arr.size, key, C.byref(info))
    nat.check(rc, f"dmx_mat5_find('{name}')")
    return info, buf


def load_matrix_to_device(path: str, key: str, device, rx_idxs=None, max_paths: Optional[int] = None) -> torch.Tensor:
    """One 2-D ray matrix file -> float32 [n_sel, min(max_paths, cols)] tensor on `device`."""
    lib = nat.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise ValueError("load_matrix_to_device needs a GPU device")
    import warnings
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        info, image = find_array(mm, key)
        if info.ndim != 2:
            raise ValueError(f"{path}: '{key}' has {info.ndim} dimensions, expected a 2-D ray matrix")
        rows, cols = int(info.dims[0]), int(info.dims[1])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")        # read-only buffer: it is only the source of one H2D copy
            view = torch.frombuffer(image, dtype=torch.uint8, count=int(info.data_bytes), offset=int(info.data_offset))
        d_payload = view.to(dev)                   # straight from the mapped pages (or the inflated element) to HBM
        del view, image
    keep = cols if max_paths is None else min(int(max_paths), cols)
    if rx_idxs is None:
        d_idx, n_sel = None, rows
    else:
        idx = np.asarray(rx_idxs, dtype=np.int64).ravel()
        if idx.size and (idx.min() < 0 or idx.max() >= rows):
            raise IndexError(f"{path}: receiver index out of range for {rows} stored receivers")
        d_idx, n_sel = torch.from_numpy(idx).to(dev), int(idx.size)
    out = torch.empty((n_sel, keep), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dmx_mat_to_rowmajor_f32(C.c_void_p(d_payload.data_ptr()), info.data_type, rows, cols,
                                         None if d_idx is None else C.c_void_p(d_idx.data_ptr()), n_sel, keep,
                                         C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    nat.check(rc, "dmx_mat_to_rowmajor_f32")
    torch.cuda.current_stream(dev).synchronize()      # d_payload / d_idx may be freed after return
    return out
