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


_PINNED = {"buf": None}          # host staging buffer, page-locked once and reused by every load of the process


def _staging(nbytes: int) -> torch.Tensor:
    buf = _PINNED["buf"]
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        _PINNED["buf"] = buf
    return buf


_MI_DTYPES = {1: np.int8, 2: np.uint8, 3: np.int16, 4: np.uint16, 5: np.int32, 6: np.uint32, 7: np.float32, 9: np.float64,
              12: np.int64, 13: np.uint64}                                    # miTYPE of the stored payload
_MX_DTYPES = {6: np.float64, 7: np.float32, 8: np.int8, 9: np.uint8, 10: np.int16, 11: np.uint16, 12: np.int32, 13: np.uint32,
              14: np.int64, 15: np.uint64}                                    # mxCLASS of the array = what loadmat returns


def read_matrix_host(path: str, key: str) -> np.ndarray:
    """One numeric array of a MAT-v5 file as `scipy.io.loadmat(path)[key]` returns it (class dtype, MATLAB dims), through
    the library's parser: `loadmat` spends 0.45 ms per file before it reads a byte, which is 0.9 of the 4.6 ms a
    device load of a 105-MB scenario takes (the two small position matrices)."""
    with open(path, "rb") as f:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        info, image = find_array(mm, key)
        dims = [int(info.dims[i]) for i in range(info.ndim)]
        a = np.frombuffer(image, dtype=_MI_DTYPES[int(info.data_type)], count=int(np.prod(dims)), offset=int(info.data_offset))
        return np.array(a.reshape(dims[::-1]).T, dtype=_MX_DTYPES[int(info.class_id)], order="F")


STAGING_CAP_BYTES = 1 << 30      # page-locked staging per pipeline run; larger scenarios go through in groups of files


def load_matrices_to_device(items, device, rx_idxs=None, max_paths: Optional[int] = None) -> dict:
    """[(path, key), ...] -> {key: tensor}: `_load_group` over groups of files whose payloads fit the staging cap together
    (a 10-million-receiver scenario has 1 GB per matrix: eight of them are not page-locked at once)."""
    import os
    items = list(items)
    out, group, size = {}, [], 0
    for it in items:
        n = os.path.getsize(it[0])
        if group and size + n > STAGING_CAP_BYTES:
            out.update(_load_group(group, device, rx_idxs, max_paths))
            group, size = [], 0
        group.append(it)
        size += n
    if group:
        out.update(_load_group(group, device, rx_idxs, max_paths))
    return out


def _load_group(items, device, rx_idxs=None, max_paths: Optional[int] = None) -> dict:
    """[(path, key), ...] 2-D ray matrix files -> {key: float32 [n_sel, min(max_paths, cols)] tensor on `device`}.

    All fields of a TX/RX pair go through ONE native pipeline, `dmx_mats_to_device` (core.py:241-254 loads them one
    `scipy.io.loadmat` at a time): reader threads of the library `pread` the payloads - every file in slices by all
    threads, file after file - into a page-locked staging buffer that the process keeps, while the calling thread queues,
    field by field as they complete, the asynchronous H2D copy and the layout kernel (`dmx_mat_to_rowmajor_f32`:
    column-major as stored -> row-major, receiver selection, `max_paths` trim); the receiver index list is uploaded once
    and the stream is synchronised once.  (Round 2 / early round 3: a pageable copy per field straight from the mapped
    file pages and one synchronisation per field, 7.0 ms per 105-MB scenario; the same pipeline with Python threads
    4.1-4.7 ms - the GIL between the issuing thread and the readers.)"""
    import os
    lib = nat.load()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise ValueError("load_matrices_to_device needs a GPU device")
    metas, total = [], 0
    for path, key in items:
        with open(path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            info, image = find_array(mm, key)
            if info.ndim != 2:
                raise ValueError(f"{path}: '{key}' has {info.ndim} dimensions, expected a 2-D ray matrix")
            inflated = None if image is mm else image          # a compressed element lives in memory, not in the file
            metas.append(dict(path=path, key=key, info=info, inflated=inflated, rows=int(info.dims[0]), cols=int(info.dims[1]),
                              nbytes=int(info.data_bytes), off=int(info.data_offset), stage=total))
            total += (int(info.data_bytes) + 255) // 256 * 256
            if inflated is None:
                try:
                    mm.close()                                  # the payload is read with preadv below
                except BufferError:
                    pass
    d_idx, n_idx = None, None
    if rx_idxs is not None:
        idx = np.asarray(rx_idxs, dtype=np.int64).ravel()
        for m in metas:
            if idx.size and (idx.min() < 0 or idx.max() >= m["rows"]):
                raise IndexError(f"{m['path']}: receiver index out of range for {m['rows']} stored receivers")
        d_idx, n_idx = torch.from_numpy(idx).to(dev), int(idx.size)
    stage = _staging(total)
    stage_np = stage.numpy()
    jobs = (nat.DmxMatJob * len(metas))()
    out, keepalive = {}, []
    stream = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        for j, m in zip(jobs, metas):
            keep = m["cols"] if max_paths is None else min(int(max_paths), m["cols"])
            n_sel = m["rows"] if d_idx is None else n_idx
            d_payload = torch.empty(m["nbytes"], dtype=torch.uint8, device=dev)
            o = torch.empty((n_sel, keep), dtype=torch.float32, device=dev)
            if m["inflated"] is not None:                       # not in the file as stored: staged here, path = NULL
                stage_np[m["stage"]: m["stage"] + m["nbytes"]] = np.frombuffer(m["inflated"], dtype=np.uint8, count=m["nbytes"],
                                                                              offset=m["off"])
            j.path = None if m["inflated"] is not None else os.fsencode(m["path"])
            j.file_offset, j.nbytes, j.stage_offset = m["off"], m["nbytes"], m["stage"]
            j.d_payload, j.d_out = d_payload.data_ptr(), o.data_ptr()
            j.data_type, j.cols_keep, j.rows, j.cols = int(m["info"].data_type), keep, m["rows"], m["cols"]
            keepalive.append(d_payload)
            out[m["key"]] = o
        rc = lib.dmx_mats_to_device(jobs, len(metas), C.c_void_p(stage.data_ptr()),
                                    None if d_idx is None else C.c_void_p(d_idx.data_ptr()), 0 if d_idx is None else n_idx,
                                    max(1, min(8, os.cpu_count() or 1)), C.c_void_p(stream.cuda_stream))
        stream.synchronize()                                    # staging buffer, d_payload, d_idx are free again
    nat.check(rc, "dmx_mats_to_device")
    return out


def load_matrix_to_device(path: str, key: str, device, rx_idxs=None, max_paths: Optional[int] = None) -> torch.Tensor:
    """One 2-D ray matrix file -> float32 [n_sel, min(max_paths, cols)] tensor on `device`."""
    return load_matrices_to_device([(path, key)], device, rx_idxs=rx_idxs, max_paths=max_paths)[key]
