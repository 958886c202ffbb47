"""Host driver of the MI355X channel-generation kernels.

``ChannelEngine`` owns nothing but a device index: it takes the float32 ray matrices a
``Dataset`` holds (core.py:209-219 layout), keeps them as PyTorch-ROCm tensors (PyTorch is the
allocator / stream provider, not the compute path), fills the C structs of
include/deepmimo_amd.h and calls the C-ABI:

    dmx_path_prep   -> per-path records + side products (LoS, path counts, FoV mask, angles, powers)
    dmx_channels_fd -> complex64 [N, M_rx, M_tx, K]     (dmx_channels_fd_lpf when rx_filter = 1)
    dmx_channels_td -> complex64 [N, M_rx, M_tx, P]

It replaces the body of Dataset.compute_channels (deepmimo/generator/dataset.py:224-268).
No CPU path exists here: without the shared library or without a GPU every entry point raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as nat
from . import consts as c


def _require_gpu(device_index: int) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("deepmimo_amd: no GPU visible (torch.cuda.is_available() is False); the channel-"
                           "generation path has no CPU fallback.")
    return torch.device("cuda", int(device_index))


@dataclass
class DeviceRays:
    """Ray matrices resident in HBM (float32 [N, L] each, contiguous)."""
    n_ue: int
    n_paths: int
    fields: Dict[str, torch.Tensor]
    doppler_vel: Optional[torch.Tensor] = None
    doppler_acc: Optional[torch.Tensor] = None


@dataclass
class PrepResult:
    workspace: torch.Tensor
    n_ue: int
    n_paths_loaded: int
    params_struct: nat.DmxParams
    keepalive: list = field(default_factory=list)
    side: Dict[str, torch.Tensor] = field(default_factory=dict)
    rays_struct: Optional[nat.DmxRays] = None
    side_struct: Optional[nat.DmxSide] = None
    workspace_bytes: int = 0


def is_full_fov(fov) -> bool:
    """dataset.py:450-459"""
    return fov[0] >= 360 and fov[1] >= 180


def uniform_stride(sel: np.ndarray):
    """(first, stride) when sel[k] == first + k * stride with stride > 0 for every k (dmx_params.sc_stride: lets the
    library pick the folded kernel without reading the device copy back), else (0, 0)."""
    sel = np.asarray(sel).astype(np.int64).ravel()
    if sel.size == 0 or abs(int(sel[0])) >= 2 ** 30:
        return 0, 0
    if sel.size == 1:
        return int(sel[0]), 1
    d = int(sel[1] - sel[0])
    if d <= 0 or d >= 2 ** 20 or not np.array_equal(sel, sel[0] + d * np.arange(sel.size)):
        return 0, 0
    return int(sel[0]), d


class ChannelEngine:
    def __init__(self, device_index: int = 0):
        self.lib = nat.load()
        self.device = _require_gpu(device_index)

    # ------------------------------------------------------------------ uploads
    def upload_rays(self, data) -> DeviceRays:
        """data: mapping with the eight float32 [N, L] matrices (numpy or torch)."""
        fields = {}
        shape = None
        for k in c.RAY_FIELDS:
            v = data[k]
            t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            if t.dim() != 2:
                raise ValueError(f"ray matrix '{k}' must be 2-D [n_ue, n_paths], got {tuple(t.shape)}")
            if shape is None:
                shape = tuple(t.shape)
            elif tuple(t.shape) != shape:
                raise ValueError(f"ray matrix '{k}' has shape {tuple(t.shape)}, expected {shape}")
            fields[k] = t
        dv = da = None
        keys = data.keys() if hasattr(data, "keys") else ()
        if c.DOPPLER_VEL_PARAM_NAME in keys and c.DOPPLER_ACC_PARAM_NAME in keys:
            dv = self._to_dev_f32(data[c.DOPPLER_VEL_PARAM_NAME])
            da = self._to_dev_f32(data[c.DOPPLER_ACC_PARAM_NAME])
        return DeviceRays(n_ue=shape[0], n_paths=shape[1], fields=fields, doppler_vel=dv, doppler_acc=da)

    def _to_dev_f32(self, v):
        t = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ C structs
    def _params_struct(self, params, bs_fov, ue_fov, ue_rot_per_user: Optional[torch.Tensor],
                       sel_dev: Optional[torch.Tensor], carrier_freq: float, have_doppler: bool,
                       sc_hint=(0, 0)) -> nat.DmxParams:
        bs, ue, ofdm = params[c.PARAMSET_ANT_BS], params[c.PARAMSET_ANT_UE], params[c.PARAMSET_OFDM]
        p = nat.DmxParams()
        p.bs_shape[0], p.bs_shape[1] = int(bs[c.PARAMSET_ANT_SHAPE][0]), int(bs[c.PARAMSET_ANT_SHAPE][1])
        p.ue_shape[0], p.ue_shape[1] = int(ue[c.PARAMSET_ANT_SHAPE][0]), int(ue[c.PARAMSET_ANT_SHAPE][1])
        p.bs_spacing, p.ue_spacing = float(bs[c.PARAMSET_ANT_SPACING]), float(ue[c.PARAMSET_ANT_SPACING])
        bs_rot = np.deg2rad(np.asarray(bs[c.PARAMSET_ANT_ROTATION]))          # geometry.py:286
        for i in range(3):
            p.bs_rotation[i] = float(bs_rot[i])
        if ue_rot_per_user is None:
            ue_rot = np.deg2rad(np.asarray(ue[c.PARAMSET_ANT_ROTATION]))
            for i in range(3):
                p.ue_rotation[i] = float(ue_rot[i])
            p.ue_rotation_per_user = None
        else:
            p.ue_rotation_per_user = ue_rot_per_user.data_ptr()
        for side, key in ((bs, "bs_pattern"), (ue, "ue_pattern")):
            name = side[c.PARAMSET_ANT_RAD_PAT]
            if name not in c.PARAMSET_ANT_RAD_PAT_VALS:                      # ant_patterns.py:119-120
                raise NotImplementedError(f"The given '{name}' antenna radiation pattern is not applicable.")
            setattr(p, key, nat.PATTERN_IDS[name])
        # FoV (dataset.py:477-504); apply_fov always stores both, a lone None is the full sphere
        if bs_fov is not None and ue_fov is None:
            ue_fov = np.array([360, 180])
        if ue_fov is not None and bs_fov is None:
            bs_fov = np.array([360, 180])
        bs_full = bs_fov is not None and is_full_fov(bs_fov)
        ue_full = ue_fov is not None and is_full_fov(ue_fov)
        enabled = not ((bs_fov is None and ue_fov is None) or (bs_full and ue_full))
        p.fov_enabled = int(enabled)
        if enabled:
            p.bs_fov_restricted, p.ue_fov_restricted = int(not bs_full), int(not ue_full)
            b, u = np.deg2rad(np.asarray(bs_fov)), np.deg2rad(np.asarray(ue_fov))   # geometry.py:184
            p.bs_fov[0], p.bs_fov[1], p.ue_fov[0], p.ue_fov[1] = float(b[0]), float(b[1]), float(u[0]), float(u[1])
        p.num_paths = int(params[c.PARAMSET_NUM_PATHS])
        p.freq_domain = int(bool(params[c.PARAMSET_FD_CH]))
        p.n_subcarriers = int(ofdm[c.PARAMSET_OFDM_SC_NUM])
        p.n_selected = 0 if sel_dev is None else int(sel_dev.numel())
        p.selected_subcarriers = None if sel_dev is None or sel_dev.numel() == 0 else sel_dev.data_ptr()
        p.bandwidth = float(ofdm[c.PARAMSET_OFDM_BANDWIDTH])
        p.rx_filter = int(bool(ofdm[c.PARAMSET_OFDM_LPF]))
        p.enable_doppler = int(bool(params[c.PARAMSET_DOPPLER_EN]) and have_doppler)
        p.carrier_freq = float(carrier_freq)
        p.sc_first, p.sc_stride = sc_hint
        return p

    def _stream_ptr(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ stage 1
    def prepare(self, rays: DeviceRays, params, bs_fov=None, ue_fov=None, ue_rotation_per_user=None,
                carrier_freq: float = 0.0, want_side=True, adaptive_terms: bool = False) -> PrepResult:
        """Run dmx_path_prep.  ue_rotation_per_user: optional [N, 3] degrees (numpy/torch).  want_side: True = every
        side product, "light" = LoS / path counts / FoV mask only, False = none.  adaptive_terms: opt into
        DMX_FLAG_ADAPTIVE_TERMS (include/deepmimo_amd.h: weak last path groups in one product term); the flag travels in
        the parameter block of the preparation, so every stage-2 call on it runs in the same mode."""
        dev = self.device
        n, L = rays.n_ue, rays.n_paths
        ofdm = params[c.PARAMSET_OFDM]
        keep = []
        sel = np.asarray(ofdm[c.PARAMSET_OFDM_SC_SAMP]).astype(np.int64).ravel()
        sel_dev = torch.from_numpy(sel.astype(np.int32)).to(dev)
        keep.append(sel_dev)
        rot_dev = None
        if ue_rotation_per_user is not None:
            r = ue_rotation_per_user
            r = r if isinstance(r, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(r, dtype=np.float64))
            rot_dev = r.to(device=dev, dtype=torch.float64).contiguous()
            if tuple(rot_dev.shape) != (n, 3):
                raise ValueError(f"per-user UE rotation must be [n_ue, 3], got {tuple(rot_dev.shape)}")
            keep.append(rot_dev)
        have_dop = rays.doppler_vel is not None and rays.doppler_acc is not None
        p = self._params_struct(params, bs_fov, ue_fov, rot_dev, sel_dev, carrier_freq, have_dop,
                                sc_hint=uniform_stride(sel))
        p.flags = nat.FLAG_ADAPTIVE_TERMS if adaptive_terms else 0

        r = nat.DmxRays()
        r.n_ue, r.n_paths, r.ld = n, L, L
        for k in c.RAY_FIELDS:
            setattr(r, k, rays.fields[k].data_ptr() if n * L > 0 else None)
        r.doppler_vel = rays.doppler_vel.data_ptr() if have_dop and n * L > 0 else None
        r.doppler_acc = rays.doppler_acc.data_ptr() if have_dop and n * L > 0 else None

        nbytes = int(self.lib.dmx_workspace_bytes(C.byref(p), n, L))
        ws = torch.empty(max(nbytes, 256) + 256, dtype=torch.uint8, device=dev)
        off = (-ws.data_ptr()) % 256
        ws = ws[off:off + max(nbytes, 256)]

        side = {}
        s = nat.DmxSide()
        if want_side:
            # "light": what is cheap beside the channel generation - LoS, path counts, the FoV mask when a FoV is set
            # (stage 1 then needs the angles as numbers anyway).  True: also the four rotated-angle matrices and the
            # powers (float64 arccos / atan2 per path: ~1 ms per 100k users x 25 paths, 120 MB of stores).
            side["fov_mask"] = torch.empty((n, L), dtype=torch.uint8, device=dev) if p.fov_enabled else None
            side["num_paths"] = torch.empty((n,), dtype=torch.int32, device=dev)
            side["los"] = torch.empty((n,), dtype=torch.int32, device=dev)
        if want_side is True:
            for k in ("aod_el_rot", "aod_az_rot", "aoa_el_rot", "aoa_az_rot", "power_linear_ant_gain"):
                side[k] = torch.empty((n, L), dtype=torch.float64, device=dev)
            side["power_linear"] = torch.empty((n, L), dtype=torch.float32, device=dev)
        side["max_delay_key"] = torch.zeros((1,), dtype=torch.int32, device=dev)
        for k, t in side.items():
            if t is not None and t.numel() > 0:
                setattr(s, k, t.data_ptr())
        with torch.cuda.device(dev):
            rc = self.lib.dmx_path_prep(C.byref(r), C.byref(p), C.c_void_p(ws.data_ptr()), nbytes, C.byref(s),
                                        self._stream_ptr())
        nat.check(rc, "dmx_path_prep")
        keep.extend(rays.fields.values())
        return PrepResult(workspace=ws, n_ue=n, n_paths_loaded=L, params_struct=p, keepalive=keep, side=side,
                          rays_struct=r, side_struct=s, workspace_bytes=nbytes)

    def relaunch(self, prep: PrepResult, out: torch.Tensor, variant: int = 0) -> torch.Tensor:
        """Re-issue stage 1 + stage 2 of an existing preparation on the current stream, reading whatever the
        ray tensors hold NOW.  No allocation, no host-device copy, no synchronisation: the two C-ABI calls only
        enqueue kernels, so this is what a HIP graph captures (tests/test_gpu_parity.py::test_hip_graph_replay)
        and what a serving loop calls per batch.  Frequency domain without rx_filter, or time domain."""
        p = prep.params_struct
        if p.freq_domain and p.rx_filter:
            raise ValueError("relaunch does not cover rx_filter = 1 (it needs a gains table per call)")
        shape = self.channel_shape(prep)
        if out.dtype != torch.complex64 or tuple(out.shape) != shape or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous complex64 tensor of shape {shape}")
        wsp = C.c_void_p(prep.workspace.data_ptr())
        stream = self._stream_ptr()
        with torch.cuda.device(self.device):
            prep.side["max_delay_key"].zero_()
            nat.check(self.lib.dmx_path_prep(C.byref(prep.rays_struct), C.byref(p), wsp, prep.workspace_bytes,
                                             C.byref(prep.side_struct), stream), "dmx_path_prep")
            if out.numel():
                if p.freq_domain:
                    rc = self.lib.dmx_channels_fd(C.byref(p), wsp, prep.n_ue, prep.n_paths_loaded, 0, prep.n_ue,
                                                  C.c_void_p(out.data_ptr()), int(variant), stream)
                else:
                    rc = self.lib.dmx_channels_td(C.byref(p), wsp, prep.n_ue, prep.n_paths_loaded, 0, prep.n_ue,
                                                  C.c_void_p(out.data_ptr()), stream)
                nat.check(rc, "stage 2")
        return out

    # ------------------------------------------------------------------ stage 2
    def channel_shape(self, prep: PrepResult, user_count: Optional[int] = None):
        p = prep.params_struct
        n = prep.n_ue if user_count is None else user_count
        m_rx, m_tx = p.ue_shape[0] * p.ue_shape[1], p.bs_shape[0] * p.bs_shape[1]
        last = p.n_selected if p.freq_domain else min(p.num_paths, prep.n_paths_loaded)
        return (n, m_rx, m_tx, last)

    def channels(self, prep: PrepResult, out: Optional[torch.Tensor] = None, user_begin: int = 0,
                 user_count: Optional[int] = None, variant: int = 0, tx_codebook=None) -> torch.Tensor:
        """Run stage 2 for users [user_begin, user_begin + user_count) into `out` (allocated if None).

        tx_codebook: optional complex [n_beams, M_tx] beamforming matrix F; the result is then the
        beam-space channel F @ H, complex64 [user_count, M_rx, n_beams, K], produced without ever writing H
        (dmx_channels_fd_beams; frequency domain without rx_filter only)."""
        p = prep.params_struct
        if user_count is None:
            user_count = prep.n_ue - user_begin
        shape = self.channel_shape(prep, user_count)
        cb = None
        if tx_codebook is not None:
            cb = tx_codebook if isinstance(tx_codebook, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(tx_codebook))
            cb = cb.to(device=self.device, dtype=torch.complex64).contiguous()
            if cb.dim() != 2 or cb.shape[1] != shape[2]:
                raise ValueError(f"tx_codebook must be [n_beams, {shape[2]}], got {tuple(cb.shape)}")
            if not p.freq_domain or p.rx_filter:
                raise ValueError("tx_codebook needs freq_domain = 1 and rx_filter = 0")
            shape = (shape[0], shape[1], int(cb.shape[0]), shape[3])
        if out is None:
            out = torch.empty(shape, dtype=torch.complex64, device=self.device)
        else:
            if out.dtype != torch.complex64 or tuple(out.shape) != shape or not out.is_contiguous():
                raise ValueError(f"out must be a contiguous complex64 tensor of shape {shape}")
        if out.numel() == 0:
            return out
        wsp = C.c_void_p(prep.workspace.data_ptr())
        with torch.cuda.device(self.device):
            if cb is not None:
                nb = int(cb.shape[0])
                nbytes = int(self.lib.dmx_beam_workspace_bytes(C.byref(p), user_count, prep.n_paths_loaded, nb))
                bws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
                off = (-bws.data_ptr()) % 256
                rc = self.lib.dmx_channels_fd_beams(C.byref(p), wsp, prep.n_ue, prep.n_paths_loaded, user_begin,
                                                    user_count, C.c_void_p(cb.data_ptr()), nb,
                                                    C.c_void_p(bws.data_ptr() + off), nbytes,
                                                    C.c_void_p(out.data_ptr()), self._stream_ptr())
                nat.check(rc, "dmx_channels_fd_beams")
                # bws / cb may go out of scope now: the launch is on torch's current stream and the caching
                # allocator reuses freed blocks in stream order, so the kernels still own them when they run
            elif p.freq_domain and p.rx_filter:
                nbytes = int(self.lib.dmx_lpf_workspace_bytes(C.byref(p), user_count, prep.n_paths_loaded))
                lws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
                off = (-lws.data_ptr()) % 256
                rc = self.lib.dmx_channels_fd_lpf(C.byref(p), wsp, prep.n_ue, prep.n_paths_loaded, user_begin,
                                                  user_count, C.c_void_p(lws.data_ptr() + off), nbytes,
                                                  C.c_void_p(out.data_ptr()), self._stream_ptr())
                nat.check(rc, "dmx_channels_fd_lpf")
            elif p.freq_domain:
                rc = self.lib.dmx_channels_fd(C.byref(p), wsp, prep.n_ue, prep.n_paths_loaded, user_begin, user_count,
                                              C.c_void_p(out.data_ptr()), int(variant), self._stream_ptr())
                nat.check(rc, "dmx_channels_fd")
            else:
                rc = self.lib.dmx_channels_td(C.byref(p), wsp, prep.n_ue, prep.n_paths_loaded, user_begin, user_count,
                                              C.c_void_p(out.data_ptr()), self._stream_ptr())
                nat.check(rc, "dmx_channels_td")
        return out

    # ------------------------------------------------------------------ stage 2 -> NumPy
    HOST_CHUNK_BYTES = 256 << 20          # per pipeline stage; two device buffers + two pinned staging buffers of this size
    HOST_COPY_THREADS = 8                 # np.copyto releases the GIL; one thread moves ~12 GB/s, PCIe Gen5 ~57 GB/s

    def channels_to_host(self, prep: PrepResult, variant: int = 0, tx_codebook=None, chunk_bytes: Optional[int] = None,
                         user_begin: int = 0, user_count: Optional[int] = None) -> np.ndarray:
        """Stage 2 straight into a NumPy array (what ``Dataset.compute_channels`` returns by default, as the reference
        does) as a three-stage pipeline over user chunks: the kernels of chunk c + 1 run on the current stream while
        chunk c crosses PCIe into one of two pinned staging buffers on a copy stream and chunk c - 1 is moved from the
        other staging buffer into the result by a few host threads (the first touch of the result's pages happens
        there too).  47 GB/s on the pool's boxes against 14 GB/s for ``tensor.cpu().numpy()``
        (tools/host_copy_probe.py), and the device never holds more than two chunks of the tensor."""
        if user_count is None:
            user_count = prep.n_ue - user_begin
        shape = self.channel_shape(prep, user_count)
        if tx_codebook is not None:
            shape = (shape[0], shape[1], int(tx_codebook.shape[0]), shape[3])
        n, per_user = shape[0], int(np.prod(shape[1:]))
        if n == 0 or per_user == 0:
            return np.empty(shape, dtype=np.complex64)
        if tx_codebook is not None and not isinstance(tx_codebook, torch.Tensor):
            tx_codebook = torch.from_numpy(np.ascontiguousarray(tx_codebook)).to(device=self.device, dtype=torch.complex64)
        chunk_bytes = int(chunk_bytes or self.HOST_CHUNK_BYTES)
        cu = max(1, chunk_bytes // (per_user * 8))
        if cu >= n:                                               # one chunk: nothing to overlap
            return self.channels(prep, user_begin=user_begin, user_count=n, variant=variant, tx_codebook=tx_codebook).cpu().numpy()
        result = np.empty(shape, dtype=np.complex64)
        flat = result.reshape(n, per_user)
        stage = self._host_stage(cu * per_user)
        stage_np = [t.numpy() for t in stage]
        dev = [torch.empty((cu,) + tuple(shape[1:]), dtype=torch.complex64, device=self.device) for _ in range(2)]
        main = torch.cuda.current_stream(self.device)
        side = self._copy_stream()
        pool = self._copy_pool()
        nthr = self.HOST_COPY_THREADS

        def drain(i, b, cnt, ev):
            ev.synchronize()                                      # chunk is in stage[i]
            src = stage_np[i][:cnt * per_user].reshape(cnt, per_user)
            per = (cnt + nthr - 1) // nthr
            jobs = [pool.submit(np.copyto, flat[b + k:b + min(cnt, k + per)], src[k:min(cnt, k + per)]) for k in range(0, cnt, per)]
            for j in jobs:
                j.result()

        pending = None
        for ci, b in enumerate(range(0, n, cu)):
            i, cnt = ci & 1, min(cu, n - b)
            # dev[i] / stage[i] last held chunk ci - 2, which was drained (hence copied) before this iteration
            self.channels(prep, out=dev[i][:cnt], user_begin=user_begin + b, user_count=cnt, variant=variant, tx_codebook=tx_codebook)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                stage[i][:cnt * per_user].copy_(dev[i][:cnt].reshape(-1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            if pending is not None:
                drain(*pending)
            pending = (i, b, cnt, ev)
        drain(*pending)
        main.wait_stream(side)                                    # dev[] goes back to the allocator in stream order
        return result

    def _host_stage(self, n_elems: int):
        st = getattr(self, "_stage", None)
        if st is None or st[0].numel() < n_elems:
            st = [torch.empty(n_elems, dtype=torch.complex64, pin_memory=True) for _ in range(2)]
            self._stage = st
        return st

    def _copy_stream(self):
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream

    def _copy_pool(self):
        if getattr(self, "_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.HOST_COPY_THREADS, thread_name_prefix="dmx-host-copy")
        return self._pool

    def beam_power(self, prep: PrepResult, tx_codebook, user_begin: int = 0, user_count: Optional[int] = None,
                   want_best: bool = True):
        """dmx_beam_power: the beam-sweep reduction of docs/manual.ipynb cell 105 without any [N, ., K] tensor.
        Returns (mean_amplitude float32 [user_count, n_beams], best_beam int32 [user_count] or None), both in HBM:
        mean_amplitude[u, b] = np.abs(F @ H[u]).mean(axis=0).mean(axis=-1)."""
        p = prep.params_struct
        if user_count is None:
            user_count = prep.n_ue - user_begin
        m_tx = p.bs_shape[0] * p.bs_shape[1]
        cb = tx_codebook if isinstance(tx_codebook, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(tx_codebook))
        cb = cb.to(device=self.device, dtype=torch.complex64).contiguous()
        if cb.dim() != 2 or cb.shape[1] != m_tx or cb.shape[0] < 1:
            raise ValueError(f"tx_codebook must be [n_beams, {m_tx}], got {tuple(cb.shape)}")
        if not p.freq_domain or p.rx_filter:
            raise ValueError("beam_power needs freq_domain = 1 and rx_filter = 0")
        nb = int(cb.shape[0])
        amp = torch.empty((user_count, nb), dtype=torch.float32, device=self.device)
        best = torch.empty((user_count,), dtype=torch.int32, device=self.device) if want_best else None
        if user_count == 0:
            return amp, best
        with torch.cuda.device(self.device):
            nbytes = int(self.lib.dmx_beam_workspace_bytes(C.byref(p), user_count, prep.n_paths_loaded, nb))
            bws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-bws.data_ptr()) % 256
            rc = self.lib.dmx_beam_power(C.byref(p), C.c_void_p(prep.workspace.data_ptr()), prep.n_ue, prep.n_paths_loaded,
                                         user_begin, user_count, C.c_void_p(cb.data_ptr()), nb,
                                         C.c_void_p(bws.data_ptr() + off), nbytes, C.c_void_p(amp.data_ptr()),
                                         C.c_void_p(best.data_ptr()) if want_best else None, self._stream_ptr())
            nat.check(rc, "dmx_beam_power")
        return amp, best

    def pathloss(self, rays: DeviceRays, coherent: bool = True) -> torch.Tensor:
        """dmx_pathloss: float32 [n_ue] dB (dataset.py:541-566)."""
        out = torch.empty((rays.n_ue,), dtype=torch.float32, device=self.device)
        r = nat.DmxRays()
        r.n_ue, r.n_paths, r.ld = rays.n_ue, rays.n_paths, rays.n_paths
        if rays.n_ue * rays.n_paths > 0:
            r.power, r.phase = rays.fields[c.POWER_PARAM_NAME].data_ptr(), rays.fields[c.PHASE_PARAM_NAME].data_ptr()
        with torch.cuda.device(self.device):
            rc = self.lib.dmx_pathloss(C.byref(r), int(bool(coherent)), C.c_void_p(out.data_ptr()), self._stream_ptr())
        nat.check(rc, "dmx_pathloss")
        return out

    def max_delay(self, prep: PrepResult) -> float:
        """nanmax(delay[:, :P]) as the kernel saw it (channel.py:231); synchronises."""
        key = int(prep.side["max_delay_key"].cpu().numpy().astype(np.uint32)[0])
        return float(self.lib.dmx_decode_max_delay(key))
