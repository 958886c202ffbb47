"""``Dataset`` / ``MacroDataset`` - the drop-in boundary of the channel-generation path.

Mirrors the parts of deepmimo/generator/dataset.py that orchestrate the hot path:

  * dict + attribute access with lazy computed attributes and the alias table
    (dataset.py:130-182, consts.py:261-322);
  * ``set_channel_params`` / ``compute_channels`` / ``apply_fov`` and the cache invalidation
    rules (dataset.py:197-268, 358-378, 423-448, 515-535);
  * the computed attributes ``los``, ``num_paths``, ``power_linear``, ``_power_linear_ant_gain``,
    rotated and FoV-filtered angles, ``_fov_mask``, ``channel`` (dataset.py:831-869);
  * ``MacroDataset`` fan-out (dataset.py:888-998).

Every number those attributes hold is produced on the GPU by the C-ABI library (engine.py ->
csrc/*.hip); this module is glue (names, caching, dtype of the returned arrays, RNG order for the
random UE rotation).  There is no NumPy implementation of the channel math in this package.

Deliberate deviation from the reference: ``compute_channels`` re-derives all
per-path side products from the current parameters on every call, so a changed radiation pattern
takes effect (the reference only invalidates its ``_power_linear_ant_gain`` cache on a rotation
or FoV change - dataset.py:213-220 - and would silently reuse the stale array).
"""
from __future__ import annotations

import inspect
from typing import Any, Dict, Optional

import numpy as np

from . import consts as c
from .channel import ChannelGenParameters
from .config import config
from .general_utils import DotDict

SHARED_PARAMS = [c.SCENE_PARAM_NAME, c.MATERIALS_PARAM_NAME, c.LOAD_PARAMS_PARAM_NAME, c.RT_PARAMS_PARAM_NAME]

_ROT_KEYS = (c.AOD_EL_ROT_PARAM_NAME, c.AOD_AZ_ROT_PARAM_NAME, c.AOA_EL_ROT_PARAM_NAME, c.AOA_AZ_ROT_PARAM_NAME)
_FOV_ANGLE_KEYS = (c.AOD_EL_FOV_PARAM_NAME, c.AOD_AZ_FOV_PARAM_NAME, c.AOA_EL_FOV_PARAM_NAME, c.AOA_AZ_FOV_PARAM_NAME)
_UE_ROT_RESOLVED = "_ue_rotation_resolved"     # [n_ue, 3] degrees actually used for the cached rotated angles
_PATTERNS_IN_EFFECT = "_patterns_in_effect"    # (bs, ue) radiation patterns behind the cached `_power_linear_ant_gain`

_engines: Dict[int, Any] = {}


class _DeviceSide:
    """Placeholder of a stage-1 side product that is still in HBM.  It sits in the Dataset's dict under the
    product's key (so ``in`` / ``keys()`` behave as in the reference, where the array is cached eagerly) and is
    replaced by the NumPy array the reference would hold the first time anything reads it."""
    __slots__ = ("make",)

    def __init__(self, make):
        self.make = make


def _engine():
    """Process-wide ChannelEngine for config('gpu_device_id').  Raises when use_gpu is off."""
    if not config.get("use_gpu", True):
        raise RuntimeError("deepmimo_amd has no CPU channel-generation path: set dm.config('use_gpu', True).")
    dev = int(config.get("gpu_device_id", 0))
    if dev not in _engines:
        from .engine import ChannelEngine
        _engines[dev] = ChannelEngine(dev)
    return _engines[dev]


class Dataset(DotDict):
    """One (TX, RX-set) pair of ray-tracing results plus everything computed from it."""

    def __init__(self, data: Optional[Dict[str, Any]] = None):
        super().__init__(data or {})

    # -------------------------------------------------------------- lookup chain (dataset.py:130-182)
    def _host(self, key: str) -> Any:
        """self._data[key], copied out of HBM first if it is still a device-side placeholder."""
        v = self._data[key]
        if isinstance(v, _DeviceSide):
            v = v.make()
            self._data[key] = v
        return v

    def _host_all(self) -> None:
        for k in [k for k, v in self._data.items() if isinstance(v, _DeviceSide)]:
            self._host(k)

    def __getattr__(self, key: str) -> Any:
        if key == "_data" or key.startswith("__"):
            raise AttributeError(key)
        try:
            return self._host(key)
        except KeyError:
            return self._resolve_key(key)

    def __getitem__(self, key: str) -> Any:
        try:
            return self._host(key)
        except KeyError:
            return self._resolve_key(key)

    def get(self, key: str, default: Any = None) -> Any:
        return self._host(key) if key in self._data else default

    def values(self):
        self._host_all()
        return self._data.values()

    def items(self):
        self._host_all()
        return self._data.items()

    def to_dict(self) -> Dict:
        self._host_all()
        return super().to_dict()

    def deepcopy(self):
        self._host_all()
        return super().deepcopy()

    def __repr__(self) -> str:
        self._host_all()
        return super().__repr__()

    def _resolve_key(self, key: str) -> Any:
        resolved = c.DATASET_ALIASES.get(key, key)
        if resolved != key:
            key = resolved
            if key in self._data:
                return self._host(key)
        if key in self._computed_attributes:
            value = getattr(self, self._computed_attributes[key])()
            if isinstance(value, dict):
                self.update(value)
                return self._host(key)
            self[key] = value
            return value
        raise KeyError(key)

    def __dir__(self):
        return sorted(set(list(super().__dir__()) + list(self._computed_attributes.keys()) +
                          list(c.DATASET_ALIASES.keys())))

    # -------------------------------------------------------------- channel parameters
    def set_channel_params(self, params: Optional[ChannelGenParameters] = None):
        """dataset.py:197-222: validate, store a deep copy, drop rotation-dependent caches when a
        rotation changed."""
        if params is None:
            params = ChannelGenParameters()
        params.validate(self.n_ue)
        old = self._data.get(c.CH_PARAMS_PARAM_NAME)
        self.ch_params = params.deepcopy()
        if old is not None:
            rot = c.PARAMSET_ANT_ROTATION
            if (not np.array_equal(old.bs_antenna[rot], params.bs_antenna[rot]) or
                    not np.array_equal(old.ue_antenna[rot], params.ue_antenna[rot])):
                self._clear_cache_rotated_angles()
        return params

    # -------------------------------------------------------------- GPU passes
    def _resolved_ue_rotation(self):
        """Per-user UE rotation in degrees or None for a constant one (dataset.py:327-338).  A (3, 2)
        array is a [lo, hi] range drawn with the global NumPy RNG, exactly as the reference draws it;
        the draw is cached with the rotated angles (and dropped with them)."""
        rot = np.asarray(self.ch_params.ue_antenna[c.PARAMSET_ANT_ROTATION])
        if rot.ndim == 1 and rot.shape[0] == 3:
            return None
        cached = self._data.get(_UE_ROT_RESOLVED)
        if cached is not None:
            return cached
        if rot.ndim == 2 and rot.shape == (3, 2):
            rot = np.random.uniform(rot[:, 0], rot[:, 1], (self.n_ue, 3))
        rot = np.ascontiguousarray(rot, dtype=np.float64).reshape(-1, 3)
        self._data[_UE_ROT_RESOLVED] = rot
        return rot

    def _carrier_freq(self) -> float:
        rt = self._data.get(c.RT_PARAMS_PARAM_NAME)
        try:
            return float(rt[c.RT_PARAM_FREQUENCY]) if rt is not None else 0.0
        except Exception:
            return 0.0

    def _params_for_prep(self):
        """The parameters stage 1 runs with.  By default the current ones: a changed radiation pattern takes effect at
        once (deliberate deviation).  With ``config('strict_reference_cache', True)`` the
        reference's behaviour is reproduced instead: its `_power_linear_ant_gain` cache survives everything but a
        rotation or FoV change (dataset.py:213-220, 358-378, 515-535), so while that cache entry exists the patterns
        it was computed with stay in effect (pinned by tests/golden/aux_stale_cache.npz)."""
        params = self.ch_params
        pats = (params[c.PARAMSET_ANT_BS][c.PARAMSET_ANT_RAD_PAT], params[c.PARAMSET_ANT_UE][c.PARAMSET_ANT_RAD_PAT])
        held = self._data.get(_PATTERNS_IN_EFFECT)
        if (config.get("strict_reference_cache", False) and held is not None and held != pats
                and c.PWR_LINEAR_ANT_GAIN_PARAM_NAME in self._data):
            params = params.deepcopy()
            params[c.PARAMSET_ANT_BS][c.PARAMSET_ANT_RAD_PAT], params[c.PARAMSET_ANT_UE][c.PARAMSET_ANT_RAD_PAT] = held
            pats = held
        self._data[_PATTERNS_IN_EFFECT] = pats
        return params

    def _run_prep(self, want_side=True):
        """Stage 1 on the GPU; refreshes every per-path side product in the cache.  want_side="light" computes LoS,
        path counts and the FoV mask beside the records and leaves the rotated angles / powers to a second stage-1
        pass that runs only if one of them is ever read (`_store_side`)."""
        eng = _engine()
        params = self._params_for_prep()
        rays = eng.upload_rays(self)
        kw = dict(bs_fov=self._data.get("bs_fov"), ue_fov=self._data.get("ue_fov"),
                  ue_rotation_per_user=self._resolved_ue_rotation(), carrier_freq=self._carrier_freq(),
                  adaptive_terms=bool(config.get("adaptive_precision", False)))
        prep = eng.prepare(rays, params, want_side=want_side, **kw)
        if want_side:
            self._store_side(prep, None if want_side is True else (rays, params.deepcopy(), kw))
        return eng, prep

    def _store_side(self, prep, deferred=None) -> None:
        """Register every side product of stage 1 under the reference's cache keys.  The arrays stay in HBM
        (`_DeviceSide`) until something reads them: at 100k users x 25 paths they are ~120 MB of device-to-host
        copies that a caller who only wants the channel tensor never pays for.  With `deferred` = (device rays, params,
        prepare keywords) the rotated angles and powers are not even computed yet: the first read of any of them
        re-runs stage 1 on the SAME uploaded rays (80 MB per 100k users x 25 paths stay in HBM until then) with the same
        parameters, FoV and resolved rotations."""
        memo: Dict[str, np.ndarray] = {}
        heavy: Dict[str, Any] = {}

        def side(name):
            if name in prep.side:
                return prep.side[name]
            if not heavy:                                 # second stage-1 pass, once, for all deferred products
                heavy.update(_engine().prepare(deferred[0], deferred[1], want_side=True, **deferred[2]).side)
            return heavy[name]

        def host(name):                                   # one D2H copy per product, shared by its dependants
            if name not in memo:
                memo[name] = side(name).cpu().numpy()
            return memo[name]

        has_mask = prep.side["fov_mask"] is not None
        d = self._data
        d[c.FOV_MASK_PARAM_NAME] = _DeviceSide(lambda: host("fov_mask").astype(bool)) if has_mask else None
        for k_rot, k_fov, name in zip(_ROT_KEYS, _FOV_ANGLE_KEYS, ("aod_el_rot", "aod_az_rot", "aoa_el_rot", "aoa_az_rot")):
            d[k_rot] = _DeviceSide(lambda name=name: host(name))
            if has_mask:                                                      # dataset.py:506-511
                d[k_fov] = _DeviceSide(lambda name=name: np.where(host("fov_mask").astype(bool), host(name), np.nan))
            else:
                d[k_fov] = _DeviceSide(lambda name=name: host(name))
        d[c.PWR_LINEAR_PARAM_NAME] = _DeviceSide(lambda: host("power_linear"))
        iso = all(pat == c.PARAMSET_ANT_RAD_PAT_VALS[0] for pat in self._data[_PATTERNS_IN_EFFECT])
        # float32 * 1.0 stays float32 in the reference (ant_patterns.py:167-168)
        d[c.PWR_LINEAR_ANT_GAIN_PARAM_NAME] = _DeviceSide(
            lambda: host("power_linear_ant_gain").astype(np.float32) if iso else host("power_linear_ant_gain"))
        d[c.NUM_PATHS_PARAM_NAME] = _DeviceSide(lambda: host("num_paths").astype(np.int64))
        d[c.LOS_PARAM_NAME] = _DeviceSide(lambda: host("los").astype(np.int64))

    @staticmethod
    def _guard_host_copy(nbytes: int) -> None:
        """Refuse a device-to-host copy that cannot fit: at the headline shape the channel tensor is 1 MB per user
        (105 GB per 100k users) and `.cpu()` of it would take the process down with the host's OOM killer."""
        if not config.get("host_copy_guard", True):
            return
        avail = None
        try:
            with open("/proc/meminfo") as f:
                for line in f:
                    if line.startswith("MemAvailable:"):
                        avail = int(line.split()[1]) * 1024
                        break
        except OSError:
            pass
        if avail is not None and nbytes > 0.9 * avail:
            raise MemoryError(
                f"the channel tensor is {nbytes / 1e9:.1f} GB but only {avail / 1e9:.1f} GB of host memory are available: "
                "keep it on the GPU with dm.config('channel_output', 'torch'), or generate it in user chunks with "
                "Dataset.iter_channels(params, chunk_users=...); dm.config('host_copy_guard', False) disables this check.")

    def __getstate__(self):
        """Pickling: side products still in HBM are copied out first (their placeholders are closures)."""
        self._host_all()
        return {"_data": self._data}

    def __setstate__(self, state):
        object.__setattr__(self, "_data", state["_data"])

    def compute_channels(self, params: Optional[ChannelGenParameters] = None):
        """dataset.py:224-268.  Returns complex64 [n_ue, M_rx, M_tx, K] (freq_domain) or
        [n_ue, M_rx, M_tx, num_paths]: a NumPy array by default, the HBM-resident torch tensor when
        ``config('channel_output') == 'torch'``.  Cached as ``dataset.channel``."""
        if params is None:
            params = ChannelGenParameters() if self._data.get(c.CH_PARAMS_PARAM_NAME) is None else self.ch_params
        self.set_channel_params(params)
        np.random.seed(1001)                                                   # dataset.py:250
        to_host = config.get("channel_output", "numpy") != "torch"
        if to_host:                                                            # before any GPU work is spent on it
            ofdm_, n_ant = params[c.PARAMSET_OFDM], [int(np.prod(params[s_][c.PARAMSET_ANT_SHAPE])) for s_ in (c.PARAMSET_ANT_BS, c.PARAMSET_ANT_UE)]
            last = len(np.atleast_1d(ofdm_[c.PARAMSET_OFDM_SC_SAMP])) if params[c.PARAMSET_FD_CH] else int(params[c.PARAMSET_NUM_PATHS])
            self._guard_host_copy(8 * int(self.n_ue) * n_ant[0] * n_ant[1] * last)
        eng, prep = self._run_prep(want_side="light")
        variant = int(config.get("fd_kernel_variant", 0))
        out = eng.channels_to_host(prep, variant=variant) if to_host else eng.channels(prep, variant=variant)
        ofdm = params[c.PARAMSET_OFDM]
        if params[c.PARAMSET_FD_CH]:
            self._warn_symbol_duration(eng.max_delay(prep), ofdm)
        self[c.CHANNEL_PARAM_NAME] = out
        return out

    def iter_channels(self, params: Optional[ChannelGenParameters] = None, chunk_users: int = 4096):
        """Generate channels in user chunks (extension): yields ``(user_begin, H_chunk)`` with ``H_chunk`` a
        NumPy complex64 array [<= chunk_users, M_rx, M_tx, K].  For scenarios whose full tensor (1 MB per user at
        64x4 antennas x 512 subcarriers) fits neither host memory nor one NumPy array; stage 1 runs once, stage 2
        per chunk through the C-ABI's user_begin / user_count range.  Nothing is cached as ``channel``."""
        if params is None:
            params = ChannelGenParameters() if self._data.get(c.CH_PARAMS_PARAM_NAME) is None else self.ch_params
        self.set_channel_params(params)
        np.random.seed(1001)
        eng, prep = self._run_prep(want_side="light")
        n = prep.n_ue
        variant = int(config.get("fd_kernel_variant", 0))
        for b in range(0, n, max(1, int(chunk_users))):
            cnt = min(int(chunk_users), n - b)
            yield b, eng.channels_to_host(prep, variant=variant, user_begin=b, user_count=cnt)

    def compute_beam_channels(self, codebook, params: Optional[ChannelGenParameters] = None):
        """Beam-space channels ``codebook @ H`` for a TX codebook [n_beams, M_tx] (rows e.g. from
        ``dm.steering_vec``), complex64 [n_ue, M_rx, n_beams, K] - what docs/manual.ipynb cell 105 computes
        as ``F1 @ dataset.channel`` - without materialising H (extension).  Not cached."""
        if params is None:
            params = ChannelGenParameters() if self._data.get(c.CH_PARAMS_PARAM_NAME) is None else self.ch_params
        self.set_channel_params(params)
        np.random.seed(1001)
        eng, prep = self._run_prep(want_side="light")
        if config.get("channel_output", "numpy") == "torch":
            return eng.channels(prep, tx_codebook=codebook)
        cb = codebook if hasattr(codebook, "shape") else np.asarray(codebook)
        return eng.channels_to_host(prep, tx_codebook=cb)

    def compute_beam_power(self, codebook, params: Optional[ChannelGenParameters] = None, return_best: bool = False):
        """Received power per beam of a TX codebook [n_beams, M_tx] - the beam sweep of docs/manual.ipynb cells
        105 / 110 / 112 in one call, with no channel tensor written anywhere (extension):

            mean_amplitude  = np.abs(codebook @ dataset.channel).mean(axis=1).mean(axis=-1)   # computed on the GPU
            recv_bf_pwr_dbm = np.around(20 * np.log10(mean_amplitude) + 30, 1), NaN where dataset.los == -1
            best_beams      = np.argmax(recv_bf_pwr_dbm, axis=1) as float, NaN where dataset.los == -1

        Returns ``recv_bf_pwr_dbm`` float64 [n_ue, n_beams] (and ``best_beams`` with ``return_best``).  The raw means stay
        available as ``dataset['beam_mean_amplitude']`` (float32 [n_ue, n_beams])."""
        if params is None:
            params = ChannelGenParameters() if self._data.get(c.CH_PARAMS_PARAM_NAME) is None else self.ch_params
        self.set_channel_params(params)
        np.random.seed(1001)
        eng, prep = self._run_prep(want_side="light")
        amp, _ = eng.beam_power(prep, codebook, want_best=False)
        amp = amp.cpu().numpy()
        self._data["beam_mean_amplitude"] = amp
        no_paths = self[c.LOS_PARAM_NAME] == -1
        pwr = np.zeros(amp.shape) * np.nan                                      # float64, as the notebook allocates it
        with np.errstate(divide="ignore"):
            pwr[~no_paths] = np.around(20 * np.log10(amp[~no_paths]) + 30, 1)
        if not return_best:
            return pwr
        best = np.argmax(pwr, axis=1).astype(float) if pwr.shape[1] else np.zeros(len(pwr))
        best[np.isnan(pwr[:, 0])] = np.nan
        return pwr, best

    def compute_pathloss(self, coherent: bool = True) -> np.ndarray:
        """Path loss in dB assuming 0 dBm transmitted power (dataset.py:541-566); cached as ``pathloss``."""
        eng = _engine()
        pl = eng.pathloss(eng.upload_rays(self), coherent).cpu().numpy()
        self[c.PATHLOSS_PARAM_NAME] = pl
        return pl

    @staticmethod
    def _warn_symbol_duration(max_delay: float, ofdm) -> None:
        """The reference's clipping warning (channel.py:228-250), fed by the device-side max."""
        n_sc, bw = ofdm[c.PARAMSET_OFDM_SC_NUM], ofdm[c.PARAMSET_OFDM_BANDWIDTH]
        symbol = n_sc / bw
        if not (max_delay > symbol):
            return
        print("\nWarning: Some path delays exceed OFDM symbol duration")
        print("-" * 50)
        print("OFDM Configuration:")
        print(f"- Number of subcarriers (N): {n_sc}")
        print(f"- Bandwidth (B): {bw/1e6:.1f} MHz")
        print(f"- Subcarrier spacing (Δf = B/N): {bw/n_sc/1e3:.1f} kHz")
        print(f"- Symbol duration (T = 1/Δf = N/B): {symbol*1e6:.1f} μs")
        print("\nPath Information:")
        print(f"- Maximum path delay: {max_delay*1e6:.1f} μs")
        print(f"- Excess delay: {(max_delay - symbol)*1e6:.1f} μs")
        print("\nPaths arriving after the symbol duration will be clipped.")
        print("To avoid clipping, either:")
        print("1. Increase the number of subcarriers (N)")
        print("2. Decrease the bandwidth (B)")
        print(f"3. Switch to time-domain channel generation (set ch_params['{c.PARAMSET_FD_CH}'] = 0)")
        print("-" * 50)

    # -------------------------------------------------------------- lazily computed attributes
    def _side(self, key):
        """Run stage 1 if `key` is not cached yet, then return it (all side products land together)."""
        if key not in self._data:
            _ = self.ch_params                                              # resolves defaults if never set
            self._run_prep(want_side=True)
        return self._host(key)

    def _compute_rotated_angles(self) -> Dict[str, np.ndarray]:
        self._side(c.AOD_EL_ROT_PARAM_NAME)
        return {k: self._host(k) for k in _ROT_KEYS}

    def _compute_fov(self) -> Dict[str, Any]:
        self._side(c.FOV_MASK_PARAM_NAME)
        return {k: self._host(k) for k in (c.FOV_MASK_PARAM_NAME,) + _FOV_ANGLE_KEYS}

    def _compute_num_paths(self) -> np.ndarray:
        return self._side(c.NUM_PATHS_PARAM_NAME)

    def _compute_los(self) -> np.ndarray:
        return self._side(c.LOS_PARAM_NAME)

    def _compute_power_linear(self) -> np.ndarray:
        return self._side(c.PWR_LINEAR_PARAM_NAME)

    def _compute_power_linear_ant_gain(self) -> np.ndarray:
        return self._side(c.PWR_LINEAR_ANT_GAIN_PARAM_NAME)

    def _compute_array_response_product(self) -> np.ndarray:
        """complex128 [n_ue, M_rx, M_tx, L] product of the two array responses (dataset.py:398-417), from the rotated,
        FoV-filtered angles stage 1 produced on the GPU.  The channel kernels never form this tensor (10 GB at the
        headline shape, 82 GB at config 5 - they generate it per tile); it exists for callers that read the public
        attribute, on the host and for sizes that fit: above `config('array_response_max_bytes')` (default 2 GiB) a
        MemoryError names the alternatives instead of the bare KeyError an unknown attribute would raise."""
        from .geometry import _ant_indices
        p = self.ch_params
        bs, ue = p[c.PARAMSET_ANT_BS], p[c.PARAMSET_ANT_UE]
        m_tx, m_rx = int(np.prod(bs[c.PARAMSET_ANT_SHAPE])), int(np.prod(ue[c.PARAMSET_ANT_SHAPE]))
        n, L = self[c.POWER_PARAM_NAME].shape
        need = 16 * n * m_rx * m_tx * L
        limit = int(config.get("array_response_max_bytes", 2 << 30))
        if need > limit:
            raise MemoryError(
                f"array_response_product would take {need / 1e9:.1f} GB ([{n}, {m_rx}, {m_tx}, {L}] complex128); the GPU path never "
                f"materialises it - use compute_channels() / dataset.channel (or compute_beam_power), read it for a "
                f"dataset.subset(idxs) of users, or raise config('array_response_max_bytes')")
        aod_el, aod_az = self[c.AOD_EL_FOV_PARAM_NAME], self[c.AOD_AZ_FOV_PARAM_NAME]
        aoa_el, aoa_az = self[c.AOA_EL_FOV_PARAM_NAME], self[c.AOA_AZ_FOV_PARAM_NAME]

        def resp(ant, theta, phi):                                           # geometry.py:38-102
            idx = _ant_indices(ant[c.PARAMSET_ANT_SHAPE]).astype(np.float64)
            kd = 2 * np.pi * float(ant[c.PARAMSET_ANT_SPACING])
            out = np.zeros((theta.shape[0], idx.shape[0], theta.shape[1]), dtype=np.complex128)
            ok = ~np.isnan(theta)
            t, f = theta[ok], phi[ok]
            gamma = 1j * kd * np.stack([np.sin(t) * np.cos(f), np.sin(t) * np.sin(f), np.cos(t)], axis=0)   # [3, n_valid]
            ub, ul = np.nonzero(ok)
            out[ub, :, ul] = np.exp(idx @ gamma).T
            return out

        a_tx, a_rx = resp(bs, aod_el, aod_az), resp(ue, aoa_el, aoa_az)
        return a_rx[:, :, None, :] * a_tx[:, None, :, :]

    def _compute_n_ue(self) -> int:
        return self.rx_pos.shape[0]                                          # dataset.py:657-659

    def _compute_distances(self) -> np.ndarray:
        return np.linalg.norm(self.rx_pos - self.tx_pos, axis=1)             # dataset.py:661-663

    def _compute_inter_int(self) -> np.ndarray:
        v = np.array(self._inter_host(), copy=True)                          # dataset.py:629-637
        v[np.isnan(v)] = -1
        return v.astype(int)

    def _inter_host(self) -> np.ndarray:
        inter = self.inter
        if hasattr(inter, "detach"):                                         # device-resident rays (load(device=...))
            inter = inter.detach().cpu().numpy()
        return np.asarray(inter)

    def _compute_num_interactions(self) -> np.ndarray:
        """Bounces per path = decimal digits of the interaction code; 0 for LoS (code 0), NaN where there is no path
        (dataset.py:621-627)."""
        inter = self._inter_host()
        n = np.zeros_like(inter)
        n[np.isnan(inter)] = np.nan
        pos = inter > 0
        n[pos] = np.floor(np.log10(inter[pos])) + 1
        return n

    _INTER_LETTERS = {"0": "", "1": "R", "2": "D", "3": "S", "4": "T"}

    def _compute_inter_str(self) -> np.ndarray:
        """Interaction codes as letter strings: 1 reflection 'R', 2 diffraction 'D', 3 scattering 'S', 4 transmission
        'T'; LoS (0) is '', no path is 'n' (dataset.py:639-655: the code's float text without its '.0', digit by digit)."""
        inter = self._inter_host()
        table = str.maketrans(self._INTER_LETTERS)
        codes, inverse = np.unique(inter.astype(str), return_inverse=True)      # few distinct codes: translate each once
        words = np.array(["n" if s == "nan" else s[:-2].translate(table) for s in codes.tolist()])
        return words[inverse].reshape(inter.shape)

    # -------------------------------------------------------------- user grid and index helpers (dataset.py:702-795)
    def _compute_grid_info(self) -> Dict[str, np.ndarray]:
        """Distinct x / y receiver coordinates -> grid_size [nx, ny] and the mean spacing along each axis."""
        pos = np.asarray(self.rx_pos)
        xs, ys = np.unique(pos[:, 0]), np.unique(pos[:, 1])
        return {"grid_size": np.array([len(xs), len(ys)]),
                "grid_spacing": np.array([np.mean(np.diff(xs)), np.mean(np.diff(ys))])}

    def _is_valid_grid(self) -> bool:
        return np.prod(self.grid_size) == self.n_ue

    def subset(self, idxs) -> "Dataset":
        """New Dataset of the selected users (dataset.py:739-773): shared objects (scene, materials, load / ray-tracing
        parameters) by reference, every public array whose first axis is the user axis indexed by `idxs`, everything
        else as is; private (cached per-path) entries are dropped and recomputed on demand.  Device-resident rays stay
        on the device."""
        n_ue = self.n_ue
        init = {k: self._host(k) for k in SHARED_PARAMS if k in self._data}
        init[c.N_UE_PARAM_NAME] = len(idxs)
        out = Dataset(init)
        for key, value in self.to_dict().items():
            if key.startswith("_") or key in init:
                continue
            if isinstance(value, np.ndarray) and value.ndim > 0 and value.shape[0] == n_ue:
                value = value[idxs]
            elif hasattr(value, "detach") and value.dim() > 0 and value.shape[0] == n_ue:
                import torch
                value = value[torch.as_tensor(np.asarray(idxs), device=value.device)]
            out[key] = value
        return out

    def get_active_idxs(self) -> np.ndarray:
        """Users with at least one path (dataset.py:775-781)."""
        return np.where(self.num_paths > 0)[0]

    def get_uniform_idxs(self, steps) -> np.ndarray:
        """Every steps[0]-th column and steps[1]-th row of the user grid (dataset.py:783-795)."""
        from .generator_utils import get_uniform_idxs
        return get_uniform_idxs(self.n_ue, self.grid_size, steps)

    # -------------------------------------------------------------- field of view (dataset.py:423-448)
    def apply_fov(self, bs_fov: np.ndarray = np.array([360, 180]), ue_fov: np.ndarray = np.array([360, 180])) -> None:
        self._clear_cache_fov()
        self.bs_fov = bs_fov
        self.ue_fov = ue_fov

    def _clear_cache_fov(self) -> None:
        """dataset.py:515-535"""
        for k in (c.FOV_MASK_PARAM_NAME, c.NUM_PATHS_PARAM_NAME, c.LOS_PARAM_NAME, c.CHANNEL_PARAM_NAME,
                  c.PWR_LINEAR_ANT_GAIN_PARAM_NAME) + _FOV_ANGLE_KEYS:
            self._data.pop(k, None)

    def _clear_cache_rotated_angles(self) -> None:
        """dataset.py:358-378"""
        for k in _ROT_KEYS + (_UE_ROT_RESOLVED,):
            self._data.pop(k, None)
        self._clear_cache_fov()

    # -------------------------------------------------------------- orientation helpers (dataset.py:274-308)
    @property
    def tx_ori(self) -> np.ndarray:
        return self.ch_params["bs_antenna"]["rotation"] * np.pi / 180

    @property
    def bs_ori(self) -> np.ndarray:
        return self.tx_ori

    @property
    def rx_ori(self) -> np.ndarray:
        return self.ch_params["ue_antenna"]["rotation"] * np.pi / 180

    @property
    def ue_ori(self) -> np.ndarray:
        return self.rx_ori

    _computed_attributes = {
        c.N_UE_PARAM_NAME: "_compute_n_ue",
        c.NUM_PATHS_PARAM_NAME: "_compute_num_paths",
        c.DIST_PARAM_NAME: "_compute_distances",
        c.PATHLOSS_PARAM_NAME: "compute_pathloss",
        c.CHANNEL_PARAM_NAME: "compute_channels",
        c.LOS_PARAM_NAME: "_compute_los",
        c.CH_PARAMS_PARAM_NAME: "set_channel_params",
        c.PWR_LINEAR_PARAM_NAME: "_compute_power_linear",
        c.AOA_AZ_ROT_PARAM_NAME: "_compute_rotated_angles",
        c.AOA_EL_ROT_PARAM_NAME: "_compute_rotated_angles",
        c.AOD_AZ_ROT_PARAM_NAME: "_compute_rotated_angles",
        c.AOD_EL_ROT_PARAM_NAME: "_compute_rotated_angles",
        "fov": "_compute_fov",
        c.FOV_MASK_PARAM_NAME: "_compute_fov",
        c.AOA_AZ_FOV_PARAM_NAME: "_compute_fov",
        c.AOA_EL_FOV_PARAM_NAME: "_compute_fov",
        c.AOD_AZ_FOV_PARAM_NAME: "_compute_fov",
        c.AOD_EL_FOV_PARAM_NAME: "_compute_fov",
        c.PWR_LINEAR_ANT_GAIN_PARAM_NAME: "_compute_power_linear_ant_gain",
        c.NUM_INTERACTIONS_PARAM_NAME: "_compute_num_interactions",
        "grid_size": "_compute_grid_info",
        "grid_spacing": "_compute_grid_info",
        c.INTER_STR_PARAM_NAME: "_compute_inter_str",
        c.INTER_INT_PARAM_NAME: "_compute_inter_int",
        "array_response_product": "_compute_array_response_product",       # dataset.py:849
    }


class MacroDataset:
    """List of Datasets (one per TX / RX-set pair); attribute access and method calls fan out to
    every child, a single child returns its value unwrapped (dataset.py:888-998)."""

    SINGLE_ACCESS_METHODS = {"info"}
    PROPAGATE_METHODS = {name for name, _ in inspect.getmembers(Dataset, predicate=inspect.isfunction)
                         if not name.startswith("__")}

    def __init__(self, datasets=None):
        self.datasets = datasets if datasets is not None else []

    def _get_single(self, key):
        if not self.datasets:
            raise IndexError("MacroDataset is empty")
        return self.datasets[0][key]

    def __getattr__(self, name):
        if name in ("datasets",) or name.startswith("__"):
            raise AttributeError(name)
        if name in self.PROPAGATE_METHODS:
            if name in self.SINGLE_ACCESS_METHODS:
                return lambda *a, **k: getattr(self.datasets[0], name)(*a, **k)

            def fan_out(*a, **k):
                res = [getattr(d, name)(*a, **k) for d in self.datasets]
                return res[0] if len(res) == 1 else res
            return fan_out
        if name in SHARED_PARAMS:
            return self._get_single(name)
        res = [getattr(d, name) for d in self.datasets]
        return res[0] if len(res) == 1 else res

    def __getitem__(self, idx):
        if isinstance(idx, (int, slice)):
            return self.datasets[idx]
        if idx in SHARED_PARAMS:
            return self._get_single(idx)
        res = [d[idx] for d in self.datasets]
        return res[0] if len(res) == 1 else res

    def __setitem__(self, key, value):
        for d in self.datasets:
            d[key] = value

    def __len__(self):
        return len(self.datasets)

    def append(self, dataset):
        self.datasets.append(dataset)
