"""The reference-side binding, as working code.

``install(dm)`` patches an imported DeepMIMO package (the reference, v4.0.0a3) so that
``Dataset.compute_channels`` runs on the MI355X through the C-ABI whenever the reference's own - declared
but never read - switch is on::

    import deepmimo as dm
    import deepmimo_amd.reference_patch as gpu
    gpu.install(dm)
    dm.config('use_gpu', True)            # deepmimo/config.py:58
    H = dataset.compute_channels(params)   # same call, same return type, same cached attributes

With ``use_gpu`` off the reference's NumPy path runs untouched.  The patched method keeps the reference's
prologue (``set_channel_params`` + ``np.random.seed(1001)``, dataset.py:245-250), feeds the Dataset's own
float32 matrices to ``ChannelEngine`` and stores the same keys the reference caches: ``channel``, the rotated /
FoV-filtered angles, ``_fov_mask``, ``power_linear``, ``_power_linear_ant_gain``, ``num_paths``, ``los``.
"""
from __future__ import annotations

import numpy as np

from . import consts as c

_ROT = (c.AOD_EL_ROT_PARAM_NAME, c.AOD_AZ_ROT_PARAM_NAME, c.AOA_EL_ROT_PARAM_NAME, c.AOA_AZ_ROT_PARAM_NAME)
_FOV = (c.AOD_EL_FOV_PARAM_NAME, c.AOD_AZ_FOV_PARAM_NAME, c.AOA_EL_FOV_PARAM_NAME, c.AOA_AZ_FOV_PARAM_NAME)


def _gpu_compute_channels(ds, params, device_index: int):
    from .engine import ChannelEngine
    eng = ChannelEngine(device_index)
    n_ue = ds.n_ue
    rot = np.asarray(ds.ch_params.ue_antenna[c.PARAMSET_ANT_ROTATION])
    rot_pu = None
    if not (rot.ndim == 1 and rot.shape[0] == 3):
        if c.AOA_AZ_ROT_PARAM_NAME in ds.keys() and "_ue_rotation_resolved" in ds.keys():
            rot_pu = ds["_ue_rotation_resolved"]           # cached draw (rotated angles were resolved earlier)
        else:
            rot_pu = np.random.uniform(rot[:, 0], rot[:, 1], (n_ue, 3)) if rot.shape == (3, 2) else rot
            ds["_ue_rotation_resolved"] = np.ascontiguousarray(rot_pu, dtype=np.float64)
    rays = eng.upload_rays({k: ds[k] for k in c.RAY_FIELDS})
    rt = ds.get(c.RT_PARAMS_PARAM_NAME)
    fc = float(rt[c.RT_PARAM_FREQUENCY]) if rt is not None and c.RT_PARAM_FREQUENCY in rt else 0.0
    prep = eng.prepare(rays, ds.ch_params, bs_fov=ds.get("bs_fov"), ue_fov=ds.get("ue_fov"),
                       ue_rotation_per_user=rot_pu, carrier_freq=fc, want_side=True)
    chan = eng.channels(prep).cpu().numpy()
    side = {k: (None if v is None else v.cpu().numpy()) for k, v in prep.side.items() if k != "max_delay_key"}
    mask = None if side["fov_mask"] is None else side["fov_mask"].astype(bool)
    rot_arrays = dict(zip(_ROT, (side["aod_el_rot"], side["aod_az_rot"], side["aoa_el_rot"], side["aoa_az_rot"])))
    for k, v in rot_arrays.items():
        ds[k] = v
    ds[c.FOV_MASK_PARAM_NAME] = mask
    for kr, kf in zip(_ROT, _FOV):
        ds[kf] = rot_arrays[kr] if mask is None else np.where(mask, rot_arrays[kr], np.nan)
    ds[c.PWR_LINEAR_PARAM_NAME] = side["power_linear"]
    iso = all(ds.ch_params[s][c.PARAMSET_ANT_RAD_PAT] == "isotropic" for s in (c.PARAMSET_ANT_BS, c.PARAMSET_ANT_UE))
    g = side["power_linear_ant_gain"]
    ds[c.PWR_LINEAR_ANT_GAIN_PARAM_NAME] = g.astype(np.float32) if iso else g
    ds[c.NUM_PATHS_PARAM_NAME] = side["num_paths"].astype(np.int64)
    ds[c.LOS_PARAM_NAME] = side["los"].astype(np.int64)
    return chan


def install(dm) -> None:
    """Patch ``dm.Dataset.compute_channels`` (deepmimo/generator/dataset.py:224-268) in place.  Idempotent."""
    Dataset = dm.Dataset
    if getattr(Dataset.compute_channels, "_mi355x", False):
        return
    original = Dataset.compute_channels
    ChannelGenParameters = dm.ChannelGenParameters

    def compute_channels(self, params=None):
        if not dm.config.get("use_gpu"):
            return original(self, params)
        if params is None:
            params = ChannelGenParameters() if self.ch_params is None else self.ch_params
        self.set_channel_params(params)
        np.random.seed(1001)                                              # dataset.py:250
        channel = _gpu_compute_channels(self, params, int(dm.config.get("gpu_device_id") or 0))
        self[c.CHANNEL_PARAM_NAME] = channel
        return channel

    compute_channels._mi355x = True
    compute_channels.__doc__ = original.__doc__
    Dataset.compute_channels = compute_channels
