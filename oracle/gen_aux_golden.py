"""Goldens for the rows that used to be checked against this repo's own restatements only (VERDICT r1, items a13,
f2-pathloss, f4, stale-cache behaviour), produced by running the REAL reference in the build container:

  aux_steering.npz   dm.steering_vec (deepmimo/generator/geometry.py:322-339) over shapes / angles / spacings
  aux_pathloss.npz   Dataset.compute_pathloss, coherent and incoherent (deepmimo/generator/dataset.py:541-566)
  aux_sionna.npz     (a, tau) samples of the reference's DeepMIMOSionnaAdapter
                     (deepmimo/integrations/sionna_adapter.py:22-200) fed a v3-layout dict built from the reference's
                     own time-domain channels of two basestations
  aux_stale_cache.npz  a call sequence in which ONLY the radiation pattern changes: the reference keeps its cached
                     `_power_linear_ant_gain` (dataset.py:213-220 invalidates on rotation / FoV changes only) and
                     returns the isotropic channel again; the fresh dipole result is recorded beside it

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python3 /root/repo/oracle/gen_aux_golden.py

Fixtures are data only.  The GPU box never runs this file (it has no /root/reference).
"""
from __future__ import annotations

import io
import os
import sys
from contextlib import redirect_stdout, redirect_stderr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle_np import synth_rays  # noqa: E402  (input generator only)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

STEERING_CASES = [  # shape, phi, theta, spacing
    ([8, 1], 0.0, 0.0, 0.5), ([8, 1], 30.0, 0.0, 0.5), ([8, 1], -47.5, 12.0, 0.5), ([4, 4], 20.0, -35.0, 0.5),
    ([8, 8], 60.0, 10.0, 0.5), ([1, 4], 15.0, 80.0, 0.3), ([16, 2], -5.0, 5.0, 0.7), ([3, 5], 123.0, -77.0, 0.25),
    ([1, 1], 40.0, 40.0, 0.5), ([64, 1], 1.0, 0.0, 0.5),
]


def sionna_inputs():
    """Rays of two basestations and the TD parameters (shared with tests/test_gpu_parity.py through the fixture)."""
    rays = [synth_rays(12, 6, seed=801), synth_rays(12, 6, seed=802)]
    return rays, dict(bs_shape=[4, 2], ue_shape=[2, 1], num_paths=5, bs_rot=[0, 10, -20])


def v3_layout(channels, rays_list, num_paths):
    """The v3 dict the reference adapter reads (sionna_adapter.py:72-93, 188-198): per basestation
    {'user': {'channel': [n_ue, M_rx, M_tx, P], 'paths': [{'num_paths', 'ToA'} per user]}}; a user's valid paths are
    its non-NaN entries among the first `num_paths` (the ones the TD channel compacts to the front, channel.py:285-287)."""
    out = []
    for H, rays in zip(channels, rays_list):
        paths = []
        for i in range(H.shape[0]):
            d = rays["delay"][i, :num_paths]
            v = ~np.isnan(rays["power"][i, :num_paths])
            paths.append({"num_paths": int(v.sum()), "ToA": d[v]})
        out.append({"user": {"channel": H, "paths": paths}})
    return out


def main():
    import deepmimo as dm
    from deepmimo.integrations.sionna_adapter import DeepMIMOSionnaAdapter
    os.makedirs(OUT, exist_ok=True)

    # ---- steering vectors
    sv = {}
    for i, (shape, phi, theta, spacing) in enumerate(STEERING_CASES):
        sv[f"v{i}"] = np.asarray(dm.steering_vec(np.array(shape), phi=phi, theta=theta, spacing=spacing))
    np.savez_compressed(os.path.join(OUT, "aux_steering.npz"),
                        cases=np.array([[s[0], s[1], p, t, sp] for s, p, t, sp in STEERING_CASES], dtype=np.float64), **sv)

    # ---- pathloss
    rays = synth_rays(200, 9, seed=515)
    rays["power"][5, :] = np.nan                     # a user without paths
    rays["phase"][5, :] = np.nan
    with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
        ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
        pl_c = np.asarray(ds.compute_pathloss(coherent=True)).copy()
        ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
        pl_i = np.asarray(ds.compute_pathloss(coherent=False)).copy()
    np.savez_compressed(os.path.join(OUT, "aux_pathloss.npz"), ray_power=rays["power"], ray_phase=rays["phase"],
                        ref_coherent=pl_c, ref_incoherent=pl_i)

    # ---- Sionna adapter on the reference's own TD channels
    rays_list, cfg = sionna_inputs()
    chans = []
    with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
        for r in rays_list:
            p = dm.ChannelGenParameters()
            p.bs_antenna.shape = np.array(cfg["bs_shape"])
            p.ue_antenna.shape = np.array(cfg["ue_shape"])
            p.bs_antenna.rotation = np.array(cfg["bs_rot"])
            p.num_paths = cfg["num_paths"]
            p.freq_domain = 0
            chans.append(np.asarray(dm.Dataset({k: v.copy() for k, v in r.items()}).compute_channels(p)).copy())
    bs_idx, ue_idx = np.array([[0, 1]]), np.array([[0, 1, 2], [3, 4, 5], [9, 10, 11]])
    ad = DeepMIMOSionnaAdapter(v3_layout(chans, rays_list, cfg["num_paths"]), bs_idx=bs_idx, ue_idx=ue_idx)
    samples = list(ad())
    ad1 = DeepMIMOSionnaAdapter(v3_layout(chans, rays_list, cfg["num_paths"]))                 # defaults: BS 0, every user
    samples1 = list(ad1())
    save = {"bs_idx": bs_idx, "ue_idx": ue_idx, "n_samples": np.array(len(ad)), "n_samples_default": np.array(len(ad1)),
            "a": np.stack([s[0] for s in samples]), "tau": np.stack([s[1] for s in samples]),
            "a_default": np.stack([s[0] for s in samples1]), "tau_default": np.stack([s[1] for s in samples1]),
            "td_channel_bs0": chans[0], "td_channel_bs1": chans[1]}
    for b, r in enumerate(rays_list):
        save.update({f"bs{b}_ray_{k}": v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "aux_sionna.npz"), **save)

    # ---- only the radiation pattern changes
    rays = synth_rays(24, 7, seed=616)

    def params(pattern):
        p = dm.ChannelGenParameters()
        p.bs_antenna.shape = np.array([4, 2])
        p.ue_antenna.shape = np.array([2, 1])
        p.bs_antenna.rotation = np.array([10, 20, 30])
        p.bs_antenna.radiation_pattern = pattern
        p.ofdm.subcarriers = 64
        p.ofdm.selected_subcarriers = np.arange(0, 64, 4)
        return p

    with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
        ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
        h_iso = np.asarray(ds.compute_channels(params("isotropic"))).copy()
        h_stale = np.asarray(ds.compute_channels(params("halfwave-dipole"))).copy()       # same Dataset: cache kept
        g_stale = np.asarray(ds["_power_linear_ant_gain"]).copy()
        ds.apply_fov()                                                                    # any FoV call drops the cache
        h_after_fov = np.asarray(ds.compute_channels(params("halfwave-dipole"))).copy()
        fresh = dm.Dataset({k: v.copy() for k, v in rays.items()})
        h_fresh = np.asarray(fresh.compute_channels(params("halfwave-dipole"))).copy()
        g_fresh = np.asarray(fresh["_power_linear_ant_gain"]).copy()
    save = {f"ray_{k}": v for k, v in rays.items()}
    save.update(ref_iso=h_iso, ref_dipole_same_dataset=h_stale, ref_dipole_after_apply_fov=h_after_fov,
                ref_dipole_fresh=h_fresh, ref_gain_same_dataset=g_stale, ref_gain_fresh=g_fresh)
    np.savez_compressed(os.path.join(OUT, "aux_stale_cache.npz"), **save)
    print("steering", len(sv), "| pathloss", pl_c.shape, np.isnan(pl_c).sum(), "NaN | sionna", save and len(samples), "samples",
          "| stale == iso:", bool(np.array_equal(h_stale, h_iso)), " fresh == after_fov:", bool(np.array_equal(h_fresh, h_after_fov)),
          " stale == fresh:", bool(np.array_equal(h_stale, h_fresh)))


if __name__ == "__main__":
    main()
