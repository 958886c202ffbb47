"""Shared helpers for the parity tests: load golden fixtures (tests/golden/*.npz, produced by
oracle/gen_golden.py from the real reference) and turn a fixture's case description into the
oracle's parameter dict / the product's ChannelGenParameters."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RAY_KEYS = ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter")

# |H_test - H_ref| <= TOL_REL * max|H_ref[user]| + TOL_ABS   (BASELINE.json north_star)
TOL_REL = 5e-5
TOL_ABS = 1e-12


def golden_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "g[0-9]*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    case = json.loads(str(z["case_json"]))
    rays = {k[4:]: z[k] for k in z.files if k.startswith("ray_")}
    ref = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    for k in ("channel", "channel_sub", "channel_checksum", "channel_user_peak"):
        if k in z.files:
            ref[k] = z[k]
    return case, rays, z["ue_rot"], ref


def oracle_params(case, ue_rot):
    from oracle import oracle_np as onp
    return onp.make_params(
        bs_antenna=dict(shape=case["bs_shape"], spacing=case["bs_spacing"], rotation=np.array(case["bs_rot"]),
                        radiation_pattern=case["bs_pattern"]),
        ue_antenna=dict(shape=case["ue_shape"], spacing=case["ue_spacing"], rotation=np.array(ue_rot),
                        radiation_pattern=case["ue_pattern"]),
        num_paths=case["num_paths"], freq_domain=case["freq_domain"],
        ofdm=dict(subcarriers=case["subcarriers"], selected_subcarriers=np.array(case["selected"]),
                  bandwidth=case["bandwidth"], rx_filter=case["rx_filter"]))


def fov_args(case):
    bs = None if case["bs_fov"] is None else np.array(case["bs_fov"])
    ue = None if case["ue_fov"] is None else np.array(case["ue_fov"])
    return bs, ue


def channel_err(H, Href):
    """max over users of max|dH| / max|Href[user]| (users whose reference is all-zero must be exactly zero)."""
    H = np.asarray(H).astype(np.complex128)
    Href = np.asarray(Href).astype(np.complex128)
    n = Href.shape[0]
    d = np.abs(H - Href).reshape(n, -1).max(axis=1) if n else np.zeros(0)
    peak = np.abs(Href).reshape(n, -1).max(axis=1) if n else np.zeros(0)
    return d, peak


def assert_channel_close(H, Href, tol_rel=TOL_REL, tol_abs=TOL_ABS, what=""):
    assert H.shape == Href.shape, f"{what}: shape {H.shape} vs {Href.shape}"
    fin = np.isfinite(Href.astype(np.complex128))
    assert np.array_equal(np.isfinite(np.asarray(H).astype(np.complex128)), fin), f"{what}: NaN pattern differs"
    d, peak = channel_err(np.where(fin, H, 0), np.where(fin, Href, 0))
    bad = d > tol_rel * peak + tol_abs
    assert not bad.any(), (f"{what}: {bad.sum()} users out of tolerance; worst rel err "
                           f"{np.max(d / np.maximum(peak, 1e-300)):.3e} (tol {tol_rel})")
    return float(np.max(d / np.maximum(peak, 1e-300))) if d.size else 0.0
