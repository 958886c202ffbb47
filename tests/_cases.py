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


# ---- seeded random configurations (shared by the GPU sweep and the C-oracle cross-check) --------------------
def _random_config(rng):
    bs = [int(rng.integers(1, 13)), int(rng.integers(1, 9))]
    ue = [int(rng.integers(1, 4)), int(rng.integers(1, 4))]
    L = int(rng.integers(1, 33))
    N = int(rng.choice([16, 48, 64, 100, 256, 512]))
    K = int(rng.integers(1, min(N, 96) + 1))
    sel = np.sort(rng.choice(N, size=K, replace=False)) if rng.uniform() < 0.7 else rng.integers(0, N, K)
    mode = rng.choice(["fd", "fd", "fd", "td", "lpf"])
    cfg = dict(bs=bs, ue=ue, L=L, N=N, sel=sel, mode=mode,
               n_ue=int(rng.integers(1, 90)),
               num_paths=int(rng.integers(1, L + 6)),
               bs_rot=rng.integers(-180, 181, 3) if rng.uniform() < 0.6 else np.zeros(3, int),
               ue_rot_mode=rng.choice(["zero", "const", "per_user"]),
               bs_fov=[int(rng.integers(30, 361)), int(rng.integers(20, 181))] if rng.uniform() < 0.5 else None,
               ue_fov=[int(rng.integers(30, 361)), int(rng.integers(20, 181))] if rng.uniform() < 0.3 else None,
               bs_pat=rng.choice(["isotropic", "halfwave-dipole"], p=[0.7, 0.3]),
               ue_pat=rng.choice(["isotropic", "halfwave-dipole"], p=[0.8, 0.2]),
               spacing=float(rng.choice([0.5, 0.25, 0.7])),
               bandwidth=float(rng.choice([10e6, 20e6, 100e6])),
               max_delay=float(rng.choice([5e-7, 2e-6, 2e-5])),
               holes=rng.uniform() < 0.3)
    if mode == "lpf":
        cfg["N"] = int(rng.choice([16, 48, 64]))
        cfg["sel"] = np.arange(0, cfg["N"], int(rng.integers(1, 4)))
    return cfg


def random_case(seed, lpf_n=None):
    """(config, rays, ue_rotation, oracle params, bs_fov, ue_fov) of sweep seed `seed`.  lpf_n: the same configuration
    forced to rx_filter = 1 at that OFDM size, with a selection drawn from a generator of its own (the recorded draw
    order of the sweep stays what it was): all bins, a stride, the first part, a wrapped run or a random subset."""
    from oracle import oracle_np as onp
    rng = np.random.default_rng(9000 + seed)
    c = _random_config(rng)
    if lpf_n is not None:
        r2 = np.random.default_rng(77000 + seed)
        kind = r2.choice(["all", "stride", "first", "wrap", "random"])
        c["mode"], c["N"] = "lpf", int(lpf_n)
        c["sel"] = {"all": np.arange(lpf_n), "stride": np.arange(0, lpf_n, int(r2.integers(2, 5))),
                    "first": np.arange(int(r2.integers(1, lpf_n + 1))), "wrap": np.arange(lpf_n, 2 * lpf_n),
                    "random": np.sort(r2.choice(lpf_n, size=int(r2.integers(1, min(lpf_n, 200) + 1)), replace=False))}[str(kind)]
        c["n_ue"] = min(c["n_ue"], 24 if lpf_n >= 512 else 60)
        c["max_delay"] = float(r2.choice([0.3, 0.9, 1.2])) * lpf_n / c["bandwidth"]
    n = c["n_ue"]
    rays = onp.synth_rays(n, c["L"], seed=seed, max_delay=c["max_delay"])
    if c["holes"]:
        hole = rng.uniform(size=(n, c["L"])) < 0.2
        for k in onp.RAY_KEYS:
            rays[k][hole] = np.nan
    if c["ue_rot_mode"] == "zero":
        ue_rot = np.zeros(3, int)
    elif c["ue_rot_mode"] == "const":
        ue_rot = rng.integers(-90, 91, 3)
    else:
        ue_rot = rng.uniform(-60, 60, (n, 3))
        if n == 3:                                     # keeps the generator's draw order of the recorded sweep
            ue_rot = rng.uniform(-60, 60, (n, 3))
    fd = c["mode"] != "td"
    op = onp.make_params(
        bs_antenna=dict(shape=c["bs"], spacing=c["spacing"], rotation=np.array(c["bs_rot"]), radiation_pattern=str(c["bs_pat"])),
        ue_antenna=dict(shape=c["ue"], spacing=c["spacing"], rotation=np.array(ue_rot), radiation_pattern=str(c["ue_pat"])),
        num_paths=c["num_paths"], freq_domain=int(fd),
        ofdm=dict(subcarriers=c["N"], selected_subcarriers=np.asarray(c["sel"]), bandwidth=c["bandwidth"],
                  rx_filter=int(c["mode"] == "lpf")))
    bs_fov = None if c["bs_fov"] is None else np.array(c["bs_fov"])
    ue_fov = None if c["ue_fov"] is None else np.array(c["ue_fov"])
    return c, rays, ue_rot, op, bs_fov, ue_fov
