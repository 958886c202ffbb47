"""Generate tests/golden/*.npz by running the REAL reference in the build container.

Run from the repo root (the reference is read-only at /root/reference and is NOT copied):

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python3 /root/repo/oracle/gen_golden.py

Each fixture holds the synthetic inputs (float32 ray matrices), the channel parameters as
JSON, and what the reference returned for them (channel tensor or a sub-sampled view of it
plus float64 checksums, LoS, path counts, FoV mask, rotated angles, powers).  Fixtures are
data only.  Cases follow SURVEY.md section 8(c) (G1..G11) and mirror the parameter grid of
the reference's own test/test_v3_correspondence.py:21-34.

The GPU box never runs this file (it has no /root/reference).
"""
from __future__ import annotations

import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle_np import synth_rays  # noqa: E402  (input generator only)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _case(name, n_ue, n_paths, seed, bs_shape, ue_shape, n_sc, sel, **kw):
    c = dict(name=name, n_ue=n_ue, n_paths=n_paths, seed=seed, bs_shape=bs_shape, ue_shape=ue_shape,
             subcarriers=n_sc, selected=sel, bs_rot=[0, 0, 0], ue_rot=[0, 0, 0], bs_fov=None, ue_fov=None,
             bs_pattern="isotropic", ue_pattern="isotropic", freq_domain=1, rx_filter=0, num_paths=25,
             bandwidth=10e6, bs_spacing=0.5, ue_spacing=0.5, max_delay=2e-6, all_valid=False,
             ue_rot_mode="const", doppler=False, subsample=None, nan_holes=False)
    c.update(kw)
    return c


CASES = [
    _case("g01_plumbing", 64, 5, 101, [4, 1], [1, 1], 64, list(range(64))),
    _case("g02_panel_order", 16, 10, 102, [8, 4], [2, 2], 256, list(range(0, 256, 4))),
    _case("g03_rot_fov", 48, 10, 103, [4, 2], [2, 1], 64, [0, 3, 6], bs_rot=[30, 40, 30],
          ue_rot_mode="per_user", bs_fov=[140, 120], ue_fov=[90, 80]),
    _case("g03b_rot_fov_bs_only", 48, 10, 113, [4, 2], [1, 1], 64, [0, 3, 6], bs_rot=[30, 40, 30],
          bs_fov=[140, 120]),
    _case("g04_time_domain", 32, 10, 104, [4, 2], [2, 2], 64, [0], freq_domain=0, bs_rot=[10, 20, 30],
          bs_fov=[140, 120], ue_fov=[360, 180]),
    _case("g05_rx_filter", 16, 5, 105, [4, 1], [2, 1], 64, list(range(0, 64, 2)), rx_filter=1),
    _case("g06_dipole", 32, 10, 106, [4, 2], [2, 2], 64, list(range(64)), bs_pattern="halfwave-dipole",
          ue_pattern="halfwave-dipole", bs_rot=[5, 10, 15], bs_fov=[180, 120], ue_fov=[360, 180]),
    _case("g07_delay_clip", 32, 10, 107, [4, 2], [1, 1], 64, list(range(64)), max_delay=12e-6),
    _case("g08_num_paths_nan", 40, 12, 108, [4, 2], [2, 1], 64, list(range(0, 64, 3)), num_paths=7,
          nan_holes=True),
    _case("g09_random_ue_rot", 24, 8, 109, [4, 2], [2, 2], 64, [0, 3, 6], ue_rot_mode="random",
          ue_rot=[[0, 30], [0, 20], [-45, 45]], ue_fov=[120, 90]),
    _case("g10_doppler_v3", 16, 8, 110, [4, 2], [2, 1], 64, list(range(64)), doppler=True, all_valid=False),
    _case("g13_doppler_lpf_v3", 8, 6, 114, [4, 1], [2, 1], 32, list(range(0, 32, 3)), doppler=True, rx_filter=1,
          max_delay=2.5e-6),
    _case("g11_headline_phase", 4, 25, 111, [8, 8], [2, 2], 512, list(range(512)), all_valid=True,
          max_delay=0.98 * 512 / 10e6, subsample=dict(tx=list(range(0, 64, 7)), k=list(range(0, 512, 16)) + [511])),
    _case("g12_ula64_rot", 8, 25, 112, [64, 1], [1, 4], 128, list(range(0, 128, 5)), bs_rot=[0, 0, -135],
          ue_rot=[20, 0, 90], max_delay=10e-6),
]


def build_inputs(c):
    rays = synth_rays(c["n_ue"], c["n_paths"], seed=c["seed"], all_valid=c["all_valid"],
                      max_delay=c["max_delay"], with_doppler=c["doppler"])
    if c["nan_holes"]:
        rng = np.random.default_rng(c["seed"] + 7)
        # NaN in the middle of rows (every field of that path) + some all-NaN users
        hole = rng.uniform(size=rays["power"].shape) < 0.15
        for k in ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter"):
            rays[k][hole] = np.nan
        for k in ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter"):
            rays[k][:3] = np.nan
    ue_rot = np.asarray(c["ue_rot"], dtype=np.float64)
    if c["ue_rot_mode"] == "per_user":
        ue_rot = np.random.default_rng(42).uniform(0, 45, (c["n_ue"], 3))   # test_v3_correspondence.py:74-76
    return rays, ue_rot


def run_v4(c, rays, ue_rot):
    import deepmimo as dm
    ds = dm.Dataset({k: v.copy() for k, v in rays.items() if not k.startswith("doppler")})
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array(c["bs_shape"])
    p.ue_antenna.shape = np.array(c["ue_shape"])
    p.bs_antenna.spacing = c["bs_spacing"]
    p.ue_antenna.spacing = c["ue_spacing"]
    p.bs_antenna.rotation = np.array(c["bs_rot"])
    p.ue_antenna.rotation = np.array(ue_rot)
    p.bs_antenna.radiation_pattern = c["bs_pattern"]
    p.ue_antenna.radiation_pattern = c["ue_pattern"]
    p.num_paths = c["num_paths"]
    p.freq_domain = c["freq_domain"]
    p.ofdm.subcarriers = c["subcarriers"]
    p.ofdm.selected_subcarriers = np.array(c["selected"])
    p.ofdm.bandwidth = c["bandwidth"]
    p.ofdm.rx_filter = c["rx_filter"]
    if c["bs_fov"] is not None or c["ue_fov"] is not None:
        kw = {}
        if c["bs_fov"] is not None:
            kw["bs_fov"] = np.array(c["bs_fov"])
        if c["ue_fov"] is not None:
            kw["ue_fov"] = np.array(c["ue_fov"])
        ds.apply_fov(**kw)
    buf = io.StringIO()
    with redirect_stdout(buf):
        H = ds.compute_channels(p)
    out = dict(channel=H, los=np.asarray(ds.los), num_paths=np.asarray(ds.num_paths),
               power_linear=np.asarray(ds.power_linear),
               power_linear_ant_gain=np.asarray(ds["_power_linear_ant_gain"]),
               aod_el_rot=ds["_aod_el_rot"], aod_az_rot=ds["_aod_az_rot"],
               aoa_el_rot=ds["_aoa_el_rot"], aoa_az_rot=ds["_aoa_az_rot"],
               warned=np.array("exceed OFDM symbol duration" in buf.getvalue()))
    m = ds["_fov_mask"]
    if m is not None:
        out["fov_mask"] = m
    return out


def run_v3(c, rays, ue_rot, enable_doppler):
    """Legacy generator = the only Python definition of the Doppler term
    (deepmimo_v3/generator/python/construct_deepmimo.py:267-280)."""
    from deepmimo_v3.generator.python.construct_deepmimo import generate_MIMO_channel
    n = c["n_ue"]
    raydata = []
    for i in range(n):
        v = ~np.isnan(rays["power"][i])
        raydata.append({
            "num_paths": int(v.sum()),
            "DoD_phi": rays["aod_az"][i, v].copy(), "DoD_theta": rays["aod_el"][i, v].copy(),
            "DoA_phi": rays["aoa_az"][i, v].copy(), "DoA_theta": rays["aoa_el"][i, v].copy(),
            "phase": rays["phase"][i, v].copy(), "ToA": rays["delay"][i, v].copy(),
            "power": (10 ** (rays["power"][i, v] / 10)).copy(), "LoS": (rays["inter"][i, v] == 0).astype(int),
            "Doppler_vel": rays["doppler_vel"][i, v].copy(), "Doppler_acc": rays["doppler_acc"][i, v].copy(),
        })
    params = {"ofdm": {"subcarriers": c["subcarriers"], "selected_subcarriers": np.array(c["selected"]),
                       "bandwidth": c["bandwidth"] / 1e9, "rx_filter": c["rx_filter"]},
              "freq_domain": c["freq_domain"], "num_paths": c["num_paths"], "enable_doppler": int(enable_doppler),
              "scenario_params": {"doppler_available": 1, "carrier_freq": 3.5e9}}
    tx = {"shape": np.array(c["bs_shape"]), "spacing": c["bs_spacing"], "rotation": np.array(c["bs_rot"]),
          "fov": np.array([360, 180]), "radiation_pattern": "isotropic"}
    rx = {"shape": np.array(c["ue_shape"]), "spacing": c["ue_spacing"],
          "rotation": np.tile(np.asarray(ue_rot, dtype=np.float64), (n, 1)) if np.ndim(ue_rot) == 1 else ue_rot,
          "fov": np.array([360, 180]), "radiation_pattern": "isotropic"}
    buf = io.StringIO()
    with redirect_stdout(buf):
        H, _ = generate_MIMO_channel(raydata, params, tx, rx)
    return H


def checksums(H):
    """float64 size-independent fingerprints of a channel tensor (used when H is sub-sampled)."""
    Hd = H.astype(np.complex128)
    w = np.cos(np.arange(Hd.size, dtype=np.float64) * 0.37).reshape(Hd.shape)
    return np.array([np.sum(np.abs(Hd) ** 2), np.sum(Hd.real * w), np.sum(Hd.imag * w)])


def main():
    os.makedirs(OUT, exist_ok=True)
    for c in CASES:
        rays, ue_rot = build_inputs(c)
        out = run_v4(c, rays, ue_rot)
        if c["doppler"]:
            # v3 == v4 with Doppler off (SURVEY finding 4), then the v3 Doppler-on channel is the golden
            h3_off = run_v3(c, rays, ue_rot, False)
            diff = float(np.max(np.abs(h3_off - out["channel"])))
            assert diff < 1e-10, f"v3/v4 correspondence broken: {diff}"
            out["channel_doppler"] = run_v3(c, rays, ue_rot, True)
            out["v3_v4_maxdiff"] = np.array(diff)
        save = {f"ray_{k}": v for k, v in rays.items()}
        save["ue_rot"] = np.asarray(ue_rot, dtype=np.float64)
        save["case_json"] = np.array(json.dumps(c))
        H = out.pop("channel")
        save["channel_checksum"] = checksums(H)
        save["channel_user_peak"] = np.abs(H).reshape(H.shape[0], -1).max(axis=1)
        if c["subsample"]:
            save["channel_sub"] = H[:, :, c["subsample"]["tx"], :][..., c["subsample"]["k"]]
        else:
            save["channel"] = H
        for k, v in out.items():
            save[f"ref_{k}"] = v
        path = os.path.join(OUT, c["name"] + ".npz")
        np.savez_compressed(path, **save)
        print(f"{c['name']:24s} H{H.shape} peak={np.abs(H).max():.3e} -> {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
