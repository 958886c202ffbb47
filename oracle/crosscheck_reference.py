"""One-off fidelity sweep (build container only): the NumPy oracle against the imported reference on the
random configurations of tests/test_gpu_random_sweep.py (panel shapes, paths, subcarriers, rotations, FoV,
patterns, FD/TD/LPF, num_paths slicing, NaN holes).  Complements the committed goldens.

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python3 /root/repo/oracle/crosscheck_reference.py 300
"""
import io
import os
import sys
from contextlib import redirect_stdout, redirect_stderr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as onp  # noqa: E402
from tests._cases import random_case  # noqa: E402


def main(n_cfg=None):
    import deepmimo as dm
    if n_cfg is None:
        n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    worst = 0.0
    for seed in range(n_cfg):
        c, rays, ue_rot, op, bs_fov, ue_fov = random_case(seed)       # the configurations of tests/test_gpu_random_sweep.py
        fd = c["mode"] != "td"
        p = dm.ChannelGenParameters()
        p.bs_antenna.shape, p.ue_antenna.shape = np.array(c["bs"]), np.array(c["ue"])
        p.bs_antenna.spacing = p.ue_antenna.spacing = c["spacing"]
        p.bs_antenna.rotation, p.ue_antenna.rotation = np.array(c["bs_rot"]), np.array(ue_rot)
        p.bs_antenna.radiation_pattern, p.ue_antenna.radiation_pattern = str(c["bs_pat"]), str(c["ue_pat"])
        p.num_paths, p.freq_domain = c["num_paths"], int(fd)
        p.ofdm.subcarriers, p.ofdm.selected_subcarriers = c["N"], np.asarray(c["sel"])
        p.ofdm.bandwidth, p.ofdm.rx_filter = c["bandwidth"], int(c["mode"] == "lpf")
        ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
        if bs_fov is not None or ue_fov is not None:
            kw = {}
            if bs_fov is not None:
                kw["bs_fov"] = bs_fov
            if ue_fov is not None:
                kw["ue_fov"] = ue_fov
            ds.apply_fov(**kw)
        with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
            Href = ds.compute_channels(p)
            los_ref, np_ref, mask_ref = np.asarray(ds.los), np.asarray(ds.num_paths), ds["_fov_mask"]
        res = onp.compute_channels(rays, op, bs_fov=bs_fov, ue_fov=ue_fov)
        H = res["channel"]
        fin = np.isfinite(Href)
        assert np.array_equal(np.isfinite(H), fin), f"seed {seed}: NaN pattern differs"
        d = float(np.max(np.abs(np.where(fin, H - Href, 0)))) if H.size else 0.0
        peak = float(np.max(np.abs(np.where(fin, Href, 0)))) if H.size else 1.0
        worst = max(worst, d / max(peak, 1e-300))
        assert d <= 1e-7 * peak + 1e-12, f"seed {seed}: channel differs by {d:.3e} (peak {peak:.3e}) cfg {c}"
        assert np.array_equal(res["los"], los_ref), f"seed {seed}: los"
        assert np.array_equal(res["num_paths"], np_ref), f"seed {seed}: num_paths"
        assert (mask_ref is None and res["_fov_mask"] is None) or np.array_equal(res["_fov_mask"], mask_ref), f"seed {seed}: mask"
    print(f"oracle == reference on {n_cfg} random configurations; worst |dH|/peak = {worst:.2e}")
    return worst


if __name__ == "__main__":
    main()
