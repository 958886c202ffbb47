"""Golden for the cache / invalidation plumbing of Dataset (deepmimo/generator/dataset.py:144-222, 358-378,
515-535): a SEQUENCE of API calls run on the real reference, recording what each step returned.  The GPU test
replays the same sequence on deepmimo_amd.Dataset.  Build container only.

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 /root/repo/oracle/gen_sequence_golden.py
"""
import io
import os
import sys
from contextlib import redirect_stdout, redirect_stderr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle_np import synth_rays  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "seq_cache_plumbing.npz")


def run_sequence(dm, rays):
    """The sequence itself is shared with tests/test_gpu_parity.py::test_cache_plumbing_sequence (kept in sync by hand:
    it is short).  Returns a dict of step -> arrays."""
    out = {}
    ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array([4, 2])
    p.ue_antenna.shape = np.array([2, 1])
    p.ue_antenna.rotation = np.array([[0, 40], [0, 25], [-60, 60]])        # random range per user
    p.ofdm.subcarriers = 64
    p.ofdm.selected_subcarriers = np.array([0, 5, 9])
    # step 1: parameters set, FoV applied, a lazy attribute touched BEFORE compute_channels: the random rotation is
    # drawn now (with whatever the global RNG holds) and stays cached through compute_channels' reseed
    ds.set_channel_params(p)
    ds.apply_fov(bs_fov=np.array([200, 150]), ue_fov=np.array([150, 120]))
    np.random.seed(7)
    out["s1_num_paths"] = np.asarray(ds.num_paths).copy()
    out["s1_los"] = np.asarray(ds.los).copy()
    out["s1_channel"] = ds.compute_channels(p).copy()
    out["s1_mask"] = ds["_fov_mask"].copy()
    # step 2: BS rotation changes -> rotated-angle cache dropped -> new draw after seed(1001)
    p.bs_antenna.rotation = np.array([10, -20, 45])
    out["s2_channel"] = ds.compute_channels(p).copy()
    out["s2_num_paths"] = np.asarray(ds.num_paths).copy()
    out["s2_mask"] = ds["_fov_mask"].copy()
    # step 3: same parameters again -> cached rotated angles reused (no new draw)
    out["s3_channel"] = ds.compute_channels(p).copy()
    # step 4: FoV widened to the full sphere -> mask None, counts change, channel recomputed lazily via .channel
    ds.apply_fov()
    out["s4_mask_is_none"] = np.array(ds["_fov_mask"] is None)
    out["s4_num_paths"] = np.asarray(ds.num_paths).copy()
    out["s4_channel"] = np.asarray(ds.channel).copy()
    # step 5: constant UE rotation, time domain
    p.ue_antenna.rotation = np.array([5, 10, 15])
    p.freq_domain = 0
    p.num_paths = 4
    out["s5_channel"] = ds.compute_channels(p).copy()
    out["s5_los"] = np.asarray(ds.los).copy()
    return out


def main():
    import deepmimo as dm
    rays = synth_rays(30, 8, seed=314)
    with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
        out = run_sequence(dm, rays)
    save = {f"ray_{k}": v for k, v in rays.items()}
    save.update({f"ref_{k}": v for k, v in out.items()})
    np.savez_compressed(OUT, **save)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
