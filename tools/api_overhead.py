#!/usr/bin/env python3
"""Wall time of the public API (Dataset.compute_channels with channel_output='torch') next to the two C-ABI
calls it wraps, headline shape: shows what the host side (ray upload, side products to NumPy, cache plumbing)
adds on top of the kernels.

    python tools/api_overhead.py [--users 100000] [--rounds 5]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import deepmimo_amd as dm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=100000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--workload", default="c3_headline")
    args = ap.parse_args()
    w = dict(bench.WORKLOADS[args.workload])
    w["n_ue"] = args.users
    from oracle import oracle_np as onp        # synthetic host rays only (tool, not product)
    rays = onp.synth_rays(args.users, w["L"], seed=5, all_valid=True)
    p = bench.make_params(w)
    dm.config("channel_output", "torch")
    dev = torch.device("cuda", 0)
    # (1) the floor: the two C-ABI calls on resident rays (what bench.py times), wall clock incl. launch + final sync
    from deepmimo_amd.engine import ChannelEngine
    eng = ChannelEngine(0)
    drays = {k: torch.from_numpy(v).to(dev) for k, v in rays.items()}
    dr = eng.upload_rays(drays)
    prep = eng.prepare(dr, p, want_side=False)
    out = torch.empty(eng.channel_shape(prep), dtype=torch.complex64, device=dev)
    eng.relaunch(prep, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.rounds):
        t0 = time.perf_counter()
        eng.relaunch(prep, out)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    floor = float(np.median(ts))
    print(f"two C-ABI calls (dmx_path_prep + dmx_channels_fd), resident rays, no side products: median {floor:.2f} ms")
    del out, prep
    # (2) the public API on DEVICE-resident rays (what dm.load(..., device='cuda') hands over)
    for label, touch in (("Dataset.compute_channels, device rays", False), ("... + los / num_paths read", True)):
        ts = []
        for _ in range(args.rounds + 1):
            ds = dm.Dataset(dict(drays))
            ds["rx_pos"] = np.zeros((args.users, 3), np.float32)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            H = ds.compute_channels(p)
            if touch:
                _ = ds.los, ds.num_paths
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            del H, ds
        ts = ts[1:]                                                   # first call: allocator warm-up
        print(f"{label}: median {np.median(ts):.2f} ms  min {np.min(ts):.2f} ms  = floor + {np.median(ts) - floor:.2f} ms")
    for label, touch in (("compute_channels only", False), ("compute_channels + los/num_paths", True)):
        ts = []
        for _ in range(args.rounds):
            ds = dm.Dataset(dict(rays))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            H = ds.compute_channels(p)
            if touch:
                _ = ds.los, ds.num_paths
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            del H, ds
        print(f"{label}: median {np.median(ts):.1f} ms  min {np.min(ts):.1f} ms  ({args.users} users, host rays in NumPy)")


if __name__ == "__main__":
    main()
