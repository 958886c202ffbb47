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
