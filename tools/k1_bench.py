#!/usr/bin/env python3
"""Stage-1 (k1_path_prep) time alone: `python tools/k1_bench.py [--users 200000] [--paths 25]` (lean form, as
compute_channels runs it); set DMX_LIB_PATH to time another build of the library on the same box."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deepmimo_amd.engine import ChannelEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=200000)
    ap.add_argument("--paths", type=int, default=25)
    ap.add_argument("--rounds", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    eng = ChannelEngine(0)
    w = dict(bench.WORKLOADS["d8_default_arrays"]); w["n_ue"] = args.users; w["L"] = args.paths
    rays = eng.upload_rays(bench.synth_device_rays(args.users, args.paths, 1234, dev))
    p = bench.make_params(w)
    for side, label in ((False, "lean (no side products)"), (True, "full (angles, powers, masks)")):
        ts = []
        for i in range(args.rounds + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); prep = eng.prepare(rays, p, want_side=side); e1.record(); torch.cuda.synchronize()
            if i >= 3:
                ts.append(e0.elapsed_time(e1))
        print(f"stage 1 {label}: median {np.median(ts) * 1e3:.0f} us  min {min(ts) * 1e3:.0f} us  ({args.users} users x {args.paths} paths)")


if __name__ == "__main__":
    main()
