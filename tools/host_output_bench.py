#!/usr/bin/env python3
"""End-to-end time of Dataset.compute_channels with its default NumPy return value (the reference's return type):
stage 1 + stage 2 + the way to host memory, against the same call with channel_output='torch' (tensor stays in HBM)
and against a plain tensor.cpu().numpy() of that tensor.

    python tools/host_output_bench.py [--workload d8_default_arrays] [--users 200000]
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
    ap.add_argument("--workload", default="d8_default_arrays")
    ap.add_argument("--users", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    w = dict(bench.WORKLOADS[args.workload])
    if args.users:
        w["n_ue"] = args.users
    dev = torch.device("cuda", 0)
    rays = bench.synth_device_rays(w["n_ue"], w["L"], 1234, dev)
    p = bench.make_params(w)
    ds = dm.Dataset({k: v for k, v in rays.items()})
    ds["rx_pos"] = np.zeros((w["n_ue"], 3), np.float32)
    ds["tx_pos"] = np.zeros((1, 3), np.float32)
    dm.config("channel_output", "torch")
    H = ds.compute_channels(p); torch.cuda.synchronize()
    gb = H.numel() * 8 / 1e9
    ts = []
    for _ in range(args.rounds):
        t0 = time.perf_counter(); H = ds.compute_channels(p); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{args.workload}, {w['n_ue']} users, tensor {gb:.2f} GB")
    print(f"  channel_output='torch' (stays in HBM):    {min(ts) * 1e3:9.2f} ms")
    ts = []
    for _ in range(args.rounds):
        t0 = time.perf_counter(); a = H.cpu().numpy(); ts.append(time.perf_counter() - t0)
    print(f"  + tensor.cpu().numpy() (round-1 path):     {min(ts) * 1e3:9.2f} ms = {gb / min(ts):.1f} GB/s")
    ref = a
    del H
    ds["channel"] = None
    dm.config("channel_output", "numpy")
    ts = []
    for _ in range(args.rounds + 1):
        t0 = time.perf_counter(); a = ds.compute_channels(p); ts.append(time.perf_counter() - t0)
    print(f"  channel_output='numpy' (pipelined):        {min(ts) * 1e3:9.2f} ms = {gb / min(ts):.1f} GB/s end to end (calls: {', '.join(f'{t * 1e3:.0f}' for t in ts)} ms)")
    print("  bit-identical to the resident tensor:", bool(np.array_equal(a.view(np.uint32), ref.view(np.uint32))))


if __name__ == "__main__":
    main()
