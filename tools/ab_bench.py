#!/usr/bin/env python3
"""Interleaved A/B timing of stage-2 kernel variants in ONE process (guide rule 24): N rounds over
the listed variants on the same resident inputs; prints median / min kernel ms per variant.

    python tools/ab_bench.py --variants 2 3 1 --rounds 7 [--workload c3_headline] [--users N]
"""
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
    ap.add_argument("--variants", type=int, nargs="+", default=[2, 3])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--workload", default="c3_headline")
    ap.add_argument("--users", type=int, default=0)
    ap.add_argument("--random-valid", action="store_true")
    ap.add_argument("--shape", type=int, nargs=6, metavar=("BS0", "BS1", "UE0", "UE1", "L", "N"), default=None,
                    help="custom shape instead of --workload (all N subcarriers selected)")
    ap.add_argument("--beams", type=int, default=0, help="also time the fused beam-space kernel with this many beams")
    args = ap.parse_args()
    w = dict(bench.WORKLOADS[args.workload])
    if args.shape:
        b0, b1, u0, u1, L, N = args.shape
        w = dict(n_ue=w["n_ue"], bs=[b0, b1], ue=[u0, u1], L=L, N=N)
    if args.users:
        w["n_ue"] = args.users
    dev = torch.device("cuda", 0)
    eng = ChannelEngine(0)
    params = bench.make_params(w)
    rays = eng.upload_rays(bench.synth_device_rays(w["n_ue"], w["L"], 1234, dev, all_valid=not args.random_valid))
    m_rx, m_tx = w["ue"][0] * w["ue"][1], w["bs"][0] * w["bs"][1]
    out = torch.empty((w["n_ue"], m_rx, m_tx, w["N"]), dtype=torch.complex64, device=dev)
    prep = eng.prepare(rays, params, want_side=False)
    times = {v: [] for v in args.variants}
    t1 = []
    for v in args.variants:
        eng.channels(prep, out=out, variant=v)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.prepare(rays, params, want_side=False); e1.record(); torch.cuda.synchronize()
        t1.append(e0.elapsed_time(e1))
        # the timed launches follow each other without a host synchronisation in between (the first launch after an idle
        # gap measured 0.1-0.2 ms longer than the same kernel in second place); one untimed launch in front
        eng.channels(prep, out=out, variant=args.variants[-1])
        evs = []
        for v in args.variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.channels(prep, out=out, variant=v)
            e1.record()
            evs.append((v, e0, e1))
        torch.cuda.synchronize()
        for v, e0, e1 in evs:
            times[v].append(e0.elapsed_time(e1))
    if args.beams:
        import deepmimo_amd as dm
        F = np.array([dm.steering_vec(np.array(w["bs"]), phi=a).squeeze() for a in np.linspace(-60, 60, args.beams)])
        outb = torch.empty((w["n_ue"], m_rx, args.beams, w["N"]), dtype=torch.complex64, device=dev)
        eng.channels(prep, out=outb, tx_codebook=F)
        torch.cuda.synchronize()
        tb = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.channels(prep, out=outb, tx_codebook=F); e1.record(); torch.cuda.synchronize()
            tb.append(e0.elapsed_time(e1))
        bb = w["n_ue"] * 8 * m_rx * args.beams * w["N"]
        print(f"beam-space ({args.beams} beams): median {np.median(tb):.3f} ms -> {bb/np.median(tb)/1e6:.0f} GB/s of output, "
              f"{w['n_ue']/np.median(tb)*1e3/1e6:.2f} M users/s (vs materialise H then project: H alone is {m_tx/args.beams:.0f}x the bytes)")
    bytes_ = w["n_ue"] * (8 * m_rx * m_tx * w["N"] + 32 * w["L"])
    print(f"workload {args.workload} users {w['n_ue']}  stage-1 prep median {np.median(t1):.3f} ms")
    for v in args.variants:
        t = np.array(times[v])
        print(f"variant {v}: median {np.median(t):.3f} ms  min {t.min():.3f} ms  -> {bytes_/np.median(t)/1e6:.0f} GB/s "
              f"({w['n_ue']/np.median(t)*1e3/1e6:.2f} M users/s)")


if __name__ == "__main__":
    main()
