#!/usr/bin/env python3
"""Timing of the rx_filter = 1 path (ofdm.rx_filter: sinc low-pass, channel.py:166-168, 193-194) at the headline shape:
stage 2 = gains table (k3_lpf_*) + contraction with table-loaded gains, against the plain path on the same rays.

    python tools/lpf_bench.py [--users 20000] [--rounds 5]
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
    ap.add_argument("--users", type=int, default=20000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--workload", default="c3_headline")
    ap.add_argument("--quick", action="store_true", help="skip the plain path and the round-1 FFT")
    ap.add_argument("--plain-only", action="store_true", help="the plain path alone")
    ap.add_argument("--n", type=int, default=0, help="OFDM size N = K instead of the workload's (512: k3_lpf_fft512; others: the generic k3_lpf_fft_wave)")
    args = ap.parse_args()
    w = dict(bench.WORKLOADS[args.workload])
    w["n_ue"] = args.users
    if args.n:
        w["N"] = args.n
    dev = torch.device("cuda", 0)
    eng = ChannelEngine(0)
    rays = eng.upload_rays(bench.synth_device_rays(w["n_ue"], w["L"], 1234, dev))
    m_rx, m_tx = w["ue"][0] * w["ue"][1], w["bs"][0] * w["bs"][1]
    out = torch.empty((w["n_ue"], m_rx, m_tx, w["N"]), dtype=torch.complex64, device=dev)
    out_bytes = out.numel() * 8
    res = {}
    cases = (("plain (rx_filter = 0)", 0, "0"), ("rx_filter = 1, wave-per-path radix-8 FFT", 1, "0"),
             ("rx_filter = 1, workgroup-per-user radix-2 FFT (round 1)", 1, "1"))
    for label, lpf, old in (cases[:1] if args.plain_only else cases[1:2] if args.quick else cases):
        os.environ["DMX_LPF_OLD_FFT"] = old
        p = bench.make_params(w)
        p.ofdm.rx_filter = lpf
        prep = eng.prepare(rays, p, want_side=False)
        eng.channels(prep, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.channels(prep, out=out); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[label] = out.clone() if args.users <= 4096 else None
        t = float(np.median(ts))
        print(f"{label}: stage 2 median {t:.3f} ms  min {min(ts):.3f} ms -> {out_bytes / t / 1e6:.0f} GB/s of output "
              f"({out_bytes / t / 1e6 / 8000:.3f} of 8 TB/s)")
    a, b = res.get("rx_filter = 1, wave-per-path radix-8 FFT"), res.get("rx_filter = 1, workgroup-per-user radix-2 FFT (round 1)")
    if a is not None and b is not None:
        d = (a - b).abs().amax(dim=(1, 2, 3)) / b.abs().amax(dim=(1, 2, 3)).clamp_min(1e-30)
        print(f"new vs old FFT: worst |dH| / peak over users = {float(d.max()):.2e}")


if __name__ == "__main__":
    main()
