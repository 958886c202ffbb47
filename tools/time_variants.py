#!/usr/bin/env python3
"""Timing of the non-default variants (rx_filter, time domain, Doppler) at the headline antenna/path shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deepmimo_amd.engine import ChannelEngine

def run(name, n, **kw):
    w = dict(bench.WORKLOADS["c3_headline"]); w["n_ue"] = n
    p = bench.make_params(w)
    for k, v in kw.items():
        if k in ("rx_filter",): p.ofdm[k] = v
        else: p[k] = v
    eng = ChannelEngine(0)
    rays = eng.upload_rays(bench.synth_device_rays(n, w["L"], 1, torch.device("cuda", 0)))
    prep = eng.prepare(rays, p, want_side=False)
    out = eng.channels(prep); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.channels(prep, out=out); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(f"{name:28s} {n:7d} users: {min(ts):9.3f} ms  -> {n/min(ts)*1e3/1e6:.3f} M users/s  out {tuple(out.shape)}")

run("frequency domain (default)", 20000)
run("rx_filter = 1 (LPF)", 20000, rx_filter=1)
run("time domain", 100000, freq_domain=0)
