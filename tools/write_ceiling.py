#!/usr/bin/env python3
"""Practical HBM write ceiling on this box: time torch fill / copy kernels over a buffer of the size
of the headline output (104.9 GB), for comparison with the stage-2 kernel's write rate."""
import torch
n = 100_000 * 4 * 64 * 512            # complex64 elements
x = torch.empty(n, dtype=torch.complex64, device="cuda")
xr = torch.view_as_real(x)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), sorted(ts)[len(ts) // 2]
gb = n * 8 / 1e9
mn, md = t(lambda: xr.zero_())
print(f"zero_ (memset)      {gb:.1f} GB: median {md:.2f} ms min {mn:.2f} ms -> {gb/md*1e3:.0f} GB/s")
mn, md = t(lambda: xr.fill_(1.5))
print(f"fill_ (store only)  {gb:.1f} GB: median {md:.2f} ms min {mn:.2f} ms -> {gb/md*1e3:.0f} GB/s")
h = n // 2
y = xr[:h]
mn, md = t(lambda: xr[h:2 * h].copy_(y))
print(f"copy  (read+write)  {gb:.1f} GB moved: median {md:.2f} ms -> {gb/md*1e3:.0f} GB/s total traffic")
