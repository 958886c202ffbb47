#!/usr/bin/env python3
"""How fast can a channel tensor reach a NumPy array?  (Dataset.compute_channels returns NumPy by default, as the
reference does; at DeepMIMO's default-sized arrays the copy, not the kernels, is what a caller waits for.)

    python tools/host_copy_probe.py [--gb 4]

A  tensor.cpu().numpy()                      (pageable destination, torch's staging)
B  torch.empty(pin_memory=True) + copy_      (allocation and copy timed separately)
C  chunks through two pinned staging buffers on a copy stream, np.copyto into a pageable array by T threads
"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=4.0)
    ap.add_argument("--chunk-mb", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = int(args.gb * 1e9) // 8
    src = torch.randn(n, 2, device=dev, dtype=torch.float32)
    src = torch.view_as_complex(src)
    torch.cuda.synchronize()
    gb = n * 8 / 1e9

    t0 = time.perf_counter(); a = src.cpu().numpy(); t1 = time.perf_counter()
    print(f"A  .cpu().numpy(): {t1 - t0:.3f} s = {gb / (t1 - t0):.2f} GB/s")
    t0 = time.perf_counter(); a2 = src.cpu().numpy(); t1 = time.perf_counter()
    print(f"A' second time:    {t1 - t0:.3f} s = {gb / (t1 - t0):.2f} GB/s")
    del a2

    t0 = time.perf_counter(); pin = torch.empty(n, dtype=torch.complex64, pin_memory=True); t1 = time.perf_counter()
    pin.copy_(src, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B  pinned alloc {t1 - t0:.3f} s ({gb / (t1 - t0):.2f} GB/s), copy {t2 - t1:.3f} s = {gb / (t2 - t1):.2f} GB/s; total {gb / (t2 - t0):.2f} GB/s")
    assert np.array_equal(pin.numpy()[:1000], a[:1000])
    t0 = time.perf_counter(); pin.copy_(src, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"B' copy into the warm pinned buffer: {gb / (t1 - t0):.2f} GB/s")
    del pin

    ce = args.chunk_mb * (1 << 20) // 8
    stage = [torch.empty(ce, dtype=torch.complex64, pin_memory=True) for _ in range(2)]
    stage_np = [s.numpy() for s in stage]
    copy_stream = torch.cuda.Stream(device=dev)
    for threads in (1, 2, 4, 8):
        dst = np.empty(n, dtype=np.complex64)
        t0 = time.perf_counter()
        evs = [None, None]
        chunks = [(b, min(ce, n - b)) for b in range(0, n, ce)]

        def drain(i, b, cnt):
            evs[i].synchronize()
            if threads == 1:
                np.copyto(dst[b:b + cnt], stage_np[i][:cnt])
            else:
                per = (cnt + threads - 1) // threads
                ts = [threading.Thread(target=np.copyto, args=(dst[b + k * per:b + min(cnt, (k + 1) * per)], stage_np[i][k * per:min(cnt, (k + 1) * per)]))
                      for k in range(threads) if k * per < cnt]
                [t.start() for t in ts]; [t.join() for t in ts]

        pending = None
        for ci, (b, cnt) in enumerate(chunks):
            i = ci & 1
            with torch.cuda.stream(copy_stream):
                stage[i][:cnt].copy_(src[b:b + cnt], non_blocking=True)
                evs[i] = torch.cuda.Event(); evs[i].record(copy_stream)
            if pending is not None:
                drain(*pending)
            pending = (i, b, cnt)
        drain(*pending)
        t1 = time.perf_counter()
        ok = np.array_equal(dst[:1000], a[:1000]) and np.array_equal(dst[-1000:], a[-1000:])
        print(f"C  staged, {args.chunk_mb} MB chunks, {threads} copy thread(s): {t1 - t0:.3f} s = {gb / (t1 - t0):.2f} GB/s  ok={ok}")
        del dst


if __name__ == "__main__":
    main()
