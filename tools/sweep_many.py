#!/usr/bin/env python3
"""The randomized parity sweep of tests/test_gpu_random_sweep.py over MORE seeds than the suite runs (GPU box):
    python tools/sweep_many.py [first_seed last_seed]       (default 120 1500)
Every configuration against the NumPy oracle; prints the failing seeds, a progress line every 100 and the count."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import tests.test_gpu_random_sweep as sw  # noqa: E402

first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (120, 1500)
fails, t0, modes = 0, time.time(), {}
for seed in range(first, last):
    try:
        c = sw._run_hip(seed)[0]
        modes[(c["mode"], c["N"] if c["mode"] == "lpf" else 0)] = modes.get((c["mode"], c["N"] if c["mode"] == "lpf" else 0), 0) + 1
        sw.test_random_configuration(seed)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("FAIL seed", seed, str(e)[:300].replace("\n", " "), flush=True)
        if fails > 10:
            break
    if (seed - first) % 100 == 99:
        print(f"... {seed - first + 1} seeds, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print("done, seeds", first, "..", last - 1, "failures:", fails)
print("configurations by (mode, N of the rx_filter ones):", dict(sorted(modes.items(), key=str)))
