#!/usr/bin/env python3
"""Where `dm.load(folder, device='cuda')` spends its time (GPU box): the device pipeline of the ray matrices, the host
matrices (scipy), the rest.  python tools/loader_parts.py [receivers]"""
import json, os, sys, tempfile, time
import numpy as np, scipy.io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepmimo_amd as dm
from deepmimo_amd import matio

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 131931, 25
root = tempfile.mkdtemp(prefix="dmx_parts_", dir=os.environ.get("TMPDIR", "/tmp"))
folder = os.path.join(root, "s"); os.makedirs(folder)
rng = np.random.default_rng(1)
mats = {k: rng.uniform(0, 1, size=(n, L)).astype(np.float32) for k in ("power", "phase", "delay", "aoa_az", "aoa_el", "aod_az", "aod_el", "inter")}
mats["rx_pos"] = rng.uniform(0, 400, size=(n, 3)).astype(np.float32); mats["tx_pos"] = np.zeros((1, 3), np.float32)
json.dump({"rt_params": {"frequency": 3.5e9}, "scene": {"num_scenes": 1}, "materials": {},
           "txrx_sets": {"txrx_set_0": {"id": 0, "is_tx": True, "is_rx": False, "num_points": 1, "name": "bs"},
                         "txrx_set_1": {"id": 1, "is_tx": False, "is_rx": True, "num_points": n, "name": "ue"}}}, open(os.path.join(folder, "params.json"), "w"))
for k, v in mats.items():
    scipy.io.savemat(os.path.join(folder, dm.core.get_mat_filename(k, 0, 0, 1)), {k: v})
items = [(os.path.join(folder, dm.core.get_mat_filename(k, 0, 0, 1)), k) for k in dm.consts.RAY_FIELDS]

def med(f, reps=15):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[3:]))

import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    t_all = med(lambda: dm.load(folder, device="cuda"))
t_dev = med(lambda: matio.load_matrices_to_device(items, "cuda"))
t_pos = med(lambda: [scipy.io.loadmat(os.path.join(folder, dm.core.get_mat_filename(k, 0, 0, 1)))[k] for k in ("rx_pos", "tx_pos")])
stage = matio._staging(1 << 20); nb = 13_193_100
pinned = torch.empty(8 * nb, dtype=torch.uint8, pin_memory=True); d = torch.empty(8 * nb, dtype=torch.uint8, device="cuda")
t_h2d = med(lambda: d.copy_(pinned, non_blocking=True))
def reads(th):
    from concurrent.futures import ThreadPoolExecutor
    arr = pinned.numpy()
    def r(i):
        fd = os.open(items[i][0], os.O_RDONLY); got = 0; mv = memoryview(arr[i * nb:(i + 1) * nb])
        while got < nb - 4096:
            k = os.preadv(fd, [mv[got:]], 128 + got)
            if k <= 0: break
            got += k
        os.close(fd)
    with ThreadPoolExecutor(th) as p: list(p.map(r, range(8)))
print(json.dumps({"receivers": n, "dm_load_ms": t_all, "ray_matrices_pipeline_ms": t_dev, "scipy_rx_tx_pos_ms": t_pos,
                  "h2d_105MB_pinned_ms": t_h2d, "preadv_8_files_ms": {th: med(lambda: reads(th), 9) for th in (1, 2, 4, 8)},
                  "cpus": os.cpu_count()}))
