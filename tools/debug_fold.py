#!/usr/bin/env python3
"""Where does the folded kernel (variant 12) differ from another variant?  Debug helper for the GPU box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepmimo_amd as dm
from tests._cases import load_golden
from tests.test_gpu_parity import _dataset, _dm_params

name = sys.argv[1] if len(sys.argv) > 1 else "g03_rot_fov"
case, rays, ue_rot, ref = load_golden(name)
out = {}
for v in (1, 12):
    dm.config("fd_kernel_variant", v)
    ds = _dataset(case, rays)
    out[v] = ds.compute_channels(_dm_params(case, ue_rot))
    npaths = ds.num_paths
H1, H12 = out[1], out[12]
print("shape", H12.shape, "finite v1", np.isfinite(H1).all(), "finite v12", np.isfinite(H12).all())
bad = ~np.isfinite(H12)
print("non-finite entries:", bad.sum())
idx = np.argwhere(bad)
print("first non-finite (user, rx, tx, k):", idx[:20].tolist())
users = np.unique(idx[:, 0]) if idx.size else []
print("users with non-finite:", list(users)[:30], "their num_paths:", [int(npaths[u]) for u in users][:30])
d = np.abs(np.where(bad, 0, H12) - H1).reshape(len(H1), -1).max(axis=1)
pk = np.abs(H1).reshape(len(H1), -1).max(axis=1)
print("per-user rel err:", np.round(d / np.maximum(pk, 1e-30), 8).tolist())
print("num_paths:", npaths.tolist())
