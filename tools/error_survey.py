#!/usr/bin/env python3
"""Distribution of the HIP path's error against the oracle over random configurations (GPU box):
max over users of max|dH| / max|H_ref[user]| per configuration, per kernel variant."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepmimo_amd as dm
from oracle import oracle_np as onp
from tests._cases import channel_err

worst = {1: [], 2: []}
rng = np.random.default_rng(123)
for it in range(60):
    bs = [int(rng.integers(2, 17)), int(rng.integers(1, 9))]
    ue = [int(rng.integers(1, 3)), int(rng.integers(1, 3))]
    L = int(rng.integers(1, 26)); N = int(rng.choice([64, 256, 512, 1024])); K = min(N, int(rng.integers(8, 200)))
    n = 24
    rays = onp.synth_rays(n, L, seed=it, all_valid=True, max_delay=float(rng.choice([2e-6, 0.9 * N / 10e6])))
    rot = rng.integers(-180, 181, 3)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape, p.bs_antenna.rotation = np.array(bs), np.array(ue), rot
    p.num_paths, p.ofdm.subcarriers, p.ofdm.selected_subcarriers = L, N, np.arange(K)
    op = onp.make_params(bs_antenna=dict(shape=bs, rotation=rot), ue_antenna=dict(shape=ue), num_paths=L,
                         ofdm=dict(subcarriers=N, selected_subcarriers=np.arange(K)))
    ref = onp.compute_channels(rays, op)["channel"]
    for v in (1, 2):
        dm.config("fd_kernel_variant", v)
        H = dm.Dataset(dict(rays)).compute_channels(p)
        d, peak = channel_err(H, ref)
        worst[v].append(float(np.max(d / np.maximum(peak, 1e-300))))
dm.config("fd_kernel_variant", 0)
for v, name in ((1, "fp32 vector kernel"), (2, "f16x3 MFMA kernel")):
    w = np.array(worst[v])
    print(f"{name}: worst rel err over {len(w)} configs: max {w.max():.2e}  median {np.median(w):.2e}  (tolerance 5e-5)")
