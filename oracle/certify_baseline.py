"""Certification of the in-repo CPU baseline - build container only.

Times oracle_np.compute_channels(style='reference') side by side with the imported reference
(Dataset.compute_channels) on identical synthetic rays at the config-1/2/3 shapes, and checks the outputs
agree.  The ratio is what makes bench.py's cpu_baseline (kind "port") a fair stand-in for the reference on
the GPU node, where /root/reference does not exist.

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python3 /root/repo/oracle/certify_baseline.py
"""
import io
import os
import sys
import time
from contextlib import redirect_stdout, redirect_stderr

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_np as onp  # noqa: E402

SHAPES = [("C1 4x1/1x1 L5 K64", 1000, [4, 1], [1, 1], 5, 64),
          ("C2 8x4/2x2 L10 K256", 300, [8, 4], [2, 2], 10, 256),
          ("C3 8x8/2x2 L25 K512", 60, [8, 8], [2, 2], 25, 512)]


def main():
    import deepmimo as dm
    print("| shape | users | reference s | in-repo port s | port / reference | max abs diff |")
    print("|---|---|---|---|---|---|")
    for name, n, bs, ue, L, K in SHAPES:
        rays = onp.synth_rays(n, L, seed=5, all_valid=True)
        p = dm.ChannelGenParameters()
        p.bs_antenna.shape, p.ue_antenna.shape = np.array(bs), np.array(ue)
        p.num_paths, p.ofdm.subcarriers, p.ofdm.selected_subcarriers = L, K, np.arange(K)
        op = onp.make_params(bs_antenna=dict(shape=bs), ue_antenna=dict(shape=ue), num_paths=L,
                             ofdm=dict(subcarriers=K, selected_subcarriers=np.arange(K)))
        t_ref, t_port = [], []
        for _ in range(2):
            ds = dm.Dataset({k: v.copy() for k, v in rays.items()})
            with redirect_stdout(io.StringIO()), redirect_stderr(io.StringIO()):
                t0 = time.perf_counter(); Href = ds.compute_channels(p); t_ref.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); H = onp.compute_channels(rays, op, style="reference")["channel"]; t_port.append(time.perf_counter() - t0)
        d = float(np.max(np.abs(H - Href)))
        print(f"| {name} | {n} | {min(t_ref):.2f} | {min(t_port):.2f} | {min(t_port)/min(t_ref):.2f} | {d:.1e} |")


if __name__ == "__main__":
    main()
