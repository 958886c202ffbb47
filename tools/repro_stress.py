#!/usr/bin/env python3
"""Bit-reproducibility stress: N launches of one stage-2 kernel on the same prepared users, per-user checksums against
the first launch.  `python tools/repro_stress.py [--bs 8x1] [--K 512] [--users 200000] [--launches 100] [--variant 0]`
The first mismatch is saved and classified (tests/_repro_dump.py -> gpurun_out/repro_dump_stress_tool.{npz,json}): one
failure is enough, this is not a tool for re-running until something fails."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepmimo_amd as dm  # noqa: E402
from deepmimo_amd.engine import ChannelEngine  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", default="8x1")
    ap.add_argument("--ue", default="1x1")
    ap.add_argument("--K", type=int, default=512)
    ap.add_argument("--users", type=int, default=200000)
    ap.add_argument("--launches", type=int, default=100)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--all-valid", action="store_true")
    ap.add_argument("--beams", type=int, default=0, help="stress dmx_beam_power with this many steering beams instead")
    ap.add_argument("--rx-filter", action="store_true")
    ap.add_argument("--adaptive", action="store_true", help="DMX_FLAG_ADAPTIVE_TERMS (default: three product terms)")
    args = ap.parse_args()
    bs = [int(x) for x in args.bs.split("x")]
    ue = [int(x) for x in args.ue.split("x")]
    n = args.users
    rays = onp.synth_rays(n, 25, seed=2024, all_valid=args.all_valid)
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape, p.ue_antenna.shape = np.array(bs), np.array(ue)
    p.ofdm.subcarriers = args.K
    p.ofdm.selected_subcarriers = np.arange(args.K)
    if args.rx_filter:
        p.ofdm.rx_filter = 1
    eng = ChannelEngine(0)
    prep = eng.prepare(eng.upload_rays(rays), p, want_side=True, adaptive_terms=args.adaptive)
    nk = prep.side["num_paths"].cpu().numpy()
    if args.beams:
        F = np.stack([dm.steering_vec(np.array(bs), phi=a).ravel() for a in np.linspace(-60, 60, args.beams)])
        ref = eng.beam_power(prep, F)[0].clone()
        total = 0
        for it in range(args.launches):
            amp = eng.beam_power(prep, F)[0]
            bad = torch.nonzero((amp != ref).any(dim=1)).flatten().cpu().numpy()
            total += len(bad)
            for u in bad[:3]:
                print(f"launch {it}: user {u} (kept paths {nk[u]}): max rel diff {float(((amp[u] - ref[u]).abs() / ref[u].abs().max()).max()):.2e}")
        print(f"beam power, {args.beams} beams, bs {args.bs} ue {args.ue} K {args.K}: {total} differing user-launches in {args.launches} launches x {n} users")
        return

    def checksums(H):
        bits = torch.view_as_real(H).view(torch.int32).reshape(n, -1)
        out = torch.empty(n, dtype=torch.int64, device=H.device)
        step = max(1, int(2e9 // (bits.shape[1] * 8)))
        w = torch.arange(bits.shape[1], device=H.device) % 251 + 1
        for a in range(0, n, step):
            out[a:a + step] = (bits[a:a + step].to(torch.int64) * w).sum(dim=1)
        return out

    H = eng.channels(prep, variant=args.variant)
    ref = checksums(H)
    H0 = H.clone() if H.numel() * 8 < 20e9 else None
    total = 0
    for it in range(args.launches):
        H = eng.channels(prep, out=H, variant=args.variant)
        bad = torch.nonzero(checksums(H) != ref).flatten().cpu().numpy()
        if len(bad) and not total:
            from tests._repro_dump import dump_from_checksums
            print("saved:", dump_from_checksums("stress_tool", eng, prep, p, rays, H, bad, variant=args.variant))
        total += len(bad)
        for u in bad[:3]:
            msg = f"launch {it}: user {u} (kept paths {nk[u]})"
            if H0 is not None:
                d = (H[u] - H0[u]).abs()
                idx = torch.nonzero(d.reshape(-1) > 0).flatten()
                msg += f": {len(idx)} elements differ, first flat indices {idx[:4].tolist()}, max |d| / peak {float(d.max() / H0[u].abs().max()):.2e}"
            print(msg)
    print(f"bs {args.bs} ue {args.ue} K {args.K} variant {args.variant} adaptive_terms={args.adaptive}: "
          f"{total} differing user-launches in {args.launches} launches x {n} users")


if __name__ == "__main__":
    main()
