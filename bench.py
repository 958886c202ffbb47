#!/usr/bin/env python3
"""Benchmark of the channel-generation hot path (BASELINE.json metric: user-channels/s).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path (stage 1 path prep + stage 2 paths x antennas x
subcarriers contraction) over the headline workload of BASELINE.json configs[2]: 100,000 users per
GPU, BS 8x8 (64) x UE 2x2 (4) antennas, 25 paths all valid (worst case), 512 subcarriers, all
selected; synthetic float32 rays generated on the device, resident in HBM before the timed region;
the complex64 channel tensor (104.9 GB per GPU) is written to HBM by every step.  Users are
block-partitioned over ranks with no data-path collective (weak scaling: per-GPU work is fixed).

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events around the stage-2
kernel on the stream it is launched on; `cpu_baseline` times the NumPy oracle's reference-style
per-user loop (oracle/oracle_np.py, style='reference') on a bounded user sample on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
WORKLOADS = {
    # name: users/GPU, bs_shape, ue_shape, paths, subcarriers
    "c3_headline": dict(n_ue=100_000, bs=[8, 8], ue=[2, 2], L=25, N=512),
    "c2_asu_shape": dict(n_ue=10_000, bs=[8, 4], ue=[2, 2], L=10, N=256),
    "c5_massive": dict(n_ue=6_250, bs=[16, 16], ue=[4, 4], L=25, N=1024),
    "tiny": dict(n_ue=512, bs=[8, 8], ue=[2, 2], L=25, N=512),
}


def synth_device_rays(n_ue, L, seed, device, all_valid=True):
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def U(lo, hi):
        return (torch.rand((n_ue, L), generator=g, device=device, dtype=torch.float32) * (hi - lo) + lo)

    rays = {"power": U(-140, -60), "phase": U(-180, 180), "delay": U(1e-8, 2e-6),
            "aoa_az": U(-180, 180), "aoa_el": U(0, 180), "aod_az": U(-180, 180), "aod_el": U(0, 180),
            "inter": torch.randint(0, 5, (n_ue, L), generator=g, device=device).to(torch.float32)}
    if not all_valid:
        nvalid = torch.randint(0, L + 1, (n_ue, 1), generator=g, device=device)
        pad = torch.arange(L, device=device)[None, :] >= nvalid
        for k in rays:
            rays[k] = rays[k].masked_fill(pad, float("nan"))
    return rays


def make_params(w):
    import deepmimo_amd as dm
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array(w["bs"])
    p.ue_antenna.shape = np.array(w["ue"])
    p.num_paths = w["L"]
    p.ofdm.subcarriers = w["N"]
    p.ofdm.selected_subcarriers = np.arange(w["N"])
    p.validate(w["n_ue"])
    return p


def _cpu_chunk(args):
    """One worker's share of the CPU baseline (module-level so multiprocessing can pickle it)."""
    w, n, seed = args
    from oracle import oracle_np as onp
    rays = onp.synth_rays(n, w["L"], seed=seed, all_valid=True)
    op = onp.make_params(bs_antenna=dict(shape=w["bs"]), ue_antenna=dict(shape=w["ue"]), num_paths=w["L"],
                         ofdm=dict(subcarriers=w["N"], selected_subcarriers=np.arange(w["N"])))
    t0 = time.perf_counter()
    onp.compute_channels(rays, op, style="reference")
    return time.perf_counter() - t0


def cpu_baseline_here(w, sample_users, workers):
    """Reference-style CPU generator (per-user complex128 broadcast + nansum loop, oracle/oracle_np.py
    style='reference') on a user sample: one process (the reference's execution model), then - for fairness - the
    same sample split over `workers` processes.  Runs in a process that never touches the GPU."""
    _cpu_chunk((w, max(1, min(16, sample_users // 8)), 99))      # untimed warm-up (imports, allocator, page faults)
    dt1 = min(_cpu_chunk((w, sample_users, 4321)) for _ in range(2))     # best of 2: host timing is noisy, favour the CPU
    out = {"value": sample_users / dt1, "unit": "user-channels/s", "cores": 1, "kind": "port",
           "sample": f"{sample_users} users of the same workload shape, all paths valid, NumPy oracle "
                     f"style='reference' (per-user complex128 broadcast+nansum loop), best of 2: {dt1:.1f} s, "
                     f"host has {os.cpu_count()} logical cores"}
    if workers > 1:
        import multiprocessing as mp
        per = max(1, sample_users // workers)
        with mp.get_context("fork").Pool(workers) as pool:
            t0 = time.perf_counter()
            pool.map(_cpu_chunk, [(w, per, 5000 + i) for i in range(workers)])
            dtw = time.perf_counter() - t0
        out["all_cores"] = {"value": per * workers / dtw, "unit": "user-channels/s", "cores": workers,
                            "sample": f"{workers} processes x {per} users, {dtw:.1f} s wall"}
    out["c_port"] = c_port_baseline(w, sample_users, workers)
    return out


def c_port_baseline(w, sample_users, workers):
    """The oracle's plain-C restatement (oracle/oracle_c.c: scalar loops, complex128 accumulation) on the same
    sample: one thread, then `workers` OpenMP threads over users.  Reported beside the NumPy figure because a
    compiled port is what a CPU deployment of this path would look like; `value` above stays the reference's own
    execution model."""
    from oracle import oracle_np as onp, oracle_c as oc
    op = onp.make_params(bs_antenna=dict(shape=w["bs"]), ue_antenna=dict(shape=w["ue"]), num_paths=w["L"],
                         ofdm=dict(subcarriers=w["N"], selected_subcarriers=np.arange(w["N"])))

    def run(n, threads, seed):
        rays = onp.synth_rays(n, w["L"], seed=seed, all_valid=True)
        t0 = time.perf_counter()
        oc.compute_channels(rays, op, threads=threads)
        return time.perf_counter() - t0

    run(max(1, min(16, sample_users // 8)), 1, 98)
    dt1 = min(run(sample_users, 1, 4321) for _ in range(2))
    res = {"value": sample_users / dt1, "unit": "user-channels/s", "cores": 1, "kind": "port",
           "sample": f"{sample_users} users, oracle/oracle_c.c, 1 thread, best of 2: {dt1:.1f} s"}
    if workers > 1:
        n = sample_users * min(workers, 8)
        dtw = min(run(n, workers, 777) for _ in range(2))
        res["all_cores"] = {"value": n / dtw, "unit": "user-channels/s", "cores": workers,
                            "sample": f"{n} users, {workers} OpenMP threads, best of 2: {dtw:.1f} s"}
    return res


def cpu_baseline(workload, users_override, sample_users, workers):
    """Run the CPU baseline in a child process (no CUDA context there, so forking workers is safe)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload,
           "--cpu-users", str(sample_users), "--cpu-workers", str(workers)]
    if users_override:
        cmd += ["--users", str(users_override)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed:\n" + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def measured_traffic(workload, n_ue, variant):
    """HBM bytes per stage-2 launch from the committed rocprofv3 PMC passes (profiles/traffic.json:
    WRITE_SIZE + 2 x FETCH_SIZE in KiB, separate --pmc runs, gfx950 FETCH correction applied).  Only
    reported when the profiled run matches this run's workload, user count and kernel variant."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(workload)
        if e and e["users"] == n_ue and e["variant"] == variant:
            return e["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3_headline", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 fp32 vector kernel, 2 MFMA kernel")
    ap.add_argument("--users", type=int, default=0, help="override users per GPU")
    ap.add_argument("--cpu-users", type=int, default=-1, help="CPU baseline sample size (0 = skip)")
    ap.add_argument("--random-valid", action="store_true", help="valid paths per user uniform in 0..L")
    ap.add_argument("--cpu-workers", type=int, default=-1, help="processes of the all-cores CPU figure (default min(16, cores))")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_workers < 0:
        args.cpu_workers = min(16, os.cpu_count() or 1)      # a one-GPU box gives this job a 16-core share
    if args.cpu_baseline_only:
        w = dict(WORKLOADS[args.workload])
        print(json.dumps(cpu_baseline_here(w, args.cpu_users, args.cpu_workers)))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsal only) ranks wrap around
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm) for the barrier / timing reduction; DMX_DIST_BACKEND=gloo lets two ranks
        # rehearse the multi-rank path on a single-GPU box (RCCL refuses two ranks on one device)
        backend = os.environ.get("DMX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        else:
            dist_mod.init_process_group(backend=backend)
        dist = dist_mod
    red_dev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")

    from deepmimo_amd.engine import ChannelEngine
    w = dict(WORKLOADS[args.workload])
    if args.users:
        w["n_ue"] = args.users
    n_ue = w["n_ue"]
    eng = ChannelEngine(dev_index)
    params = make_params(w)
    rays_t = synth_device_rays(n_ue, w["L"], 1234 + rank, dev, all_valid=not args.random_valid)
    rays = eng.upload_rays(rays_t)
    m_rx, m_tx = w["ue"][0] * w["ue"][1], w["bs"][0] * w["bs"][1]
    out = torch.empty((n_ue, m_rx, m_tx, w["N"]), dtype=torch.complex64, device=dev)

    # one preparation allocates the workspace once; every step then re-issues BOTH stages on the same buffers
    # (ChannelEngine.relaunch: two C-ABI calls, no allocation, no host-device copy, no sync)
    prep0 = eng.prepare(rays, params, want_side=False)
    import ctypes as C
    from deepmimo_amd import _native as nat

    def step(ev0=None, ev1=None):
        p, wsp = prep0.params_struct, C.c_void_p(prep0.workspace.data_ptr())
        stream = eng._stream_ptr()
        prep0.side["max_delay_key"].zero_()
        nat.check(eng.lib.dmx_path_prep(C.byref(prep0.rays_struct), C.byref(p), wsp, prep0.workspace_bytes,
                                        C.byref(prep0.side_struct), stream), "dmx_path_prep")
        if ev0 is not None:
            ev0.record(torch.cuda.current_stream(dev))
        nat.check(eng.lib.dmx_channels_fd(C.byref(p), wsp, prep0.n_ue, prep0.n_paths_loaded, 0, prep0.n_ue,
                                          C.c_void_p(out.data_ptr()), int(args.variant), stream), "dmx_channels_fd")
        if ev1 is not None:
            ev1.record(torch.cuda.current_stream(dev))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(*evs[i])
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    k2_ms = float(np.mean([a.elapsed_time(b) for a, b in evs])) if args.steps else float("nan")
    if dist:
        t = torch.tensor([elapsed, k2_ms], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, k2_ms = float(t[0]), float(t[1])

    # a cheap end-to-end sanity check of what was just written (not timed)
    chk = torch.view_as_real(out[:: max(1, n_ue // 64)])
    assert bool(torch.isfinite(chk).all()) and float(chk.abs().max()) > 0, "bench output is not finite / all zero"

    total_users = n_ue * world
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    bytes_per_user = 8 * m_rx * m_tx * w["N"] + 4 * w["L"] * 8
    achieved = n_ue * bytes_per_user / (k2_ms * 1e-3) / 1e9
    cmacs = n_ue * m_rx * m_tx * w["N"] * w["L"]
    res = {
        "metric": "user-channels/sec", "value": total_users / (elapsed / max(args.steps, 1)),
        "unit": "user-channels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_ue} users/GPU x BS {w['bs'][0]}x{w['bs'][1]} ({m_tx}) x UE "
                               f"{w['ue'][0]}x{w['ue'][1]} ({m_rx}) antennas x {w['L']} paths "
                               f"({'random valid count' if args.random_valid else 'all valid'}) x {w['N']} subcarriers",
                   "users_total": total_users, "parallelism": f"user-shard x{world}",
                   "fd_kernel_variant": args.variant,
                   "fd_kernel": {1: "k2_fd_valu", 2: "k2_fd_mfma", 9: "k2_fd_small"}.get(
                       int(args.variant) if args.variant else
                       eng.lib.dmx_fd_kernel_choice(C.byref(prep0.params_struct), prep0.n_paths_loaded), "k2_fd_mfma"),
                   "arithmetic": "fp32 results; k2_fd_mfma contracts on the f16 matrix cores with a 3-term split "
                                 "(hi*hi + hi*lo + lo*hi, fp32 accumulate, 1.4e-6 of peak measured), the other "
                                 "stage-2 kernels in fp32; stage 1 in float64",
                   "complex_macs_per_s": cmacs * world / (elapsed / max(args.steps, 1))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload, n_ue, args.variant),
                     "kernel": "stage-2 contraction (k2_fd_*)", "kernel_ms": k2_ms,
                     "algorithmic_bytes_per_launch": n_ue * bytes_per_user},
    }
    if rank == 0 and world == 1:
        cpu_users = args.cpu_users if args.cpu_users >= 0 else {"c3_headline": 400, "c2_asu_shape": 2000,
                                                                "c5_massive": 8, "tiny": 100}[args.workload]
        if cpu_users > 0:
            res["cpu_baseline"] = cpu_baseline(args.workload, 0, cpu_users, args.cpu_workers)
    if rank == 0:
        print(json.dumps(res))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
