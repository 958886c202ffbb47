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
MFMA_F16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA ~2.5 PFLOP/s
CARRIER_HZ = 3.5e9          # carrier of the Doppler term in config 5
WORKLOADS = {
    # name: users PER GPU (weak scaling: every rank generates this many), bs_shape, ue_shape, paths, subcarriers
    "c3_headline": dict(n_ue=100_000, bs=[8, 8], ue=[2, 2], L=25, N=512),
    "c2_asu_shape": dict(n_ue=10_000, bs=[8, 4], ue=[2, 2], L=10, N=256),
    # config 4 = 1M users over 8 GPUs: the per-GPU shard (131 GB of output); `--gpus 8 --workload c4_shard` IS config 4
    "c4_shard": dict(n_ue=125_000, bs=[8, 8], ue=[2, 2], L=25, N=512),
    # config 5 = 50k users over 8 GPUs, with the Doppler term: the per-GPU shard (209.7 GB of output)
    "c5_massive": dict(n_ue=6_250, bs=[16, 16], ue=[4, 4], L=25, N=1024, doppler=True),
    # DeepMIMO's default arrays (channel.py:36-46: BS 8x1, UE 1x1): the folded matrix-core kernel's regime
    "d8_default_arrays": dict(n_ue=200_000, bs=[8, 1], ue=[1, 1], L=25, N=512),
    "d16_k256": dict(n_ue=200_000, bs=[4, 4], ue=[1, 1], L=25, N=256),
    "d64_k256": dict(n_ue=100_000, bs=[8, 8], ue=[1, 1], L=25, N=256),      # 64 pairs: the folded kernel with workgroup-shared tables
    # headline shape, but the consumer of docs/manual.ipynb cell 105 fused in: 64-beam sweep, no [N, ., K] tensor written
    "c3_beam_power": dict(n_ue=100_000, bs=[8, 8], ue=[2, 2], L=25, N=512, beams=64),
    # the other two output modes of the same path at the headline shape: time domain (channel.py:285-287: [N, M_rx, M_tx, L]
    # taps, 5.1 GB) and the receive low-pass filter (ofdm.rx_filter = 1: FFT gains table + contraction; the headline shape: 105 GB of output + a 10 GB gains table)
    "c3_time_domain": dict(n_ue=100_000, bs=[8, 8], ue=[2, 2], L=25, N=512, td=True),
    "c3_rx_filter": dict(n_ue=100_000, bs=[8, 8], ue=[2, 2], L=25, N=512, lpf=True),
    "tiny": dict(n_ue=512, bs=[8, 8], ue=[2, 2], L=25, N=512),
    # the step BEFORE the path (core.py:186-258): a converted scenario folder of asu_campus size
    # (411 x 321 = 131,931 receivers, test/test_v3_correspondence.py:49) -> device SoA -> compute_channels with
    # DeepMIMO's default arrays.  A different kind of bench line: see bench_loader().
    "load_asu_shape": dict(n_ue=131_931, bs=[8, 1], ue=[1, 1], L=25, N=512, loader=True),
}


def synth_device_rays(n_ue, L, seed, device, all_valid=True, doppler=False):
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def U(lo, hi):
        return (torch.rand((n_ue, L), generator=g, device=device, dtype=torch.float32) * (hi - lo) + lo)

    rays = {"power": U(-140, -60), "phase": U(-180, 180), "delay": U(1e-8, 2e-6),
            "aoa_az": U(-180, 180), "aoa_el": U(0, 180), "aod_az": U(-180, 180), "aod_el": U(0, 180),
            "inter": torch.randint(0, 5, (n_ue, L), generator=g, device=device).to(torch.float32)}
    if doppler:                                            # vel U(-30, 30) m/s, acc U(-1, 1) m/s^2
        rays["doppler_vel"] = U(-30, 30)
        rays["doppler_acc"] = U(-1, 1)
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
    p.enable_doppler = int(bool(w.get("doppler")))
    p.freq_domain = 0 if w.get("td") else 1
    p.ofdm.rx_filter = int(bool(w.get("lpf")))
    p.validate(w["n_ue"])
    return p


def bench_loader(args, w, dev):
    """`--workload load_asu_shape`: scenario folder -> `dm.load(..., device='cuda')` -> `compute_channels`, timed end to
    end and in parts, against the reference's loading model (`scipy.io.loadmat` per matrix, core.py:241, then an upload)
    on the same files.  A "step" is one whole load + generate of the scenario; `value` stays user-channels/s (users of
    the scenario / step time) and `roofline` is the device transpose kernel's (`k_mat_to_rowmajor`: bytes of the stored
    payload read + bytes of the row-major matrix written, per field)."""
    import shutil
    import tempfile
    import scipy.io
    import deepmimo_amd as dm
    from deepmimo_amd import matio
    n, L = w["n_ue"] if not args.users else args.users, w["L"]
    root = tempfile.mkdtemp(prefix="dmx_scen_", dir=os.environ.get("TMPDIR", "/tmp"))
    folder = os.path.join(root, "asu_shape")
    os.makedirs(folder)
    try:
        rng = np.random.default_rng(7)
        nvalid = rng.integers(0, L + 1, size=(n, 1))
        pad = np.arange(L)[None, :] >= nvalid

        def U(lo, hi):
            m = rng.uniform(lo, hi, size=(n, L)).astype(np.float32)
            m[pad] = np.nan
            return m

        mats = {"power": U(-140, -60), "phase": U(-180, 180), "delay": U(1e-8, 2e-6), "aoa_az": U(-180, 180),
                "aoa_el": U(0, 180), "aod_az": U(-180, 180), "aod_el": U(0, 180)}
        inter = rng.integers(0, 5, size=(n, L)).astype(np.float32)
        inter[pad] = np.nan
        mats["inter"] = inter
        mats["rx_pos"] = rng.uniform(0, 400, size=(n, 3)).astype(np.float32)
        mats["tx_pos"] = np.array([[200.0, 160.0, 25.0]], dtype=np.float32)
        params = {"rt_params": {"frequency": 3.5e9}, "scene": {"num_scenes": 1}, "materials": {},
                  "txrx_sets": {"txrx_set_0": {"id": 0, "is_tx": True, "is_rx": False, "num_points": 1, "name": "bs"},
                                "txrx_set_1": {"id": 1, "is_tx": False, "is_rx": True, "num_points": n, "name": "ue"}}}
        with open(os.path.join(folder, "params.json"), "w") as f:
            json.dump(params, f)
        file_bytes = 0
        for k, v in mats.items():
            path = os.path.join(folder, dm.core.get_mat_filename(k, 0, 0, 1))
            scipy.io.savemat(path, {k: v})                       # what the converter writes (converter_utils.py:59-85)
            file_bytes += os.path.getsize(path)
        ray_bytes = 8 * n * L * 4
        chp = make_params(dict(w, n_ue=n))
        dm.config("channel_output", "torch")

        def one_step():
            t0 = time.perf_counter()
            ds = dm.load(folder, device="cuda")
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            H = ds.compute_channels(chp)
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            assert tuple(H.shape) == (n, 1, w["bs"][0] * w["bs"][1], w["N"]) and bool(torch.isfinite(torch.view_as_real(H[:: max(1, n // 64)])).all())
            los = ds.los
            return t1 - t0, t2 - t1, ds, los

        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):          # the loader prints one line per TX/RX pair, as the reference does
            for _ in range(max(1, args.warmup)):
                one_step()
            loads, gens = [], []
            for _ in range(max(1, args.steps)):
                a, b, ds, _ = one_step()
                loads.append(a)
                gens.append(b)
            # the parts of the device path, one field at a time: file pages -> HBM (as stored), then the layout kernel
            path = os.path.join(folder, dm.core.get_mat_filename("power", 0, 0, 1))
            evs = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                payload = torch.empty(n * L, dtype=torch.float32, device=dev).uniform_(-1, 1)
                out = torch.empty((n, L), dtype=torch.float32, device=dev)
                import ctypes as C
                from deepmimo_amd import _native as nat
                e0.record()
                nat.check(nat.load().dmx_mat_to_rowmajor_f32(C.c_void_p(payload.data_ptr()), 7, n, L, None, n, L,
                                                             C.c_void_p(out.data_ptr()),
                                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "dmx_mat_to_rowmajor_f32")
                e1.record()
                torch.cuda.synchronize(dev)
                evs.append(e0.elapsed_time(e1))
            k_ms = float(np.min(evs))
            t0 = time.perf_counter()
            t = matio.load_matrix_to_device(path, "power", dev)
            torch.cuda.synchronize(dev)
            one_field_s = time.perf_counter() - t0
            # the reference's model on the same files: scipy.io.loadmat per matrix + slicing (core.py:241-254), then the upload
            t0 = time.perf_counter()
            host = dm.load(folder)
            t_scipy = time.perf_counter() - t0
            t0 = time.perf_counter()
            _ = {k: torch.from_numpy(np.ascontiguousarray(host[k])).to(dev) for k in dm.consts.RAY_FIELDS}
            torch.cuda.synchronize(dev)
            t_upload = time.perf_counter() - t0
        dm.config.reset()
        load_s, gen_s = float(np.median(loads)), float(np.median(gens))
        step_s = load_s + gen_s
        k_bytes = 2 * n * L * 4
        res = {
            "metric": "user-channels/sec", "value": n / step_s, "unit": "user-channels/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"load_asu_shape: scenario folder of {n} receivers x {L} paths (8 ray matrices as MAT-v5 files, "
                                   f"{file_bytes / 1e6:.0f} MB on disk, page cache warm) -> dm.load(device='cuda') -> compute_channels "
                                   f"(BS {w['bs'][0]}x{w['bs'][1]}, UE 1x1, {w['N']} subcarriers, output kept in HBM); one step = the whole load + generate",
                       "users_total": n, "parallelism": "user-shard x1", "library_id": library_id()},
            "roofline": {"bound": "hbm", "achieved": k_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": k_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_mat_to_rowmajor (column-major payload -> row-major float32, one ray matrix)", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": k_bytes,
                         "note": "13 MB per launch: far too small to reach the HBM rate, and 8 such launches are 0.1 % of a load - the loader is bound by the host side (below)"},
            "loader": {"load_ms": load_s * 1e3, "compute_channels_ms": gen_s * 1e3,
                       "ray_bytes": ray_bytes, "load_GBps_on_ray_bytes": ray_bytes / load_s / 1e9,
                       "pcie_measured_GBps": 57.0, "load_frac_of_pcie": ray_bytes / load_s / 1e9 / 57.0,
                       "one_field_file_to_device_ms": one_field_s * 1e3,
                       "reference_model": {"what": "dm.load(folder) on the host = scipy.io.loadmat per matrix + slicing (core.py:241-254), then torch upload of the 8 ray fields",
                                           "loadmat_ms": t_scipy * 1e3, "upload_ms": t_upload * 1e3,
                                           "speedup_of_device_loader": (t_scipy + t_upload) / load_s}},
        }
        print(json.dumps(res))
    finally:
        shutil.rmtree(root, ignore_errors=True)


def _cpu_chunk(args):
    """One worker's share of the CPU baseline (module-level so multiprocessing can pickle it)."""
    w, n, seed = args
    from oracle import oracle_np as onp
    dop = bool(w.get("doppler"))
    rays = onp.synth_rays(n, w["L"], seed=seed, all_valid=True, with_doppler=dop)
    op = onp.make_params(bs_antenna=dict(shape=w["bs"]), ue_antenna=dict(shape=w["ue"]), num_paths=w["L"],
                         enable_doppler=int(dop), freq_domain=0 if w.get("td") else 1,
                         ofdm=dict(subcarriers=w["N"], selected_subcarriers=np.arange(w["N"]), rx_filter=int(bool(w.get("lpf")))))
    dkw = dict(doppler=dict(vel=rays["doppler_vel"], acc=rays["doppler_acc"], carrier_freq=CARRIER_HZ)) if dop else {}
    t0 = time.perf_counter()
    H = onp.compute_channels(rays, op, style="reference", **dkw)["channel"]
    if w.get("beams"):                                     # the consumer, as docs/manual.ipynb cell 105 writes it
        F = beam_codebook(w)
        np.abs(F @ H).mean(axis=1).mean(axis=-1)
    return time.perf_counter() - t0


def beam_codebook(w):
    """The notebook's grid of beams (docs/manual.ipynb cell 105): steering vectors over -60..60 degrees."""
    import deepmimo_amd as dm
    return np.array([dm.steering_vec(np.array(w["bs"]), phi=azi).squeeze()
                     for azi in np.around(np.linspace(-60, 60, w["beams"]), 2)]).reshape(w["beams"], -1)


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
    if not (w.get("beams") or w.get("td") or w.get("lpf")):
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

    dop = bool(w.get("doppler"))
    op["enable_doppler"] = int(dop)

    def run(n, threads, seed):
        rays = onp.synth_rays(n, w["L"], seed=seed, all_valid=True, with_doppler=dop)
        dkw = dict(doppler=dict(vel=rays["doppler_vel"], acc=rays["doppler_acc"], carrier_freq=CARRIER_HZ)) if dop else {}
        t0 = time.perf_counter()
        oc.compute_channels(rays, op, threads=threads, **dkw)
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
    # one BLAS / OpenMP thread per process: "cores" then is what it says (only the rx_filter oracle calls BLAS at all, and
    # 16 worker processes x 256 BLAS threads each is how that leg once took a minute)
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed:\n" + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def library_id():
    """First 12 hex digits of the SHA-256 of the product library this process runs (the built .so travels with the repo
    snapshot, so the id names one build everywhere)."""
    import hashlib
    from deepmimo_amd import _native as nat
    with open(nat.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:12]


def measured_traffic(workload, n_ue, variant, what="hbm_bytes_per_launch"):
    """HBM bytes per stage-2 launch from the committed rocprofv3 PMC passes (profiles/traffic.json:
    WRITE_SIZE + 2 x FETCH_SIZE, separate --pmc runs, gfx950 FETCH correction applied; written by
    tools/make_traffic_json.py).  Counters cannot be collected inside this process, so the figure comes from the
    profile - but ONLY when that profile was taken on this very library build (`library_id`), workload, user count and
    kernel variant; otherwise `traffic` is null and `traffic_note` says why."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(workload)
        if not e:
            return None, "no PMC profile committed for this workload"
        if e["users"] != n_ue or e["variant"] != variant:
            return None, "committed PMC profile is for another user count / kernel variant"
        if e.get("library_id") != library_id():
            return None, f"committed PMC profile ({e.get('profile', '?')}) was taken on library build {e.get('library_id')}, this run is {library_id()}"
        if what == "mfma_busy_frac":
            return e.get("mfma_busy_frac"), f"SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), {e.get('profile', '?')}, this library build"
        return e["hbm_bytes_per_launch"], f"rocprofv3 PMC passes {e.get('profile', '?')} on this library build"
    except Exception as ex:                                    # a missing / malformed profile is not a bench failure
        return None, f"profiles/traffic.json unreadable: {ex}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3_headline", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 fp32 vector kernel, 2 MFMA kernel, 9 small-output, 12 folded")
    ap.add_argument("--users", type=int, default=0, help="override users per GPU")
    ap.add_argument("--cpu-users", type=int, default=-1, help="CPU baseline sample size (0 = skip)")
    ap.add_argument("--random-valid", action="store_true", help="valid paths per user uniform in 0..L")
    ap.add_argument("--skip-adaptive", action="store_true", help="no extra launches with the opt-in adaptive-precision flag (profiling runs: one kind of launch only)")
    ap.add_argument("--cpu-workers", type=int, default=-1, help="processes of the all-cores CPU figure (default min(16, cores))")
    ap.add_argument("--gather", action="store_true", help="N > 1: also time the side-product all_gather and a bounded channel-slice gather to rank 0")
    ap.add_argument("--gather-users", type=int, default=2048, help="users per rank in the gathered channel slice")
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
    if w.get("loader"):
        if world != 1:
            raise SystemExit("load_asu_shape is a single-process workload")
        return bench_loader(args, w, dev)
    if args.users:
        w["n_ue"] = args.users
    n_ue = w["n_ue"]
    eng = ChannelEngine(dev_index)
    params = make_params(w)
    dop = bool(w.get("doppler"))
    rays_t = synth_device_rays(n_ue, w["L"], 1234 + rank, dev, all_valid=not args.random_valid, doppler=dop)
    rays = eng.upload_rays(rays_t)
    m_rx, m_tx = w["ue"][0] * w["ue"][1], w["bs"][0] * w["bs"][1]
    n_beams = int(w.get("beams", 0))
    import ctypes as C
    from deepmimo_amd import _native as nat

    # one preparation allocates the workspace once; every step then re-issues BOTH stages on the same buffers
    # (two C-ABI calls, no allocation, no host-device copy, no sync)
    # what the timed step runs is what Dataset.compute_channels runs: stage 1 in its "light" form (LoS, path counts - and
    # the FoV mask when a FoV is set - written every step) and stage 2 in the API's default arithmetic (three product
    # terms for every path, dmx_params.flags = 0)
    prep0 = eng.prepare(rays, params, want_side="light", carrier_freq=CARRIER_HZ if dop else 0.0)
    p0, wsp = prep0.params_struct, C.c_void_p(prep0.workspace.data_ptr())
    assert bool(p0.enable_doppler) == dop, "the Doppler term of this workload is not active"
    if n_beams:
        # fused consumer (docs/manual.ipynb cell 105): per-beam mean amplitude [n_ue, n_beams]; no channel tensor exists
        cb = torch.from_numpy(beam_codebook(w)).to(device=dev, dtype=torch.complex64).contiguous()
        out = torch.empty((n_ue, n_beams), dtype=torch.float32, device=dev)
        best = torch.empty((n_ue,), dtype=torch.int32, device=dev)
        bws_bytes = int(eng.lib.dmx_beam_workspace_bytes(C.byref(p0), n_ue, prep0.n_paths_loaded, n_beams))
        bws = torch.empty(bws_bytes + 256, dtype=torch.uint8, device=dev)
        bws_ptr = bws.data_ptr() + (-bws.data_ptr()) % 256
    else:
        out = torch.empty((n_ue, m_rx, m_tx, w["L"] if w.get("td") else w["N"]), dtype=torch.complex64, device=dev)
    if w.get("lpf"):
        lws_bytes = int(eng.lib.dmx_lpf_workspace_bytes(C.byref(p0), n_ue, prep0.n_paths_loaded))
        lws = torch.empty(lws_bytes + 256, dtype=torch.uint8, device=dev)
        lws_ptr = lws.data_ptr() + (-lws.data_ptr()) % 256

    def step(ev0=None, ev1=None, prep=None):
        prep = prep or prep0
        p = prep.params_struct
        wsp_ = C.c_void_p(prep.workspace.data_ptr())
        stream = eng._stream_ptr()
        prep.side["max_delay_key"].zero_()
        nat.check(eng.lib.dmx_path_prep(C.byref(prep.rays_struct), C.byref(p), wsp_, prep.workspace_bytes,
                                        C.byref(prep.side_struct), stream), "dmx_path_prep")
        if ev0 is not None:
            ev0.record(torch.cuda.current_stream(dev))
        if n_beams:
            nat.check(eng.lib.dmx_beam_power(C.byref(p), wsp_, prep.n_ue, prep.n_paths_loaded, 0, prep.n_ue,
                                             C.c_void_p(cb.data_ptr()), n_beams, C.c_void_p(bws_ptr), bws_bytes,
                                             C.c_void_p(out.data_ptr()), C.c_void_p(best.data_ptr()), stream), "dmx_beam_power")
        elif w.get("td"):
            nat.check(eng.lib.dmx_channels_td(C.byref(p), wsp_, prep.n_ue, prep.n_paths_loaded, 0, prep.n_ue,
                                              C.c_void_p(out.data_ptr()), stream), "dmx_channels_td")
        elif w.get("lpf"):
            nat.check(eng.lib.dmx_channels_fd_lpf(C.byref(p), wsp_, prep.n_ue, prep.n_paths_loaded, 0, prep.n_ue,
                                                  C.c_void_p(lws_ptr), lws_bytes, C.c_void_p(out.data_ptr()), stream), "dmx_channels_fd_lpf")
        else:
            nat.check(eng.lib.dmx_channels_fd(C.byref(p), wsp_, prep.n_ue, prep.n_paths_loaded, 0, prep.n_ue,
                                              C.c_void_p(out.data_ptr()), int(args.variant), stream), "dmx_channels_fd")
        if ev1 is not None:
            ev1.record(torch.cuda.current_stream(dev))

    # the same two calls with the opt-in DMX_FLAG_ADAPTIVE_TERMS (a user's weak last path group in one product term),
    # BEFORE the warm-up and the timed region and outside both: reported beside the timed number.  The flag is a field of
    # the parameter block of a second preparation - nothing in the process changes.
    k2_ms_ad = None
    if args.steps and not (w.get("td") or w.get("lpf")) and not args.skip_adaptive:
        prep_ad = eng.prepare(rays, params, want_side="light", carrier_freq=CARRIER_HZ if dop else 0.0, adaptive_terms=True)
        for _ in range(8):                                         # the chip ramps its clocks over the first launches: a fair
            step(prep=prep_ad)                                     # comparison needs this leg as warm as the timed one below
        ev3 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(5, args.steps))]
        for a_, b_ in ev3:
            step(a_, b_, prep=prep_ad)
        torch.cuda.synchronize(dev)
        k2_ms_ad = float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev3]))
        del prep_ad

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
    chk = out[:: max(1, n_ue // 64)]
    chk = torch.view_as_real(chk) if chk.is_complex() else chk
    assert bool(torch.isfinite(chk).all()) and float(chk.abs().max()) > 0, "bench output is not finite / all zero"

    total_users = n_ue * world
    step_s = elapsed / max(args.steps, 1)
    choice = int(args.variant) if args.variant else int(eng.lib.dmx_fd_kernel_choice(C.byref(p0), prep0.n_paths_loaded))
    kernel = "k2c_beam_power" if n_beams else {1: "k2_fd_valu", 2: "k2_fd_mfma", 9: "k2_fd_small", 12: "k2_fd_fold"}.get(choice, "k2_fd_mfma")
    if w.get("td"):
        kernel = "k4_td"
    if w.get("lpf"):
        kernel = ("k3_lpf_fft512" if w["N"] == 512 else "k3_lpf_fft_wave") + " + k2_fd_mfma (table-fed)"
    split = kernel in ("k2_fd_mfma", "k2_fd_fold", "k2c_beam_power") or bool(w.get("lpf"))
    rows = m_rx * (n_beams if n_beams else m_tx)
    cmacs = n_ue * rows * (1 if w.get("td") else w["N"]) * w["L"]
    if n_beams:
        # compute-bound by construction (nothing of size [N, ., K] touches HBM): priced against the dense f16 MFMA peak
        # with the ALGORITHMIC flops, 8 real flops per complex MAC of the (rx, beam) x path x subcarrier contraction;
        # the matrix cores execute 3 split terms on 32 padded path slots, i.e. 3 * 32/25 times that
        flops = 8.0 * cmacs
        ach = flops / (k2_ms * 1e-3) / 1e12
        busy, busy_note = measured_traffic(args.workload, n_ue, args.variant, "mfma_busy_frac")
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_F16_PEAK_TFLOPS, "traffic": None, "mfma_busy_frac": busy, "mfma_busy_note": busy_note,
                "kernel": "k2c_beam_power (+ k2b_beam_project)", "kernel_ms": k2_ms,
                "algorithmic_flops_per_launch": flops,
                "executed_mfma_flops_per_launch": 2.0 * n_ue * ((rows + 31) // 32 * 32) * (2 * w["N"]) * 64 * 3,
                "output_not_written_bytes": n_ue * 8 * rows * w["N"]}
    else:
        last = w["L"] if w.get("td") else w["N"]
        bytes_per_user = 8 * m_rx * m_tx * last + 4 * w["L"] * (10 if dop else 8)
        achieved = n_ue * bytes_per_user / (k2_ms * 1e-3) / 1e9
        traffic, note = measured_traffic(args.workload, n_ue, args.variant)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": note,
                "kernel": f"stage-2 contraction ({kernel})", "kernel_ms": k2_ms,
                "algorithmic_bytes_per_launch": n_ue * bytes_per_user}
    if k2_ms_ad is not None and split:
        roof["kernel_ms_adaptive_terms"] = k2_ms_ad
        roof["frac_adaptive_terms"] = roof["frac"] * k2_ms / k2_ms_ad
    res = {
        "metric": "user-channels/sec", "value": total_users / step_s,
        "unit": "user-channels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x3-split MFMA, f32 accumulate" if split else "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n_ue} users/GPU x BS {w['bs'][0]}x{w['bs'][1]} ({m_tx}) x UE "
                               f"{w['ue'][0]}x{w['ue'][1]} ({m_rx}) antennas x {w['L']} paths "
                               f"({'random valid count' if args.random_valid else 'all valid'}) x {w['N']} subcarriers"
                               + (" - TIME DOMAIN taps [N, M_rx, M_tx, L]" if w.get("td") else "")
                               + (" - with ofdm.rx_filter = 1 (sinc low-pass)" if w.get("lpf") else "")
                               + (" + Doppler term (f_c 3.5 GHz)" if dop else "")
                               + (f" -> {n_beams}-beam sweep mean |F @ H| (no channel tensor written)" if n_beams else ""),
                   "users_total": total_users, "parallelism": f"user-shard x{world}",
                   "fd_kernel_variant": args.variant, "fd_kernel": kernel, "library_id": library_id(),
                   "arithmetic": "fp32 results; the matrix-core kernels (k2_fd_mfma, k2_fd_fold, k2c_beam_power) contract "
                                 "f16 hi/lo splits of fp32 operands in 3 terms (hi*hi + hi*lo + lo*hi, fp32 accumulate; "
                                 "<= 2e-6 of each user's peak measured against the float64 oracle) for EVERY path - the "
                                 "API's default and what `value` / `roofline.frac` are measured on; "
                                 "roofline.kernel_ms_adaptive_terms is the same launch with the opt-in "
                                 "dmx_params.flags = DMX_FLAG_ADAPTIVE_TERMS (a user's last 8-path group in ONE term when all "
                                 "its paths are >= 66 dB below the user's strongest: <= 7.6e-6 of the strongest path worst "
                                 "case; fires for ~95 % of the synthetic users because their powers are uniform over 80 dB, "
                                 "rarely on ray-traced data); the other stage-2 kernels in fp32; stage 1 in float64",
                   "stage1_side_products": "light (los, num_paths written every step)",
                   "complex_macs_per_s": cmacs * world / step_s},
        "roofline": roof,
    }
    if dist and args.gather:
        res["gather"] = gather_leg(dist, eng, rays, params, out, n_ue, world, rank, dev, args.gather_users,
                                   CARRIER_HZ if dop else 0.0)
    if rank == 0 and world == 1:
        cpu_users = args.cpu_users if args.cpu_users >= 0 else {"c3_headline": 400, "c2_asu_shape": 2000, "c4_shard": 400,
                                                                "c5_massive": 8, "d8_default_arrays": 8000, "d16_k256": 8000, "d64_k256": 3000, "c3_time_domain": 20000, "c3_rx_filter": 200,
                                                                "c3_beam_power": 300, "tiny": 100}[args.workload]
        if cpu_users > 0:
            res["cpu_baseline"] = cpu_baseline(args.workload, 0, cpu_users, args.cpu_workers)
    if rank == 0:
        print(json.dumps(res))
    if dist:
        dist.destroy_process_group()


def gather_leg(dist, eng, rays, params, out, n_ue, world, rank, dev, gather_users, carrier):
    """Bringing results together, timed SEPARATELY from generation: (1) all_gather of the small per-user
    side products los / num_paths (int32 [N]); (2) point-to-point fan-in of a BOUNDED user slice of the output to
    rank 0 (the full tensors of configs 3-5 do not fit one GPU).  On RCCL the tensors stay in HBM and each peer's block
    crosses its one direct xGMI link (~153 GB/s per link is the bound, 7 links ingest at the root concurrently); under
    the gloo rehearsal (DMX_DIST_BACKEND=gloo, ranks sharing one GPU) they go through host memory."""
    from deepmimo_amd import dist as ddist
    on_dev = dist.get_backend() == "nccl"
    prep = eng.prepare(rays, params, want_side=True, carrier_freq=carrier)
    los, npaths = prep.side["los"], prep.side["num_paths"]
    g = max(1, min(n_ue, gather_users))
    piece = out[:g].contiguous()
    if not on_dev:
        los, npaths, piece = los.cpu(), npaths.cpu(), piece.cpu()
    res = {"backend": dist.get_backend(), "device_tensors": on_dev}

    def timed(fn, reps=3):
        fn()                                                      # warm-up: communicator setup, first-touch
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            dist.barrier()
            best = min(best, time.perf_counter() - t0)
        t = torch.tensor([best], dtype=torch.float64, device=dev if on_dev else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    full = {}
    t_side = timed(lambda: full.update(los=ddist.all_gather_users(los, n_ue * world), num_paths=ddist.all_gather_users(npaths, n_ue * world)))
    assert full["los"].shape[0] == n_ue * world
    res["side_products"] = {"what": "all_gather of los + num_paths (int32 [users_total] each)", "ms": t_side * 1e3,
                            "bytes_total": 2 * 4 * n_ue * world}
    got = {}
    t_g = timed(lambda: got.update(x=ddist.gather_users_to_root(piece, g * world, dst=0)))
    per_peer = piece.numel() * piece.element_size()
    if rank == 0:
        assert got["x"].shape[0] == g * world
    res["channel_slice_to_root"] = {"what": f"point-to-point fan-in of {g} users per rank to rank 0 (batch_isend_irecv)",
                                    "ms": t_g * 1e3, "bytes_per_peer": per_peer,
                                    "GBps_per_peer": per_peer / t_g / 1e9,
                                    "GBps_root_ingest": per_peer * (world - 1) / t_g / 1e9,
                                    "bound": "one direct xGMI link per peer, ~153 GB/s (MI355X_MICROARCH.md / SURVEY.md 8e)"}
    return res


if __name__ == "__main__":
    main()
