"""GPU coverage of the N > 1 path with the real kernels: 2 and 3 ranks share the box's one GPU (rendezvous on
gloo, since RCCL refuses two ranks on one device), each generates its user block, the gathered tensor must be
bit-identical to the single-process result."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n_total", [(2, 101), (3, 50)])
def test_sharded_generation_equals_single_process(world, n_total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(n_total)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-3000:]}"
        assert f"rank {r} ok" in o


def test_bench_two_rank_rehearsal_with_gather_leg():
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), rehearsed with two ranks
    sharing this box's GPU over gloo (DMX_DIST_BACKEND=gloo: RCCL refuses two ranks on one device): the JSON line keeps
    the contract (whole-job value, max-over-ranks time) and carries the separately timed gather leg."""
    import json
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DMX_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "tiny", "--gather", "--gather-users", "64"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["users_total"] == 2 * 512
    assert d["value"] == pytest.approx(d["config"]["users_total"] / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert "cpu_baseline" not in d                                 # rank 0 at N = 1 only
    g = d["gather"]
    assert g["backend"] == "gloo" and g["device_tensors"] is False
    assert g["side_products"]["bytes_total"] == 2 * 4 * 1024 and g["side_products"]["ms"] > 0
    assert g["channel_slice_to_root"]["bytes_per_peer"] == 64 * 4 * 64 * 512 * 8 and g["channel_slice_to_root"]["GBps_per_peer"] > 0
