"""CPU coverage of the N > 1 path: user-block partition + the gathers, world_size 2 and 3 on gloo."""
import os
import socket
import subprocess
import sys

import pytest

from deepmimo_amd import dist as ddist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 7, 8, 100, 100_000, 1_000_003):
        for g in (1, 2, 3, 4, 8):
            bounds = [ddist.shard_bounds(n, g, r) for r in range(g)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            for (b0, e0), (b1, e1) in zip(bounds, bounds[1:]):
                assert e0 == b1 and b0 <= e0
            sizes = ddist.shard_sizes(n, g)
            assert sum(sizes) == n and max(sizes) - min(s for s in sizes) <= -(-n // g)
    assert ddist.shard_bounds(10, 4, 3) == (9, 10)      # ceil(10/4) = 3 per rank
    with pytest.raises(ValueError):
        ddist.shard_bounds(10, 4, 4)


def test_macro_shard_plan_partitions_every_basestation_user_once():
    for n_users in ([5], [4, 4], [10, 1, 7], [0, 3, 0, 9], [100_000, 100_000, 31_931]):
        for g in (1, 2, 3, 8):
            seen = [[0] * n for n in n_users]
            loads = []
            for r in range(g):
                plan = ddist.macro_shard_plan(n_users, g, r)
                loads.append(sum(e - b for _, b, e in plan))
                assert plan == sorted(plan)
                for i, b, e in plan:
                    assert 0 <= b < e <= n_users[i]
                    if n_users[i] < 1000:
                        for u in range(b, e):
                            seen[i][u] += 1
            assert sum(loads) == sum(n_users) and max(loads) - min(loads) <= -(-sum(n_users) // g)
            assert all(v == 1 for row in seen if len(row) < 1000 for v in row)
    assert ddist.macro_shard_plan([4, 4], 2, 0) == [(0, 0, 4)] and ddist.macro_shard_plan([4, 4], 2, 1) == [(1, 0, 4)]
    assert ddist.macro_shard_plan([3, 5], 2, 0) == [(0, 0, 3), (1, 0, 1)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_total", [(2, 11), (3, 7), (2, 1)])
def test_gathers_on_gloo(world, n_total):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(n_total)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o
