"""Worker of tests/test_dist_cpu.py: one process per rank, gloo backend, CPU tensors."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimo_amd import dist as ddist  # noqa: E402


def main():
    n_total = int(sys.argv[1])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    b, e = ddist.shard_bounds(n_total, world, rank)
    # stand-in for this rank's block of the channel tensor / side products: value encodes the user
    users = torch.arange(b, e, dtype=torch.float32)
    chan = torch.complex(users[:, None, None, None] * torch.ones(1, 2, 3, 4), -users[:, None, None, None] * torch.ones(1, 2, 3, 4))
    los = (torch.arange(b, e) % 3 - 1).to(torch.int32)
    full_los = ddist.all_gather_users(los, n_total)
    want_los = (torch.arange(n_total) % 3 - 1).to(torch.int32)
    assert torch.equal(full_los, want_los), (rank, full_los, want_los)
    got = ddist.gather_users_to_root(chan, n_total, dst=0)
    if rank == 0:
        assert got.shape == (n_total, 2, 3, 4)
        assert torch.equal(got.real[:, 0, 0, 0], torch.arange(n_total, dtype=torch.float32))
        assert torch.equal(got.imag[:, 1, 2, 3], -torch.arange(n_total, dtype=torch.float32))
    else:
        assert got is None
    # gather of a subcarrier slice into a caller-provided buffer on a non-zero root
    out = torch.empty((n_total, 2, 3, 2), dtype=torch.complex64) if rank == world - 1 else None
    got = ddist.gather_users_to_root(chan[..., ::2].contiguous(), n_total, dst=world - 1, out=out)
    if rank == world - 1:
        assert got is out and torch.equal(out.real[:, 0, 0, 1], torch.arange(n_total, dtype=torch.float32))
    # MacroDataset axis: three basestations of different sizes, (basestation, user) items cut over the ranks
    n_users = [n_total, 3, max(1, n_total // 2)]
    pieces = {}
    for i, ub, ue in ddist.macro_shard_plan(n_users, world, rank):
        uu = torch.arange(ub, ue, dtype=torch.float32)
        pieces[i] = torch.complex(uu[:, None, None, None] * torch.ones(1, 2, 3, 4) + 1000 * i, torch.zeros(ue - ub, 2, 3, 4))
    full = ddist.gather_macro_to_root(pieces, n_users, dst=0)
    if rank == 0:
        assert len(full) == 3
        for i, n in enumerate(n_users):
            assert full[i].shape == (n, 2, 3, 4)
            assert torch.equal(full[i].real[:, 1, 2, 3], torch.arange(n, dtype=torch.float32) + 1000 * i)
    else:
        assert full is None
    # fewer users than ranks: the root may hold no block at all.  With the trailing shape given the gather still works;
    # without it EVERY rank raises (none would post its half of the exchange) instead of the others hanging
    if world >= 2:
        tiny = [1]
        root = world - 1                                            # the single user lands on rank 0
        pieces = {i: torch.full((ue - ub, 2), 7.0) for i, ub, ue in ddist.macro_shard_plan(tiny, world, rank)}
        full = ddist.gather_macro_to_root(pieces, tiny, dst=root, trailing_shape=(2,), dtype=torch.float32)
        if rank == root:
            assert len(full) == 1 and torch.equal(full[0], torch.full((1, 2), 7.0))
        try:
            ddist.gather_macro_to_root(pieces, tiny, dst=root)
            raise AssertionError("expected ValueError on every rank")
        except ValueError as exc:
            assert "holds no user block" in str(exc)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == world
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
