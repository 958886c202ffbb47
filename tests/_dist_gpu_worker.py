"""Worker of tests/test_gpu_dist.py: ranks share the box's GPU, rendezvous over gloo; each generates its
user block with the real kernels (deepmimo_amd.dist.compute_channels_sharded), the root gathers the blocks and
compares with the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepmimo_amd as dm  # noqa: E402
from deepmimo_amd import dist as ddist  # noqa: E402
from oracle import oracle_np as onp  # noqa: E402


def main():
    n_total = int(sys.argv[1])
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    rays = onp.synth_rays(n_total, 12, seed=77)                  # every rank holds the (tiny) full host copy
    p = dm.ChannelGenParameters()
    p.bs_antenna.shape = np.array([8, 4])
    p.ue_antenna.shape = np.array([2, 1])
    p.ue_antenna.rotation = np.random.default_rng(3).uniform(0, 30, (n_total, 3))   # per-user rotation is sliced too
    p.ofdm.selected_subcarriers = np.arange(0, 512, 16)
    p.validate(n_total)
    res = ddist.compute_channels_sharded(rays, p, bs_fov=np.array([150, 120]), device_index=0)
    b, e = ddist.shard_bounds(n_total, world, rank)
    assert (res.user_begin, res.user_end) == (b, e) and res.channel.shape[0] == e - b
    full = ddist.gather_users_to_root(res.channel.cpu(), n_total, dst=0)
    los = ddist.all_gather_users(res.side["los"].cpu(), n_total)
    if rank == 0:
        ds = dm.Dataset(dict(rays))
        ds.apply_fov(bs_fov=np.array([150, 120]))
        H = ds.compute_channels(p)
        assert np.array_equal(full.numpy(), H), "sharded result differs from the single-process result"
        assert np.array_equal(los.numpy(), ds.los)
    # MacroDataset x users: two basestations, (basestation, user-block) items partitioned over the ranks
    rays2 = onp.synth_rays(max(3, n_total // 3), 12, seed=78)
    p2 = dm.ChannelGenParameters()
    p2.bs_antenna.shape = np.array([8, 4])
    p2.ofdm.selected_subcarriers = np.arange(0, 512, 16)
    md = dm.MacroDataset([dm.Dataset(dict(rays)), dm.Dataset(dict(rays2))])
    n_users = [n_total, rays2["power"].shape[0]]
    res = ddist.compute_channels_sharded(md, p2.validate(n_total), device_index=0)
    assert sorted(res) == [i for i, _, _ in ddist.macro_shard_plan(n_users, world, rank)]
    full = ddist.gather_macro_to_root({i: r.channel.cpu() for i, r in res.items()}, n_users, dst=0)
    if rank == 0:
        want = md.compute_channels(p2)                                  # single-process fan-out: a list
        assert len(full) == len(want) == 2
        for got, w in zip(full, want):
            assert np.array_equal(got.numpy(), w), "sharded MacroDataset differs from the single-process fan-out"
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} ok")


if __name__ == "__main__":
    main()
