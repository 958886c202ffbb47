"""Multi-GPU layer of the channel path: users are independent (channel.py:264 iterates users with
no cross-user state), so the path shards as contiguous user blocks, one process per GPU, with NO
collective on the data path.  RCCL (torch.distributed backend "nccl" on ROCm) is used only to
bring results together when the caller asks:

  * ``all_gather_users``   - small per-user side products (``los``, ``num_paths``; int arrays)
  * ``gather_users_to_root`` - a channel tensor (or a user / subcarrier slice of it) to one rank,
    as a fan-in of point-to-point transfers straight into the destination rows.  xGMI is
    point-to-point (7 links x ~153 GB/s per GPU): a root ingests on all its links at once, while a
    ring gather would be bound by one link, so no ring collective is used here.

The full tensors of the big configurations do not fit one GPU (1M users x 64x4 x 512 = 1.05 TB),
so the default is sharded-resident output: each rank keeps its block (``ShardResult.channel``).
Everything here also runs on the ``gloo`` backend with CPU tensors (that is how the CPU tests
cover the N > 1 path); only ``compute_channels_sharded`` needs a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import consts as c


def shard_bounds(n_ue: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition: rank r owns users [r*ceil(N/G), min(N, (r+1)*ceil(N/G)))."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    per = -(-n_ue // world) if n_ue > 0 else 0
    b = min(n_ue, rank * per)
    return b, min(n_ue, b + per)


def shard_sizes(n_ue: int, world: int) -> List[int]:
    return [e - b for b, e in (shard_bounds(n_ue, world, r) for r in range(world))]


def _dist():
    import torch.distributed as dist
    return dist


def _world(group=None) -> Tuple[int, int]:
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def all_gather_users(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Concatenate per-user shards (dim 0) from all ranks on every rank.  Shards follow
    shard_bounds, so they may be ragged: they are padded to the largest shard for the collective."""
    rank, world = _world(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: local shard has {local.shape[0]} users, expected {sizes[rank]}")
    if world == 1:
        return local
    dist = _dist()
    per = max(sizes)
    buf = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def gather_users_to_root(local: torch.Tensor, n_total: int, dst: int = 0, group=None,
                         out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """Fan-in gather of per-user shards to rank `dst`: every peer sends its block once, the root
    receives each block directly into its rows of the result (no staging copy, no ring)."""
    rank, world = _world(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: local shard has {local.shape[0]} users, expected {sizes[rank]}")
    local = local.contiguous()
    if world == 1:
        if out is not None:
            out.copy_(local)
            return out
        return local
    dist = _dist()
    # one batch of point-to-point operations per rank: on RCCL the root's receives from all peers are grouped and
    # progress concurrently (one direct xGMI link per peer), instead of one blocking receive after the other
    ops = []
    if rank == dst:
        if out is None:
            out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        for r in range(world):
            b, e = shard_bounds(n_total, world, r)
            if e == b:
                continue
            if r == dst:
                out[b:e].copy_(local)
            else:
                ops.append(dist.P2POp(dist.irecv, out[b:e], r, group))
    elif local.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, local, dst, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return out if rank == dst else None


def macro_shard_plan(n_users: List[int], world: int, rank: int) -> List[Tuple[int, int, int]]:
    """(basestation, user_begin, user_end) items of rank `rank` for a MacroDataset whose children hold
    ``n_users[i]`` users each (deepmimo/generator/dataset.py:947-950 runs them one after the other): the children's
    users are laid end to end and that line is cut into `world` contiguous blocks - both axes are independent, so a
    rank's block may cover the tail of one basestation and the head of the next.  Every (basestation, user) lands on
    exactly one rank, loads differ by at most one user."""
    total = int(sum(n_users))
    gb, ge = shard_bounds(total, world, rank)
    plan, start = [], 0
    for i, n in enumerate(n_users):
        b, e = max(gb, start), min(ge, start + n)
        if e > b:
            plan.append((i, b - start, e - start))
        start += n
    return plan


def gather_macro_to_root(pieces: Dict[int, torch.Tensor], n_users: List[int], dst: int = 0, group=None,
                         trailing_shape=None, dtype=None, device=None) -> Optional[List[torch.Tensor]]:
    """Bring a MacroDataset's sharded results together on rank `dst`: ``pieces[i]`` is this rank's block of
    basestation i (as planned by ``macro_shard_plan``; absent when the rank holds none of it).  Returns the list of
    full per-basestation tensors on `dst` (what MacroDataset.compute_channels returns in one process), None elsewhere.
    Point-to-point fan-in, one batch per rank, as in ``gather_users_to_root``.  The receive buffers take their trailing
    shape / dtype / device from `trailing_shape`, `dtype`, `device` when given, else from a piece `dst` holds itself;
    when `dst` holds none (fewer users than ranks) and they are not given, EVERY rank raises - the other ranks would
    otherwise wait for a receive that is never posted."""
    rank, world = _world(group)
    plans = [macro_shard_plan(n_users, world, r) for r in range(world)]
    mine = {i: (b, e) for i, b, e in plans[rank]}
    for i, (b, e) in mine.items():
        if i not in pieces or pieces[i].shape[0] != e - b:
            raise ValueError(f"rank {rank}: basestation {i} block must hold {e - b} users")
    if world == 1:
        return [pieces[i] for i in range(len(n_users))]
    dist = _dist()
    if (trailing_shape is None or dtype is None) and not plans[dst]:
        # the same condition on every rank (the plan is a pure function of n_users and the world size): nobody posts anything
        raise ValueError(f"gather_macro_to_root: rank {dst} holds no user block (users per basestation {list(n_users)}, "
                         f"{world} ranks), so the receive buffers' trailing shape and dtype must be passed explicitly")
    ops, outs = [], None
    if rank == dst:
        ref = next(iter(pieces.values())) if pieces else None
        tshape = tuple(trailing_shape) if trailing_shape is not None else tuple(ref.shape[1:])
        dt = dtype if dtype is not None else ref.dtype
        dev = device if device is not None else (ref.device if ref is not None else torch.device("cpu"))
        outs = [torch.empty((n,) + tshape, dtype=dt, device=dev) for n in n_users]
        for r in range(world):
            for i, b, e in plans[r]:
                if r == dst:
                    outs[i][b:e].copy_(pieces[i])
                else:
                    ops.append(dist.P2POp(dist.irecv, outs[i][b:e], r, group))
    else:
        for i, b, e in plans[rank]:
            ops.append(dist.P2POp(dist.isend, pieces[i].contiguous(), dst, group))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return outs


@dataclass
class ShardResult:
    channel: torch.Tensor            # complex64 [n_local, M_rx, M_tx, K or P], resident on this rank's GPU
    user_begin: int
    user_end: int
    n_total: int
    side: Dict[str, torch.Tensor]    # this rank's los / num_paths / ... (device tensors)


def compute_channels_sharded(data, params, bs_fov=None, ue_fov=None, group=None, device_index: Optional[int] = None,
                             want_side: bool = True, variant: int = 0, user_range: Optional[Tuple[int, int]] = None):
    """Each rank generates the channels of its own user block on its own GPU.

    ``data`` may also be a MacroDataset or a list of per-basestation mappings: the (basestation, user) items are then
    partitioned by ``macro_shard_plan`` and the result is ``{basestation: ShardResult}`` for the blocks this rank owns
    (``gather_macro_to_root`` assembles the per-basestation list).  ``user_range`` overrides the rank's block.

    data: mapping with the full float32 [N, L] ray matrices (every rank may hold the full host
    copy - 800 B per user - or a torch tensor); only the local rows are uploaded.  params:
    ChannelGenParameters (validated by the caller).  A per-user UE rotation [N, 3] is sliced like
    the rays; the random-range form (3, 2) must be resolved by the caller so that all ranks agree."""
    from .engine import ChannelEngine
    rank, world = _world(group)
    if device_index is None:
        device_index = torch.cuda.current_device()
    from .dataset import MacroDataset
    children = data.datasets if isinstance(data, MacroDataset) else (list(data) if isinstance(data, (list, tuple)) else None)
    if children is not None:
        n_users = [int(d[c.POWER_PARAM_NAME].shape[0]) for d in children]
        return {i: compute_channels_sharded(children[i], params, bs_fov=bs_fov, ue_fov=ue_fov, group=group,
                                            device_index=device_index, want_side=want_side, variant=variant,
                                            user_range=(b, e))
                for i, b, e in macro_shard_plan(n_users, world, rank)}
    n_total = int(data[c.POWER_PARAM_NAME].shape[0])
    b, e = shard_bounds(n_total, world, rank) if user_range is None else user_range
    eng = ChannelEngine(device_index)
    local = {k: data[k][b:e] for k in c.RAY_FIELDS}
    for k in (c.DOPPLER_VEL_PARAM_NAME, c.DOPPLER_ACC_PARAM_NAME):
        if hasattr(data, "keys") and k in data.keys():
            local[k] = data[k][b:e]
    rays = eng.upload_rays(local)
    rot = np.asarray(params[c.PARAMSET_ANT_UE][c.PARAMSET_ANT_ROTATION])
    rot_pu = None
    if rot.ndim == 2:
        if rot.shape == (3, 2) and n_total != 3:
            raise ValueError("resolve the random UE rotation range to [N, 3] before sharding")
        rot_pu = rot[b:e]
    carrier = 0.0
    if hasattr(data, "keys") and c.RT_PARAMS_PARAM_NAME in data.keys():
        carrier = float(data[c.RT_PARAMS_PARAM_NAME].get(c.RT_PARAM_FREQUENCY, 0.0))
    prep = eng.prepare(rays, params, bs_fov=bs_fov, ue_fov=ue_fov, ue_rotation_per_user=rot_pu,
                       carrier_freq=carrier, want_side=want_side)
    chan = eng.channels(prep, variant=variant)
    side = {k: v for k, v in prep.side.items() if v is not None and k != "max_delay_key"}
    return ShardResult(channel=chan, user_begin=b, user_end=e, n_total=n_total, side=side)
