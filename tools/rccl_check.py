#!/usr/bin/env python3
"""Sanity check of the RCCL plumbing bench.py relies on for N > 1 (one rank is enough to exercise
init_process_group(backend='nccl', device_id=...), barrier and a MAX all-reduce on the GPU)."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)
dist.barrier()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
x = torch.arange(8, device=dev, dtype=torch.float32)
parts = [torch.empty_like(x)]
dist.all_gather(parts, x)
print("rccl ok", t.tolist(), dist.get_backend(), parts[0].sum().item())
dist.destroy_process_group()
