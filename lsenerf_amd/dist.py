"""Data-parallel exchange for the hot path: one process per GPU, rays sharded across ranks, ONE collective per step.

The reference wraps the model in torch DDP (R:lse_nerf/lse_pipeline.py:95-98): bucketed all-reduce of every gradient
(including a dead 67 MB table) plus a broadcast of all module buffers on each of its 3 forwards per step.  Here the
whole gradient is one flat fp32 buffer (lsenerf_amd.optim.FlatParams), so the exchange is a single sum-all-reduce
over RCCL/xGMI (backend "nccl" on ROCm), and the DDP mean is folded into the Adam kernel's ``grad_scale = 1/W``.
Occupancy grids stay replicated by updating them from a rank-independent RNG stream (no per-forward broadcast).

Backend-agnostic: the same code runs on gloo/CPU tensors, which is how the world_size-2 tests exercise it.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun contract).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_rays(n_rays_global: int, rank: int, world: int) -> slice:
    """Rank r takes rays [r*R/W, (r+1)*R/W)."""
    per = n_rays_global // world
    return slice(rank * per, (rank + 1) * per)


def allreduce_grads(flat_grad: torch.Tensor, async_op: bool = False):
    """Sum-all-reduce of the flat gradient buffer; returns the work handle when async."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)


def broadcast_params(flat_data: torch.Tensor, src: int = 0):
    """Initial parameter replication (DDP's constructor broadcast)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_data, src=src)


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
