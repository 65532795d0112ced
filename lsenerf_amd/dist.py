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


class OverlappedGradExchange:
    """The one all-reduce of the step, cut in two so that most of it hides behind the tail of the backward pass.

    48.8 of the 48.9 MB of gradients belong to the hash table, and the hash backward is the LAST kernel of the backward
    pass, so a plain all-reduce has nothing to overlap with.  With this object installed the hash backward runs in two
    launches (ops.HASH_BWD_SPLIT): levels >= ``split_level`` first -- at the default split 8 x 4 MB of the table -- whose
    slice of the flat gradient buffer is handed to an asynchronous all-reduce (RCCL runs it on its own stream) while the
    second launch computes the remaining levels; ``finish()`` then reduces the rest and waits.  Results are identical to
    ``allreduce_grads`` (same sum over ranks, element for element).  Requires optim.FlatParams (gradients accumulate in
    place in one buffer)."""

    def __init__(self, flat, table_param: torch.nn.Parameter, level_offsets, split_level: int):
        idx = [i for i, p in enumerate(flat.params) if p is table_param]
        assert idx, "table parameter is not part of the FlatParams"
        base = flat.offsets[idx[0]]
        self.flat = flat
        self.split_level = int(split_level)
        # table layout: level l occupies entries [offsets[l], offsets[l+1]) with 2 features each
        self.lo = base + 2 * int(level_offsets[split_level])
        self.hi = base + table_param.numel()
        self._works = []

    def install(self):
        from . import ops
        ops.HASH_BWD_SPLIT = (self.split_level, self._first_part_ready)

    def uninstall(self):
        from . import ops
        ops.HASH_BWD_SPLIT = None

    def _first_part_ready(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            self._works.append(dist.all_reduce(self.flat.grad[self.lo:self.hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Call after backward(): exchanges what the first part did not cover and waits for everything."""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            self._works.clear()
            return
        g = self.flat.grad
        if not self._works:      # the split did not trigger (e.g. no hash backward ran): one plain all-reduce
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            return
        if self.lo > 0:
            self._works.append(dist.all_reduce(g[:self.lo], op=dist.ReduceOp.SUM, async_op=True))
        if self.hi < g.numel():
            self._works.append(dist.all_reduce(g[self.hi:], op=dist.ReduceOp.SUM, async_op=True))
        for w in self._works:
            w.wait()
        self._works.clear()


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
