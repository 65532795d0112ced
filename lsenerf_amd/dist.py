"""Data-parallel exchange for the hot path: one process per GPU, rays sharded across ranks, ONE collective per step.

The reference wraps the model in torch DDP (R:lse_nerf/lse_pipeline.py:95-98): bucketed all-reduce of every gradient
(including a dead 67 MB table) plus a broadcast of all module buffers on each of its 3 forwards per step.  Here the
whole gradient is one flat fp32 buffer (lsenerf_amd.optim.FlatParams), so the exchange is a single sum-all-reduce
over RCCL/xGMI (backend "nccl" on ROCm), and the DDP mean is folded into the Adam kernel's ``grad_scale = 1/W``.
Occupancy grids stay replicated without DDP's per-forward buffer broadcast: ``LSEOccGridEstimator`` draws its update
cells and jitter from its OWN generator, re-seeded from ``(base seed, step)`` at every update, so every rank (whatever
its global seed, R:train.py:104 seeds by rank) refreshes the same cells with the same positions from the same
parameters; ``check_grid_consistency`` asserts it (rehearsed in tools/dp_rehearsal.py and tests/test_dist_cpu.py).
"The same parameters give the same densities" holds to the last bit with one process per GPU (the production layout; every
refresh of the round-3 rehearsal agrees across ranks before any broadcast).  ``attach_grid_sync`` additionally broadcasts rank 0's
``occs`` / ``binaries`` after every refresh (33.5 + 8.4 MB once per 16 steps: what DDP's buffer broadcast does on every forward,
R:lse_nerf/lse_pipeline.py:97), so that a divergence -- round 2 saw one with two processes' kernels interleaved on ONE card,
DESIGN.md section 7 -- cannot persist.

Backend-agnostic: the same code runs on gloo/CPU tensors, which is how the world_size-2 tests exercise it.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

# Run every collective of this module even in a ONE-rank process group.  With one rank each of them is the identity, so the results
# must equal the path without torch.distributed bit for bit -- which is how the RCCL calls themselves (argument types, views into the
# flat buffers, work handles, the bool grid through a uint8 view) are executed on a one-GPU box, where two RCCL ranks cannot share
# the card (tools/dp_rehearsal.py --backend nccl, tests/test_gpu_dp.py).  Off in production: a single GPU needs no process group.
SINGLE_RANK_COLLECTIVES = False


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun contract).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or SINGLE_RANK_COLLECTIVES) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def _active() -> bool:
    """True when the collectives of this module run: a process group exists and has more than one rank (or
    SINGLE_RANK_COLLECTIVES asks for them in a one-rank group)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or SINGLE_RANK_COLLECTIVES)


def shard_rays(n_rays_global: int, rank: int, world: int) -> slice:
    """Rank r takes rays [r*R/W, (r+1)*R/W)."""
    assert n_rays_global % world == 0, f"{n_rays_global} rays do not split evenly over {world} ranks"
    per = n_rays_global // world
    return slice(rank * per, (rank + 1) * per)


def allreduce_grads(flat_grad: torch.Tensor, async_op: bool = False):
    """Sum-all-reduce of the flat gradient buffer; returns the work handle when async."""
    if not _active():
        return None
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)


class OverlappedGradExchange:
    """The one all-reduce of the step, cut in two so that most of it hides behind the tail of the backward pass.

    48.8 of the 48.9 MB of gradients belong to the hash table, and the hash backward is the LAST kernel of the backward
    pass, so a plain all-reduce has nothing to overlap with.  With this object installed the hash backward runs in two
    launches (ops.set_hash_bwd_hook(table, "split", ...)): levels >= ``split_level`` first -- at the default split 8 x 4 MB of the table -- whose
    slice of the flat gradient buffer is handed to an asynchronous all-reduce (RCCL runs it on its own stream) while the
    second launch computes the remaining levels; ``finish()`` then reduces the rest and waits.  Results are identical to
    ``allreduce_grads`` (same sum over ranks, element for element).  Requires optim.FlatParams (gradients accumulate in
    place in one buffer).

    Steps with SEVERAL hash backwards (the event configs run three field passes per step -- colour, previous and next
    event bundle -- hence three hash backwards per ``loss.backward()``, all accumulating into the same table gradient):
    call ``begin_step(n_backwards)`` before ``backward()``; the early all-reduce is issued only after the first launch of
    the LAST hash backward, when the fine levels' gradients are final.  Without ``begin_step`` one hash backward per
    step is assumed, and a second callback in the same step raises instead of reducing a slice twice."""

    def __init__(self, flat, table_param: torch.nn.Parameter, level_offsets, split_level: int):
        idx = [i for i, p in enumerate(flat.params) if p is table_param]
        assert idx, "table parameter is not part of the FlatParams"
        base = flat.offsets[idx[0]]
        self.flat, self.table_param = flat, table_param
        self.split_level = int(split_level)
        # table layout: level l occupies entries [offsets[l], offsets[l+1]) with 2 features each
        self.lo = base + 2 * int(level_offsets[split_level])
        self.hi = base + table_param.numel()
        self._works = []
        self._expected, self._seen = 1, 0

    def begin_step(self, n_backwards: int = 1):
        """Arm the exchange for a step whose backward pass runs ``n_backwards`` hash backwards."""
        assert n_backwards >= 1
        assert not self._works, "begin_step() while the previous step's exchange is still in flight: call finish() first"
        self._expected, self._seen = int(n_backwards), 0

    def install(self):
        """Hook the hash backward of THIS table (ops.set_hash_bwd_hook: keyed by the table, so other models of the process --
        an evaluation copy, a viewer thread -- keep their single-launch backward)."""
        from . import ops
        ops.set_hash_bwd_hook(self.table_param, "split", (self.split_level, self._first_part_ready))

    def uninstall(self):
        from . import ops
        ops.set_hash_bwd_hook(self.table_param, "split", None)

    def _first_part_ready(self):
        if self._seen >= self._expected:
            raise RuntimeError(f"OverlappedGradExchange: hash backward #{self._seen + 1} in a step armed for "
                               f"{self._expected}; call begin_step(n_backwards) before backward()")
        self._seen += 1
        if self._seen < self._expected:
            return          # later hash backwards still add to this slice
        if _active():
            self._works.append(dist.all_reduce(self.flat.grad[self.lo:self.hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Call after backward(): exchanges what the first part did not cover and waits for everything."""
        seen, self._seen = self._seen, 0
        if not _active():
            self._works.clear()
            return
        g = self.flat.grad
        assert seen in (0, self._expected), f"{seen} hash backwards ran in a step armed for {self._expected}"
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


class ShardedAdamExchange:
    """reduce-scatter -> per-shard Adam -> all-gather: every rank reduces and owns 1/W of the flat
    gradient buffer, keeps Adam moments only for that shard (1/W of the optimizer state) and updates only that shard of the
    parameters; one all-gather then replicates the updated parameters.  Same wire volume as one all-reduce, 1/W of the
    optimizer work and state per rank, and bit-identical parameters on every rank by construction.

    The shards are 64-float aligned.  Build the buffers with ``FlatParams(params, total_multiple=W * 64)`` and the
    collectives run on them in place; otherwise persistent padded staging buffers (allocated once here, never per step) are
    used.  Backends without reduce_scatter (gloo) fall back to all-reduce + slicing, which is what the world-2 CPU test
    exercises; on RCCL the native collectives run."""

    def __init__(self, flat, lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-15, adam_fn=None):
        self.flat = flat
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        n = flat.data.numel()
        per = (n + self.world * 64 - 1) // (self.world * 64) * 64
        self.per, self.padded = per, per * self.world
        dev = flat.data.device
        self._pad = self.padded - n
        self.grad_shard = torch.zeros(per, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(per, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(per, dtype=torch.float32, device=dev)
        self.param_full = torch.zeros(self.padded, dtype=torch.float32, device=dev) if self._pad else None
        self.grad_full = torch.zeros(self.padded, dtype=torch.float32, device=dev) if self._pad else None   # pad stays 0
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        if adam_fn is None:             # the HIP Adam kernel (lse_adam_step)
            from . import ops
            adam_fn = ops.adam_step
        self.adam_fn = adam_fn          # (param, grad, m, v, lr, b1, b2, eps, step, grad_scale) -> None, in place

    def _native(self) -> bool:
        return _active() and dist.get_backend() == "nccl"

    def step(self):
        """Call after backward(): exchanges gradients, updates this rank's shard, replicates the parameters."""
        self.start()
        self.finish()

    def start(self):
        """First half: launch the gradient exchange (reduce-scatter; all-reduce on backends without one) asynchronously.
        ``finish`` waits for it, runs Adam on this rank's shard and all-gathers the parameters -- dist.GradPipeline puts the
        next step's ray marcher between the two."""
        assert getattr(self, "_work", None) is None and not getattr(self, "_started", False), "start() twice without finish()"
        g = self.flat.grad
        if self._pad:
            gp = self.grad_full
            gp[: g.numel()].copy_(g)
        else:
            gp = g
        self._gp, self._work = gp, None
        if _active():
            if self._native():
                self._work = dist.reduce_scatter_tensor(self.grad_shard, gp, op=dist.ReduceOp.SUM, async_op=True)
            else:
                self._work = dist.all_reduce(gp, op=dist.ReduceOp.SUM, async_op=True)
        self._started = True

    def finish(self):
        assert getattr(self, "_started", False), "finish() without start()"
        p, gp = self.flat.data, self._gp
        lo, hi = self.rank * self.per, (self.rank + 1) * self.per
        if self._work is not None:
            self._work.wait()
        if not self._native():
            self.grad_shard.copy_(gp[lo:hi])
        self._work, self._started = None, False
        pfull = self.param_full if self._pad else p
        if self._pad:
            pfull[: p.numel()].copy_(p)
        shard = pfull[lo:hi]
        self.step_count += 1
        self.adam_fn(shard, self.grad_shard, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                     self.step_count, 1.0 / self.world)
        if _active():
            if self._native():
                dist.all_gather_into_tensor(pfull, shard.clone())
            else:
                parts = [torch.empty_like(shard) for _ in range(self.world)]
                dist.all_gather(parts, shard.clone())
                for r, t in enumerate(parts):
                    pfull[r * self.per:(r + 1) * self.per].copy_(t)
        if self._pad:
            p.copy_(pfull[: p.numel()])


class GradPipeline:
    """N > 1: the one all-reduce of step k runs asynchronously (RCCL's own stream) while step k+1 marches its rays.

    The ray marcher reads the occupancy grid only -- no parameters, no gradients -- so ``all-reduce(k) -> Adam(k)`` can
    be finished AFTER the marcher of step k+1 without changing a single value: everything that reads parameters in step
    k+1 (the visibility pre-pass ``sigma_fn`` when ``alpha_thre > 0``, the field pass, the occupancy refresh) still sees
    the parameters updated by step k.  ``attach(estimator)`` installs ``flush`` as the estimator's ``after_march_hook``,
    which ``LSEOccGridEstimator.sampling`` calls right after the marcher and BEFORE ``sigma_fn``; callers that drive the
    sampler themselves call ``flush()`` at that point.  ``flush()`` is idempotent; call it before anything else that reads
    the parameters (grid refresh, evaluation, checkpoint) and at the end of a timed region, so that exactly K optimizer
    steps and K all-reduces belong to K steps."""

    def __init__(self, opt, world: Optional[int] = None, sharded: Optional["ShardedAdamExchange"] = None):
        """``sharded``: finish the step with ``ShardedAdamExchange`` (reduce-scatter -> Adam on 1/W -> all-gather) instead of the
        all-reduce + full Adam of ``opt``; ``opt`` then only supplies the learning-rate schedule and the step count."""
        self.opt = opt
        self.world = world if world is not None else world_size()
        self.sharded = sharded
        self.work, self.pending = None, False
        # measurement hook: a list makes flush() bracket the wait for the collective with two events on the current stream; the time
        # between them is what the exchange still costs the step AFTER the overlap (bench.py: exchange_exposed_ms_per_step)
        self.exposed_events: Optional[list] = None

    def attach(self, estimator):
        estimator.after_march_hook = self.flush
        return self

    def start(self):
        """Call after backward(): launches the asynchronous all-reduce of the flat gradient buffer."""
        assert not self.pending, "GradPipeline.start() twice without flush()"
        if self.sharded is not None:
            self.sharded.lr = self.opt.current_lr()
            self.sharded.start()
        else:
            self.work = allreduce_grads(self.opt.flat.grad, async_op=True)
        self.pending = True

    def flush(self):
        """wait -> Adam (mean over ranks folded into grad_scale).  No-op when nothing is in flight."""
        if self.pending:
            ev = None
            if self.exposed_events is not None and torch.cuda.is_available():
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self.sharded is not None:
                self.opt.step_count += 1
                self.sharded.finish()      # (wait + Adam on 1/W + all-gather: the bracket then holds all three)
            else:
                if self.work is not None:
                    self.work.wait()
            if ev is not None:
                ev[1].record()
                self.exposed_events.append(ev)
            if self.sharded is None:
                self.opt.step(grad_scale=1.0 / self.world)
            self.work, self.pending = None, False


def check_grid_consistency(estimator) -> bool:
    """True iff ``occs`` and ``binaries`` are bit-identical on every rank (the invariant DDP's buffer broadcast provides
    in the reference, R:lse_nerf/lse_pipeline.py:97)."""
    if not _active():
        return True
    ok = True
    for buf in (estimator.occs, estimator.binaries.to(torch.uint8)):
        ref = buf.clone()
        dist.broadcast(ref, src=0)
        same = torch.tensor([1 if torch.equal(ref, buf) else 0], dtype=torch.int32, device=buf.device)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        ok = ok and bool(same.item())
    return ok


def sync_grid(estimator, src: int = 0) -> None:
    """Rank ``src``'s occupancy state replaces every other rank's (``occs`` fp32, ``binaries`` bool)."""
    if not _active():
        return
    dist.broadcast(estimator.occs, src=src)
    b8 = estimator.binaries.view(torch.uint8) if estimator.binaries.dtype == torch.bool else estimator.binaries
    dist.broadcast(b8, src=src)
    if hasattr(estimator, "_bump_grid_version"):
        estimator._bump_grid_version()
    if hasattr(estimator, "_invalidate_occ_mean"):       # host / device copies of occs.mean() (the cap of the alpha threshold)
        estimator._invalidate_occ_mean()
    elif hasattr(estimator, "_occ_mean_host"):
        estimator._occ_mean_host = None


def attach_grid_sync(estimator, src: int = 0):
    """Broadcast the grid from rank ``src`` after every refresh of ``estimator`` (its ``after_update_hook``)."""
    estimator.after_update_hook = lambda: sync_grid(estimator, src)
    return estimator


def broadcast_params(flat_data: torch.Tensor, src: int = 0):
    """Initial parameter replication (DDP's constructor broadcast)."""
    if _active():
        dist.broadcast(flat_data, src=src)


def max_over_ranks(value: float, device) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
