"""The whole training step as ONE replayed HIP graph.

In the configuration LSENeRF trains in (3510 rays per step, ~260 samples per ray after culling) the kernels of a step take
2.8 ms, but issuing them from Python -- ~26 C-ABI launches, a dozen autograd nodes each way, ~40 allocations -- takes 3.3 ms
of host time, and the two read-backs of the sampler's sample counts stop the host from running ahead of the GPU
(bench.py ``cfg2_composition``: 3.6 - 4.5 ms per step depending on the box's CPU).  The reference pays the same kind of cost
in nerfstudio's trainer.  With the counts kept on the device (``LSENeRFModel.deferred_counts``, lse_set_device_count) nothing in
a step depends on the host any more, so the step -- sampler, visibility pre-pass, field, volume rendering, loss epilogue,
backward, Adam -- is captured once into a HIP graph (``torch.cuda.graph``: the C-ABI launches go to torch's current stream,
which is the capturing stream) and replayed with one call per step.

What varies from step to step enters through device memory at fixed addresses:
  * the rays and targets of the step   -> copied into the static input tensors before the replay;
  * the stratified jitter              -> ``torch.rand`` inside the graph (graph-safe Philox offsets), or a static input;
  * the learning rate / bias corrections -> three floats staged by ``FlatAdam.prepare_step`` (lse_adam_step_dev);
  * the occupancy grid and ``occs.mean()`` (the cap of the alpha threshold) -> refreshed in place, outside the graph, by
    ``LSENeRFModel.update_occupancy_grid`` between replays (lse_visibility_mask_cap reads the mean on the device).
Everything else (step size, cone angle, shapes, capacities) is constant for a given model and batch composition.
The values are those of the eager step: the same kernels run on the same samples (tests/test_gpu_graph.py).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .optim import FlatAdam
from .rays import RayBundle


def _static_like(b: Optional[RayBundle], requires_grad: bool) -> Optional[RayBundle]:
    if b is None:
        return None
    c = lambda t: None if t is None else t.detach().clone()
    return RayBundle(origins=c(b.origins).requires_grad_(requires_grad), directions=c(b.directions).requires_grad_(requires_grad),
                     pixel_area=c(b.pixel_area), camera_indices=c(b.camera_indices), nears=c(b.nears), fars=c(b.fars),
                     times=c(b.times), metadata={k: c(v) for k, v in b.metadata.items()})


def _copy_bundle(dst: Optional[RayBundle], src: Optional[RayBundle]) -> None:
    if (dst is None) != (src is None):
        raise ValueError("the batch composition of a captured step is fixed: a bundle appeared or disappeared")
    if dst is None:
        return
    if len(dst) != len(src):
        raise ValueError(f"the batch composition of a captured step is fixed: {len(dst)} rays captured, {len(src)} given")
    with torch.no_grad():
        for name in ("origins", "directions", "pixel_area", "camera_indices", "nears", "fars", "times"):
            d, s = getattr(dst, name), getattr(src, name)
            if (d is None) != (s is None):
                raise ValueError(f"RayBundle.{name} was {'set' if d is not None else 'absent'} at capture")
            if d is not None:
                d.copy_(s, non_blocking=True)
        if set(dst.metadata) != set(src.metadata):
            raise ValueError("RayBundle.metadata keys differ from the captured step's")
        for k, v in dst.metadata.items():
            v.copy_(src.metadata[k], non_blocking=True)


def capture_body(body, opt: Optional[FlatAdam], estimator, warmup: int = 3):
    """Capture ``body()`` -- a zero-argument callable that runs one step on tensors at fixed addresses and synchronises with
    nothing -- into a HIP graph.  Warm-up runs on a side stream as torch.cuda.graph requires (allocator pools, lazy
    initialisation); the optimizer state those eager runs change is saved and restored, so capturing trains nothing.
    Returns (graph, overflow flags of the captured marcher calls)."""
    dev = opt.flat.data.device if opt is not None else estimator.occs.device
    estimator._occ_mean_device()                          # allocate / refresh the device-side alpha cap before the capture
    saved = (opt.flat.data.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count) if opt is not None else None
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            if opt is not None:
                opt.prepare_step()
            body()
    torch.cuda.current_stream(dev).wait_stream(side)
    estimator.check_deferred_overflow()                   # (the eager warm-up runs' flags; one synchronisation, at construction)
    graph = torch.cuda.CUDAGraph()
    if opt is not None:
        opt.prepare_step()
    with torch.cuda.graph(graph):
        body()
    # the capture records launches, it does not run them: the overflow flag of a captured marcher call means something only
    # after a replay -- it is handed to the caller instead of staying in the estimator's list
    flags = estimator.__dict__.get("_deferred_flags", [])
    captured_flags = list(flags)
    del flags[:]
    if opt is not None:
        with torch.no_grad():
            opt.flat.data.copy_(saved[0]); opt.exp_avg.copy_(saved[1]); opt.exp_avg_sq.copy_(saved[2])
        opt.step_count = saved[3]
    return graph, captured_flags


class GraphedTrainStep:
    """``model.train_step_bundles`` + ``backward`` + ``FlatAdam`` step, captured once and replayed.

        step = GraphedTrainStep(model, opt, col, prev, nxt, batch)      # example inputs fix the composition
        for it in range(...):
            model.update_occupancy_grid(it)                              # eager, in place (every 16th step does work)
            losses = step(col_it, prev_it, nxt_it, batch_it)             # {"rgb_loss", "event_loss"}: device scalars
            # step.ray_grads: gradients w.r.t. the rays of this step (per bundle), for a pose optimiser outside the graph

    ``ray_grads=True`` makes the static ray tensors leaves that require gradients (BASELINE config 4: BAD-NeRF pose
    optimisation); their ``.grad`` after a replay is the gradient of the summed loss w.r.t. the rays that were copied in.
    ``jitter``: "graph" draws the stratified offsets inside the graph; a tensor-valued call argument ``jitter=`` is copied
    into a static input instead when the object was built with ``jitter="input"``."""

    def __init__(self, model, opt: FlatAdam, col: Optional[RayBundle], prev: Optional[RayBundle], nxt: Optional[RayBundle],
                 batch: Dict[str, object], ray_grads: bool = False, jitter: str = "graph", warmup: int = 3,
                 grad_scale: float = 1.0):
        assert jitter in ("graph", "input")
        assert model.training, "the captured step is the training step"
        self.model, self.opt, self.grad_scale = model, opt, grad_scale
        self._deferred_before = (model.deferred_counts, model.deferred_max_slots)
        model.deferred_counts, model.deferred_max_slots = True, 1 << 62     # nothing inside the graph may wait for the host
        self.col, self.prev, self.nxt = (_static_like(b, ray_grads) for b in (col, prev, nxt))
        self.batch = self._static_batch(batch)
        n_total = sum(len(b) for b in (self.col, self.prev, self.nxt) if b is not None)
        dev = opt.flat.data.device
        self.jitter = torch.rand(n_total, device=dev) if jitter == "input" else None
        self.losses: Dict[str, Tensor] = {}
        self.outputs = None
        self.graph, self._overflow_flags = capture_body(self._body, opt, model.occupancy_grid, warmup)
        self.replays = 0

    @staticmethod
    def _static_batch(batch):
        out = {}
        for k, v in batch.items():
            if isinstance(v, dict):
                out[k] = {kk: (vv.detach().clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
            else:
                out[k] = v.detach().clone() if torch.is_tensor(v) else v
        return out

    def _body(self):
        self.opt.zero_grad()
        for b in (self.col, self.prev, self.nxt):
            if b is not None and b.origins.requires_grad:
                b.origins.grad = None
                b.directions.grad = None
        out, losses, _ = self.model.train_step_bundles(self.col, self.prev, self.nxt, self.batch, jitter=self.jitter)
        sum(losses.values()).backward()
        self.opt.step_staged(self.grad_scale)
        self.losses, self.outputs = losses, out

    @property
    def ray_grads(self) -> Dict[str, Optional[Tuple[Tensor, Tensor]]]:
        return {k: (None if b is None or not b.origins.requires_grad else (b.origins.grad, b.directions.grad))
                for k, b in (("col", self.col), ("prev", self.prev), ("next", self.nxt))}

    def __call__(self, col: Optional[RayBundle], prev: Optional[RayBundle], nxt: Optional[RayBundle], batch: Dict[str, object],
                 jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
        _copy_bundle(self.col, col)
        _copy_bundle(self.prev, prev)
        _copy_bundle(self.nxt, nxt)
        with torch.no_grad():
            for k, v in self.batch.items():
                if isinstance(v, dict):
                    for kk, vv in v.items():
                        if torch.is_tensor(vv):
                            vv.copy_(batch[k][kk], non_blocking=True)
                elif torch.is_tensor(v):
                    v.copy_(batch[k], non_blocking=True)
            if self.jitter is not None:
                if jitter is None:
                    self.jitter.uniform_()
                else:
                    self.jitter.copy_(jitter, non_blocking=True)
            elif jitter is not None:
                raise ValueError('this step draws its jitter inside the graph; build it with jitter="input" to pass one')
        self.opt.prepare_step()
        self.graph.replay()
        self.replays += 1
        return self.losses

    def check_overflow(self) -> None:
        """One host synchronisation: raises if a replayed marcher call exceeded the proven per-ray capacity (never expected)."""
        if self.replays and self._overflow_flags and bool(torch.stack([f.reshape(()) for f in self._overflow_flags]).any().item()):
            raise RuntimeError("captured step: a ray produced more samples than LSEOccGridEstimator._cap_per_ray allows")

    def close(self):
        self.model.deferred_counts, self.model.deferred_max_slots = self._deferred_before
