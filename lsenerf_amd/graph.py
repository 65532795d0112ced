"""The whole training step as ONE replayed HIP graph.

In the configuration LSENeRF trains in (3510 rays per step, ~260 samples per ray after culling) the kernels of a step take
2.8 ms, but issuing them from Python -- ~26 C-ABI launches, a dozen autograd nodes each way, ~40 allocations -- takes 3.3 ms
of host time, and the two read-backs of the sampler's sample counts stop the host from running ahead of the GPU
(bench.py ``cfg2_composition``: 3.6 - 4.5 ms per step depending on the box's CPU).  The reference pays the same kind of cost
in nerfstudio's trainer.  With the counts kept on the device (``LSENeRFModel.deferred_counts``, the ``n_dev`` argument of the per-sample entry points) nothing in
a step depends on the host any more, so the step -- sampler, visibility pre-pass, field, volume rendering, loss epilogue,
backward, Adam -- is captured once into a HIP graph (``torch.cuda.graph``: the C-ABI launches go to torch's current stream,
which is the capturing stream) and replayed with one call per step.

What varies from step to step enters through device memory at fixed addresses:
  * the rays and targets of the step   -> copied into the static input tensors before the replay;
  * the stratified jitter              -> ``torch.rand`` inside the graph (graph-safe Philox offsets), or a static input;
  * the learning rate / bias corrections -> derived ON THE DEVICE from a step counter the graph itself advances
    (lse_adam_schedule_dev in front of lse_adam_step_dev): no per-step host -> device copy that a host running ahead could race;
  * the occupancy grid and ``occs.mean()`` (the cap of the alpha threshold) -> refreshed in place, outside the graph, by
    ``LSENeRFModel.update_occupancy_grid`` between replays (lse_visibility_mask_cap reads the mean on the device).
Everything else (step size, cone angle, shapes, capacities) is constant for a given model and batch composition.
The values are those of the eager step: the same kernels run on the same samples (tests/test_gpu_graph.py).

``prefetch_march``: the ray marcher reads the rays and the occupancy grid and nothing else -- no parameters -- and it is the one
kernel of the step that cannot fill the chip (one wave per ray, a serial float recurrence: 0.26 - 0.31 ms for 3510 rays, 10 % of the
3-bundle step).  With prefetching the step's graph forks: a side stream marches the NEXT step's rays while the main stream runs the
current step on the samples the previous replay left behind; the side stream forks off right before the hash backward, whose
launch waits for the memory-side atomic units rather than for the CUs (tools/overlap_probe.py: issued together on two streams the two
kernels take 1.52 ms against 0.31 + 1.38 one after the other).  Inside the graph the gain is a quarter of the marcher: 2.67 -> 2.61 ms
per step in the reference's default configuration, 2.70 -> 2.63 for the 3-bundle step, 5.82 -> 5.74 at the metric size.  Two graphs alternate between two sample buffers (one reads A and fills B, the other reads B and fills A), so
nothing is copied.  The marcher of a step therefore sees the occupancy grid as it was one step earlier: when the grid has
changed in between (``LSEOccGridEstimator.grid_version``; every 16th step), or no rays were announced, the current rays are
marched again, eagerly, before the replay -- the samples are always those of the eager step.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .optim import FlatAdam
from .rays import RayBundle


def _static_bundles(bundles: Sequence[Optional[RayBundle]]) -> Tuple[Optional[RayBundle], ...]:
    """Static copies of the (up to three) bundles of a step whose fields are ROW BLOCKS OF ONE BUFFER PER FIELD: the step joins its
    bundles into one (``RayBundle.cat(alias_blocks=True)``), and for such parts the join is the buffer itself -- no concatenation
    kernels inside the graph (four forward, and the split of the ray gradients backward).  The static tensors never require
    gradients: a step that wants ray gradients runs on FRESH leaves over the same storage (``_fresh_ray_leaves``)."""
    given = [b for b in bundles if b is not None]
    if not given:
        return tuple(None for _ in bundles)
    joined = RayBundle.cat([RayBundle(origins=b.origins.detach(), directions=b.directions.detach(), pixel_area=b.pixel_area,
                                      camera_indices=b.camera_indices, nears=b.nears, fars=b.fars, times=b.times,
                                      metadata=dict(b.metadata)) for b in given]) if len(given) > 1 else given[0]
    c = lambda t: None if t is None else t.detach().clone().contiguous()
    base = RayBundle(origins=c(joined.origins), directions=c(joined.directions),
                     pixel_area=c(joined.pixel_area), camera_indices=c(joined.camera_indices), nears=c(joined.nears), fars=c(joined.fars),
                     times=c(joined.times), metadata={k: c(v) for k, v in joined.metadata.items()})
    if len(given) == 1:
        return tuple(base if b is not None else None for b in bundles)
    return _row_blocks(base, bundles)


def _row_blocks(base: RayBundle, bundles: Sequence[Optional[RayBundle]], origins: Optional[Tensor] = None,
                directions: Optional[Tensor] = None) -> Tuple[Optional[RayBundle], ...]:
    """``base`` cut into the row blocks of ``bundles`` (plain windows, cut WITHOUT autograd: a view made with grad mode on would own
    an edge to its buffer's AccumulateGrad node).  ``origins`` / ``directions`` replace the base's ray tensors (fresh leaves)."""
    o_src = base.origins if origins is None else origins
    d_src = base.directions if directions is None else directions
    out, lo = [], 0
    for b in bundles:
        if b is None:
            out.append(None)
            continue
        hi = lo + len(b)
        with torch.no_grad():
            cut = lambda t: None if t is None else t[lo:hi]
            blk = RayBundle(origins=cut(o_src), directions=cut(d_src), pixel_area=cut(base.pixel_area),
                            camera_indices=cut(base.camera_indices), nears=cut(base.nears), fars=cut(base.fars), times=cut(base.times),
                            metadata={k: cut(v) for k, v in base.metadata.items()})
        out.append(blk)
        lo = hi
    return tuple(out)


def _leaf(t: Tensor) -> Tensor:
    """The tensor that owns ``t``'s storage window: ``t`` itself, or the buffer it is a row block of."""
    return t if t._base is None else t._base


def _fresh_ray_leaves(statics: Sequence[Optional[RayBundle]]):
    """The static bundles again, with origins / directions replaced by NEW leaf tensors over the same storage that require
    gradients: ``(bundles, origins_leaf, directions_leaf)``.

    Why new tensors on every run: a leaf's AccumulateGrad node is created on first use, bound to the stream that is current then,
    and lives as long as any autograd graph references it.  A leaf that survives from the warm-up runs (or from an earlier capture)
    carries its node into the capture; if that node's stream is not the capturing stream, torch synchronises the two streams inside
    the capture ("The AccumulateGrad node's stream does not match ...") -- with the default stream on the other side the HIP runtime
    crashed in hipStreamEndCapture (round 4).  ``detach()`` costs no kernel, the new leaf has no node yet, so the node of a run is
    always created by that run, on that run's stream, and dies with that run's graph.  The ray gradients of the captured run land
    in ``leaf.grad`` -- memory of the graph's private pool, rewritten by every replay, alive as long as the leaf is held."""
    given = [b for b in statics if b is not None]
    base_o, base_d = _leaf(given[0].origins), _leaf(given[0].directions)
    o = base_o.detach().requires_grad_(True)
    d = base_d.detach().requires_grad_(True)
    if given[0].origins._base is None:               # one bundle: it IS the buffer
        b0 = given[0]
        fresh = RayBundle(origins=o, directions=d, pixel_area=b0.pixel_area, camera_indices=b0.camera_indices, nears=b0.nears,
                          fars=b0.fars, times=b0.times, metadata=b0.metadata)
        return tuple(fresh if b is not None else None for b in statics), o, d
    whole = lambda t: None if t is None else _leaf(t)
    b0 = given[0]
    base = RayBundle(origins=base_o, directions=base_d, pixel_area=whole(b0.pixel_area), camera_indices=whole(b0.camera_indices),
                     nears=whole(b0.nears), fars=whole(b0.fars), times=whole(b0.times),
                     metadata={k: whole(v) for k, v in b0.metadata.items()})
    return _row_blocks(base, statics, origins=o, directions=d), o, d


def _copy_pairs(dst: Optional[RayBundle], src: Optional[RayBundle], pairs: list) -> None:
    """Append the (static tensor, given tensor) pairs of one bundle to ``pairs`` (checked: same composition as captured)."""
    if (dst is None) != (src is None):
        raise ValueError("the batch composition of a captured step is fixed: a bundle appeared or disappeared")
    if dst is None:
        return
    if len(dst) != len(src):
        raise ValueError(f"the batch composition of a captured step is fixed: {len(dst)} rays captured, {len(src)} given")
    for name in ("origins", "directions", "pixel_area", "camera_indices", "nears", "fars", "times"):
        d, s = getattr(dst, name), getattr(src, name)
        if (d is None) != (s is None):
            raise ValueError(f"RayBundle.{name} was {'set' if d is not None else 'absent'} at capture")
        if d is not None:
            pairs.append((d, s))
    if set(dst.metadata) != set(src.metadata):
        raise ValueError("RayBundle.metadata keys differ from the captured step's")
    for k, v in dst.metadata.items():
        pairs.append((v, src.metadata[k]))


@torch.no_grad()
def _run_copies(pairs: list) -> None:
    """All input copies of a step in as few launches as possible.  One ``copy_`` per tensor is a ~5 us kernel each, and a 3-bundle
    step with prefetching has 26 of them queued in front of every replay: 0.125 ms of a 2.6 ms step (rocprofv3 timeline,
    tools/graph_timeline.py).  ``torch._foreach_copy_`` moves each group of same-typed device tensors with one multi-tensor kernel."""
    pairs = [(d, s) for d, s in pairs if d.data_ptr() != s.data_ptr() or d.shape != s.shape]      # already in place: nothing to do
    groups: Dict[tuple, list] = {}
    for d, s in pairs:
        if d.shape != s.shape:
            raise ValueError(f"a static input of shape {tuple(d.shape)} was given a tensor of shape {tuple(s.shape)}")
        key = (d.dtype, s.dtype, s.device)
        groups.setdefault(key, []).append((d, s))
    for (dd, sd, sdev), items in groups.items():
        if len(items) > 1 and dd == sd and sdev == items[0][0].device and hasattr(torch, "_foreach_copy_"):
            torch._foreach_copy_([d for d, _ in items], [s for _, s in items], non_blocking=True)
        else:
            for d, s in items:
                d.copy_(s, non_blocking=True)


def capture_body(body, opt: Optional[FlatAdam], estimator, warmup: int = 3, pool=None, stream: Optional[torch.cuda.Stream] = None,
                 before_capture=None):
    """Capture ``body()`` -- a zero-argument callable that runs one step on tensors at fixed addresses and synchronises with
    nothing -- into a HIP graph.  Warm-up runs on a side stream as torch.cuda.graph requires (allocator pools, lazy
    initialisation); the optimizer state those eager runs change is saved and restored, so capturing trains nothing.

    Warm-up and capture run on ONE stream (``stream``, or a new one), and ``before_capture()`` lets the caller drop whatever the
    warm-up runs left behind (losses, outputs: their autograd graphs) before the capture starts: an autograd node that outlives the
    warm-up -- an AccumulateGrad node above all -- is bound to the stream it was created on, and a captured backward that runs
    through a node of ANOTHER stream synchronises with that stream inside the capture (_fresh_ray_leaves).  Returns the graph."""
    dev = opt.flat.data.device if opt is not None else estimator.occs.device
    estimator._occ_mean_device()                          # allocate / refresh the device-side alpha cap before the capture
    saved = (opt.flat.data.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count) if opt is not None else None
    side = stream if stream is not None else torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(warmup):
            if opt is not None:
                opt.prepare_step()
            body()
    torch.cuda.current_stream(dev).wait_stream(side)
    estimator.check_deferred_overflow()                   # (the eager warm-up runs' flags; one synchronisation, at construction)
    if before_capture is not None:
        before_capture()
    graph = torch.cuda.CUDAGraph()
    if opt is not None:
        opt.prepare_step()
    with torch.cuda.graph(graph, pool=pool, stream=side):
        body()
    # (the captured marcher calls OR into the estimator's sticky overflow accumulator at every replay, like the eager ones:
    #  LSEOccGridEstimator._overflow_flag -- read at the occupancy refresh and by check_overflow())
    if opt is not None:
        with torch.no_grad():
            opt.flat.data.copy_(saved[0]); opt.exp_avg.copy_(saved[1]); opt.exp_avg_sq.copy_(saved[2])
        opt.step_count = saved[3]
    return graph


class GraphedTrainStep:
    """``model.train_step_bundles`` + ``backward`` + ``FlatAdam`` step, captured once and replayed.

        step = GraphedTrainStep(model, opt, col, prev, nxt, batch)      # example inputs fix the composition
        for it in range(...):
            model.update_occupancy_grid(it)                              # eager, in place (every 16th step does work)
            losses = step(col_it, prev_it, nxt_it, batch_it)             # {"rgb_loss", "event_loss"}: device scalars
            # step.ray_grads: gradients w.r.t. the rays of this step (per bundle), for a pose optimiser outside the graph

    Earlier EAGER steps of the same model whose losses / outputs are still alive keep their autograd graphs and with them the
    parameters' AccumulateGrad nodes, bound to the stream they were created on; a capture whose backward hands a gradient to such a
    node leaves the capturing stream (torch warns, the HIP runtime crashed in hipStreamEndCapture).  The step hands none over: every
    parameter receives its gradient directly -- the hash table, the MLPs, the embedding, the loss epilogue's scalars
    (ops._scalar_param_grads) and, since round 5, the parameters of the torch route as well (ops.direct_grad_params: MLP intensity
    mappers, ``ThreeToOne``, ``Powpow`` outside the fused epilogue); the rays are fresh leaves in every run (_fresh_ray_leaves);
    warm-up and capture share one stream and the warm-up's outputs are dropped before the capture (capture_body).
    tests/test_gpu_graph.py runs captures behind live eager graphs in a child process, with that torch warning as an error.

    ``ray_grads=True``: the step differentiates w.r.t. the rays (BASELINE config 4: BAD-NeRF pose optimisation);
    ``step.ray_grads`` after a replay holds the gradient of the summed loss w.r.t. the rays that were copied in.
    ``jitter``: "graph" draws the stratified offsets inside the graph; a tensor-valued call argument ``jitter=`` is copied
    into a static input instead when the object was built with ``jitter="input"``.
    ``prefetch_march=True`` (module docstring): announce the next step's rays with ``next_bundles=(col, prev, nxt)`` (and
    ``next_jitter=`` for jitter="input") at every call and hand the next call those very tensors, unmodified.  A call whose rays are
    not the announced ones (other tensors, or the same ones written to in between) does not train on the wrong samples: the samples
    marched ahead are dropped and the given rays marched before the replay (``remarched_unannounced`` counts these calls)."""

    def __init__(self, model, opt: FlatAdam, col: Optional[RayBundle], prev: Optional[RayBundle], nxt: Optional[RayBundle],
                 batch: Dict[str, object], ray_grads: bool = False, jitter: str = "graph", warmup: int = 3,
                 grad_scale: float = 1.0, prefetch_march: bool = False, prefetch_fork: str = "hash_bwd",
                 optimizer_in_graph: bool = True):
        assert jitter in ("graph", "input")
        assert model.training, "the captured step is the training step"
        self.model, self.opt, self.grad_scale = model, opt, grad_scale
        # False: the graph ends with the backward pass and the caller finishes the step -- data parallel: all-reduce of
        # opt.flat.grad (lsenerf_amd.dist), then opt.step(grad_scale=1 / world) -- two more launches instead of ~25
        self.optimizer_in_graph = bool(optimizer_in_graph)
        self._deferred_before = (model.deferred_counts, model.deferred_max_slots)
        model.deferred_counts, model.deferred_max_slots = True, 1 << 62     # nothing inside the graph may wait for the host
        self.col, self.prev, self.nxt = _static_bundles((col, prev, nxt))
        self.want_ray_grads = bool(ray_grads)
        self._ray_leaves: Optional[Tuple[Tensor, Tensor]] = None      # (origins, directions) leaves of the last run of _body
        self.batch = self._static_batch(batch)
        n_total = sum(len(b) for b in (self.col, self.prev, self.nxt) if b is not None)
        dev = opt.flat.data.device
        self.jitter = torch.rand(n_total, device=dev) if jitter == "input" else None
        self.losses: Dict[str, Tensor] = {}
        self.outputs = None
        self._one = torch.ones((), dtype=torch.float32, device=dev)
        self.prefetch = bool(prefetch_march)
        assert prefetch_fork in ("start", "hash_bwd")
        self.prefetch_fork = prefetch_fork
        self.replays = 0
        self._capture_stream = torch.cuda.Stream(device=dev)           # warm-up runs AND captures (capture_body)
        if not self.prefetch:
            self.graph = capture_body(self._body, opt, model.occupancy_grid, warmup, stream=self._capture_stream,
                                      before_capture=self._drop_run)
            return
        # -- marcher of the next step on a side stream: two graphs alternate between two sample buffers
        self.next_col, self.next_prev, self.next_nxt = _static_bundles((col, prev, nxt))
        self.next_jitter = torch.rand(n_total, device=dev) if jitter == "input" else None
        with torch.no_grad():
            self._pm = [model.premarch_bundles(self.col, self.prev, self.nxt, jitter=self.jitter, alias_blocks=True) for _ in range(2)]
        est = model.occupancy_grid
        self._side = torch.cuda.Stream(device=dev)
        self._graphs, self._losses_of, self._outputs_of, self._ray_grads_of = [], [], [], []
        for x in (0, 1):
            g = capture_body(lambda x=x: self._body_prefetch(x), opt, est, warmup if x == 0 else 1,
                             pool=self._graphs[0].pool() if self._graphs else None, stream=self._capture_stream,
                             before_capture=self._drop_run)
            self._graphs.append(g)
            self._losses_of.append(self.losses)
            self._outputs_of.append(self.outputs)
            self._ray_grads_of.append(self._collect_ray_grads())
        self._announced = None       # identity of the rays the other buffer was marched for (_signature: holds the tensors)
        self.remarched_unannounced = 0   # calls whose rays were not the announced ones (their samples were marched again)
        self._cur = 0                # the buffer that holds (or will be given) the samples of the rays in the static CURRENT bundles
        self._pm_version = None      # grid_version under which the other buffer was filled by the last replay; None: not filled
        self._last = 0

    @staticmethod
    def _static_batch(batch):
        out = {}
        for k, v in batch.items():
            if isinstance(v, dict):
                out[k] = {kk: (vv.detach().clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
            else:
                out[k] = v.detach().clone() if torch.is_tensor(v) else v
        return out

    def _drop_run(self):
        """Forget the last run of ``_body`` (capture_body calls this between the warm-up runs and the capture): its losses and
        outputs hold that run's autograd graph, and with it autograd nodes bound to the warm-up."""
        self.losses, self.outputs, self._ray_leaves = {}, None, None

    def _body(self, premarched=None):
        self.opt.zero_grad()
        col, prev, nxt = self.col, self.prev, self.nxt
        if self.want_ray_grads:
            (col, prev, nxt), o, d = _fresh_ray_leaves((col, prev, nxt))
            self._ray_leaves = (o, d)
        out, losses, _ = self.model.train_step_bundles(col, prev, nxt, self.batch, jitter=self.jitter,
                                                       premarched=premarched, alias_blocks=True)
        # backward of rgb_loss + event_loss without forming the sum: every root gets the upstream gradient 1 from one static tensor
        # (the sum, its backward and the two ones_like fills were five ~5 us launches of every replay)
        vals = list(losses.values())
        torch.autograd.backward(vals, [self._one] * len(vals))
        if self.optimizer_in_graph:
            self.opt.step_staged(self.grad_scale)
        self.losses, self.outputs = losses, out

    def _body_prefetch(self, x: int):
        """The step on the samples in buffer x; beside it, on the side stream, the marcher of the NEXT rays into buffer 1 - x.
        Where the side stream forks off decides what the marcher shares the chip with: "start" = the whole step; "hash_bwd" = it
        becomes runnable together with the hash backward (the last and longest kernel of the step, bound by the memory-side atomic
        units rather than by the CUs).  Measured (tools/ab_prefetch_fork.sh, graphed step without -> with the marcher on the side
        stream): "start" 2.668 -> 2.646 ms (default configuration) / 2.688 -> 2.662 (3-bundle step); "hash_bwd" 2.671 -> 2.605 /
        2.699 -> 2.630; behind the fine levels of a two-launch hash backward: no gain (the split itself costs what the overlap
        wins)."""
        from . import ops
        main = torch.cuda.current_stream()

        forked = []

        def fork():
            if forked:                                 # (a step with several hash backwards: the first one forks)
                return
            forked.append(True)
            cur = torch.cuda.current_stream()          # (the backward's stream when called from the autograd thread)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side), torch.no_grad():
                self.model.premarch_bundles(self.next_col, self.next_prev, self.next_nxt, jitter=self.next_jitter, out=self._pm[1 - x],
                                            alias_blocks=True)

        table = self.model.field.mlp_base_grid.params
        saved = ops.get_hash_bwd_hook(table, "before")
        try:
            if self.prefetch_fork == "start":
                fork()
            else:
                ops.set_hash_bwd_hook(table, "before", fork)       # (this model's table only)
            self._body(premarched=self._pm[x])
        finally:
            ops.set_hash_bwd_hook(table, "before", saved)
        if not forked:          # (no hash backward ran: a frozen table)
            fork()
        main.wait_stream(self._side)

    def _collect_ray_grads(self) -> Dict[str, Optional[Tuple[Tensor, Tensor]]]:
        """Per bundle: the row block of the ray leaves' gradient that belongs to it (one leaf per field for all bundles)."""
        out, lo = {}, 0
        go, gd = (self._ray_leaves[0].grad, self._ray_leaves[1].grad) if self._ray_leaves is not None else (None, None)
        single = sum(b is not None for b in (self.col, self.prev, self.nxt)) == 1
        for k, b in (("col", self.col), ("prev", self.prev), ("next", self.nxt)):
            if b is None:
                out[k] = None
                continue
            hi = lo + len(b)
            out[k] = None if go is None else ((go, gd) if single else (go[lo:hi], gd[lo:hi]))
            lo = hi
        return out

    @property
    def ray_grads(self) -> Dict[str, Optional[Tuple[Tensor, Tensor]]]:
        """Gradients w.r.t. the rays of the last replayed step (each of the two alternating graphs owns its gradient tensors)."""
        return self._ray_grads_of[self._last] if self.prefetch else self._collect_ray_grads()

    def __call__(self, col: Optional[RayBundle], prev: Optional[RayBundle], nxt: Optional[RayBundle], batch: Dict[str, object],
                 jitter: Optional[Tensor] = None, next_bundles: Optional[Sequence[Optional[RayBundle]]] = None,
                 next_jitter: Optional[Tensor] = None) -> Dict[str, Tensor]:
        if (next_bundles is not None or next_jitter is not None) and not self.prefetch:
            raise ValueError("next_bundles / next_jitter belong to a step built with prefetch_march=True")
        pairs: list = []
        _copy_pairs(self.col, col, pairs)
        _copy_pairs(self.prev, prev, pairs)
        _copy_pairs(self.nxt, nxt, pairs)
        for k, v in self.batch.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    if torch.is_tensor(vv):
                        pairs.append((vv, batch[k][kk]))
            elif torch.is_tensor(v):
                pairs.append((v, batch[k]))
        if self.jitter is not None:
            if jitter is None:
                with torch.no_grad():
                    self.jitter.uniform_()
            else:
                pairs.append((self.jitter, jitter))
        elif jitter is not None:
            raise ValueError('this step draws its jitter inside the graph; build it with jitter="input" to pass one')
        if self.prefetch and next_bundles is not None:       # the announced rays travel with the same launches
            for dst, src in zip((self.next_col, self.next_prev, self.next_nxt), next_bundles):
                _copy_pairs(dst, src, pairs)
            if self.next_jitter is not None and next_jitter is not None:
                pairs.append((self.next_jitter, next_jitter))
        _run_copies(pairs)
        if self.prefetch:
            # the samples marched ahead belong to the rays that were ANNOUNCED: if the rays given now are not those very tensors,
            # unmodified (same storage, same version counters; jitter likewise), the samples are dropped and these rays marched
            if self._pm_version is not None and not self._same_rays(self._announced, self._signature((col, prev, nxt), jitter)):
                self._pm_version = None
                self.remarched_unannounced += 1
            return self._replay_prefetch(next_bundles, next_jitter)
        if self.optimizer_in_graph:
            self.opt.prepare_step()
        self.graph.replay()
        self.replays += 1
        return self.losses

    @staticmethod
    def _signature(bundles, jitter):
        """Identity of a set of rays: the tensor OBJECTS (held, so that their storage cannot be handed to other rays in between) and
        their in-place version counters -- origins, directions, nears / fars, times, camera indices, every metadata tensor, the
        jitter.  Addresses alone are not an identity: a caller that drops the announced bundle and builds a new one commonly gets the
        same block back from the caching allocator, at version 0."""
        sig = []
        for b in bundles:
            if b is None:
                sig.append(None)
                continue
            ts = [b.origins, b.directions] + [getattr(b, n) for n in ("nears", "fars", "times", "camera_indices", "pixel_area")
                                              if getattr(b, n) is not None] + \
                 [v for _, v in sorted(b.metadata.items()) if torch.is_tensor(v)]
            sig.append([(t, t._version) for t in ts])
        sig.append(None if jitter is None else [(jitter, jitter._version)])
        return sig

    @staticmethod
    def _same_rays(a, b) -> bool:
        if a is None or b is None or len(a) != len(b):
            return False
        for x, y in zip(a, b):
            if (x is None) != (y is None):
                return False
            if x is None:
                continue
            if len(x) != len(y) or any(t1 is not t2 or v1 != v2 for (t1, v1), (t2, v2) in zip(x, y)):
                return False
        return True

    def _replay_prefetch(self, next_bundles, next_jitter) -> Dict[str, Tensor]:
        est, x = self.model.occupancy_grid, self._cur
        if self._pm_version is None or self._pm_version != est.grid_version:
            # nothing was marched ahead for these rays, or the grid has been refreshed since: march them now (eagerly, same stream)
            with torch.no_grad():
                self.model.premarch_bundles(self.col, self.prev, self.nxt, jitter=self.jitter, out=self._pm[x], alias_blocks=True)
        if next_bundles is not None:         # (the announced rays themselves were copied in by __call__, with this step's inputs)
            if self.next_jitter is not None:
                if next_jitter is None:
                    with torch.no_grad():
                        self.next_jitter.uniform_()
            elif next_jitter is not None:
                raise ValueError('this step draws its jitter inside the graph; build it with jitter="input" to pass one')
        version = est.grid_version                 # (the grid cannot change while the replay runs: refreshes are eager, in order)
        if self.optimizer_in_graph:
            self.opt.prepare_step()
        self._graphs[x].replay()
        self.replays += 1
        self._pm_version = version if next_bundles is not None else None
        self._announced = self._signature(next_bundles, next_jitter) if next_bundles is not None else None
        self._last, self._cur = x, 1 - x
        self.losses, self.outputs = self._losses_of[x], self._outputs_of[x]
        return self.losses

    def check_overflow(self) -> None:
        """One host synchronisation: raises if a marcher call -- replayed or eager -- exceeded the per-ray capacity (the estimator's
        sticky accumulator; ``LSENeRFModel.update_occupancy_grid`` makes the same check at every refresh)."""
        self.model.occupancy_grid.check_deferred_overflow()

    def close(self):
        self.model.deferred_counts, self.model.deferred_max_slots = self._deferred_before
