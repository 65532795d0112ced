#!/usr/bin/env python3
"""Benchmark of the LSENeRF hot path on MI355X: train-step rays/sec on a 4096-ray x 1024-sample batch
(BASELINE.json metric; SURVEY.md section 8d workload "M-march", taken literally since round 3).

One step = occupancy-grid ray marching (ONE-level 128^3 grid, all cells occupied, cone angle 0, no culling;
rays from the sphere of radius 1.5 aimed at uniform targets in [-0.5, 0.5]^3; constant step chosen so that the rays
yield 1024 samples on average -- the actual N is reported) -> hash-grid field + fused MLPs forward -> packed volume
rendering -> MSE(rgb, target) -> backward (incl. ray/pose gradients) -> [RCCL all-reduce of the flat gradient] -> fused
Adam.  Synthetic rays, random-init parameters, fp32 throughout.  Extra keys of the JSON line (N = 1 only): the round-1/2
headline workload (`m_march_inside_box`: origins inside the box, exactly 1024 samples per ray -- the friendlier input),
SURVEY 8d's M-packed, the reference's default configuration, and the reference's real 3-bundle step compositions.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU over RCCL; per-GPU work is fixed (weak scaling): each rank renders its own 4096 rays and the gradient
is all-reduced once per step.  Under a launcher (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``:
WORLD_SIZE is set) this process IS a rank.  Started bare (``python bench.py --gpus N``, no WORLD_SIZE) it is the PARENT: before
anything touches the GPU -- torch is not even imported -- it starts N fresh rank processes (lsenerf_amd/launch.py; what
R:train.py:171-234 does with mp.spawn), relays rank 0's JSON line and exits with the ranks' code.  A rank whose WORLD_SIZE
differs from --gpus is an error, never a silent N = 1 measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _load_launcher():
    """lsenerf_amd/launch.py loaded by PATH: importing the package would import torch, and the launching parent must not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lse_launch", os.path.join(ROOT, "lsenerf_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _gpus_requested(argv) -> int:
    """--gpus N / --gpus=N from the command line, before argparse (and torch) come into play."""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


# `python bench.py --gpus N` without a launcher: this process is the PARENT of N ranks and stays clear of torch and the GPU
IS_LAUNCHING_PARENT = __name__ == "__main__" and _load_launcher().needs_launch(_gpus_requested(sys.argv[1:]))
if not IS_LAUNCHING_PARENT:
    import torch

RAYS_PER_GPU = 4096
SAMPLES_PER_RAY = 1024
BYTES_PER_SAMPLE_STEP = 2136           # 1024 gather + 1024 scatter + 88 packed streams
HASH_BYTES_PER_SAMPLE = 1024           # L*8*F*4: algorithmic bytes of one hash gather / scatter pass
# fused-MLP flops per sample, fwd + bwd(data) + wgrad, as executed (base 32->64->16, head 16(+per-ray bias)->64->64->16):
# fwd 2*(32*64+64*16) + 2*(16*64+64*64+64*16) = 18432; bwd data = same contraction sizes; wgrad = same  -> 3x
MLP_FLOP_PER_SAMPLE = 3 * 18432
# ... and as EXECUTED since round 2: the kernels of lsenerf_amd/csrc/mlp_x6.h cut every f32 operand into three bf16 pieces and run
# six piece products per multiply-add on v_mfma_f32_16x16x32_bf16 (f32-equivalent result), recompute the hidden layers in the
# backward and cut the pieces with remainder MFMAs.  MFMA instructions per 32-sample tile, counted in the ISA
# (tools/asm_timeline.py): head fwd 180, base fwd 96, head bwd 528, base bwd 244; 16*16*32*2 flop each.
MLP_BF16_FLOP_PER_SAMPLE = (180 + 96 + 528 + 244) * 16384 // 32
HBM_PEAK = 8.0e12                      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


N_RAY_SETS = 8                         # the timed loop cycles this many pre-generated (rays, targets, stratified offsets) sets


def build_workload(device, seed, kind="sphere", n_sets=1):
    """kind "sphere": SURVEY 8d M-march exactly (the headline).  kind "inside": the round-1/2 headline (origins in
    [-0.5, 0.5]^3, 4-level grid, per-ray t_max capped: exactly 1024 samples per ray, 65 % of them in the contracted shell).

    Returns ``(model, sets, counts)``: ``sets`` = ``n_sets`` tuples ``(ray bundle, target, jitter)``, all resident in HBM -- a training
    step draws NEW pixels and a new stratified offset per ray every step (R:lse_nerf/lse_datamanager.py:135-144,
    R:lse_nerf/lse_grid_estimator.py:85-87), so the timed loop cycles the sets instead of replaying one input; ``counts`` = the
    samples each set yields at the calibrated step size."""
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    torch.manual_seed(96)                                     # identical parameters on every rank
    inside = kind == "inside"
    cfg = LSENeRFModelConfig(cone_angle=0.0, alpha_thre=0.0, grid_levels=4 if inside else 1)   # constant step, no culling
    model = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(device)
    model.train()
    model.occupancy_grid.mark_all_occupied()
    g = torch.Generator().manual_seed(seed)                   # rank-dependent rays (R:train.py:104 seeds by rank)
    R = RAYS_PER_GPU
    sets = []
    for k in range(n_sets):
        if inside:
            o = (torch.rand(R, 3, generator=g) - 0.5)
            d = torch.randn(R, 3, generator=g)
            d = d / d.norm(dim=-1, keepdim=True)
            step = cfg.render_step_size
            fars = torch.full((R, 1), cfg.near_plane + SAMPLES_PER_RAY * step - 0.25 * step).to(device)
        else:
            o, d = sphere_rays(R, g)
            fars = None
        target = torch.rand(R, 3, generator=g)
        # stratified sampling: one offset in [0, 1) steps per ray and step.  A single set keeps the round-1..4 fixed draw (zeros),
        # so that the one-set context workloads stay comparable with the earlier rounds
        jitter = torch.rand(R, generator=g) if n_sets > 1 else torch.zeros(R)
        rb = RayBundle(origins=o.to(device).requires_grad_(True), directions=d.to(device).requires_grad_(True),
                       camera_indices=torch.zeros(R, 1, dtype=torch.long, device=device), fars=fars,
                       metadata={"appearance_id": torch.randint(0, 64, (R,), generator=g).to(device)})
        sets.append((rb, target.to(device), jitter.to(device)))
    est = model.occupancy_grid

    def counts():
        return [est.sampling(rb.origins.detach(), rb.directions.detach(), near_plane=cfg.near_plane, far_plane=cfg.far_plane,
                             t_max=rb.fars.reshape(-1) if rb.fars is not None else None, render_step_size=cfg.render_step_size,
                             stratified=True, jitter=jit, return_packed=True)[1].shape[0] for rb, _, jit in sets]
    if not inside:
        # "step chosen so the mean count is ~1024 per ray": the count is ~ (chord inside the box) / step, so one marcher
        # call per set at the reference's step and two corrections settle the MEAN over the sets (the same rule on every rank;
        # rays differ by rank).  A single set's count then sits within ~1 % of 1024 per ray (4096 independent chords)
        for _ in range(3):
            n = sum(counts()) / n_sets
            cfg.render_step_size = float(cfg.render_step_size * n / (R * SAMPLES_PER_RAY))
    return model, sets, counts()


def train_step(model, rb, target, jitter, opt, world, exchange=None, pipeline=None, sharded=None, graphed=None, exposed=None):
    """One step of the headline workload.  ``exposed``: a list that collects (event, event) pairs around the blocking part of the
    gradient exchange (the exchange modes that overlap collect theirs inside lsenerf_amd.dist)."""
    from lsenerf_amd import dist as ldist
    if graphed is not None:
        # N > 1, exchange mode "graphed": everything up to the backward pass is ONE replayed HIP graph (optimizer_in_graph=False),
        # then the all-reduce of the flat gradient, then Adam -- three host calls per step (R:lse_nerf/lse_pipeline.py:95-98's DDP
        # all-reduce sits at the same place: between backward and the optimizer)
        loss = graphed(rb, None, None, {"col_batch": {"image": target}, "evs_batch": None}, jitter=jitter)["rgb_loss"]
        ev = None
        if exposed is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        ldist.allreduce_grads(opt.flat.grad)
        if ev is not None:
            ev[1].record()
            exposed.append(ev)
        opt.step(grad_scale=1.0 / world)
        return graphed.outputs["col_out"]["num_samples_per_ray"].sum(), loss
    rb.origins.grad = None
    rb.directions.grad = None
    cfg = model.config
    # the sample count stays on the device (deferred=True: capacity-extent packed arrays + n_dev handed to every per-sample
    # kernel); values are those of the synchronising path bit for bit (tests/test_gpu_deferred.py)
    res = model.occupancy_grid.sampling(
        rb.origins.detach(), rb.directions.detach(), sigma_fn=None, near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        t_max=rb.fars.reshape(-1) if rb.fars is not None else None, render_step_size=cfg.render_step_size, stratified=True, cone_angle=cfg.cone_angle,
        alpha_thre=cfg.alpha_thre, jitter=jitter, **({"deferred": True} if model.use_deferred_counts(len(rb)) else {"return_packed": True}))
    ri, ts, te, packed = res[:4]
    n_dev = res[4] if len(res) == 5 else None       # (None: the model's slot budget sent this batch down the synchronising path)
    # (pipelined exchange: the all-reduce + Adam of the previous step finished inside sampling(), right after the marcher,
    #  through the estimator's after_march_hook -- dist.GradPipeline.attach)
    opt.zero_grad()
    out = model.render_packed(rb, ri, ts, te, packed, n_dev=n_dev)
    # routing (training clamp) + rgb MSE in the fused epilogue kernel pair, as the training step does
    loss = model.fused_loss_dict({"col_out": out, "prev_out": None, "next_out": None},
                                 {"col_batch": {"image": target}, "evs_batch": None})["rgb_loss"]
    loss.backward()
    if pipeline is not None:
        pipeline.start()
        return (n_dev if n_dev is not None else ri.shape[0]), loss
    if sharded is not None:        # reduce-scatter -> Adam on this rank's 1/W shard -> all-gather
        sharded.lr = opt.current_lr()
        opt.step_count += 1
        sharded.step()
        return (n_dev if n_dev is not None else ri.shape[0]), loss
    if exchange is not None:
        exchange.finish()          # the fine levels' table gradients have been in flight since the middle of the hash backward
    elif world > 1:
        ev = None
        if exposed is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        ldist.allreduce_grads(opt.flat.grad)
        if ev is not None:
            ev[1].record()
            exposed.append(ev)
    opt.step(grad_scale=1.0 / world)
    return (n_dev if n_dev is not None else ri.shape[0]), loss


def _event_pass(step_fn, first, steps, names):
    """`steps` calls of step_fn with the C-ABI entry points in `names` (None = all) bracketed by HIP events on the launch
    stream.  Returns ({entry: ms per step}, {entry: launches per step})."""
    from lsenerf_amd import _lib
    _lib.TIMING = {"names": names, "events": []}
    torch.cuda.synchronize()
    for i in range(steps):
        step_fn(first + i)
    torch.cuda.synchronize()
    ev, _lib.TIMING = _lib.TIMING["events"], None
    per = {}
    for name, e0, e1 in ev:
        per.setdefault(name, []).append(e0.elapsed_time(e1))
    return {k: round(sum(v) / steps, 4) for k, v in sorted(per.items())}, {k: len(v) / steps for k, v in sorted(per.items())}


def _timed_steps(step_fn, steps, warmup, breakdown_steps=8):
    """warmup untimed + `steps` timed calls of step_fn(i) WITHOUT instrumentation (an event pair around each of the ~25
    entry points of a step costs 1-2 ms per step in queue bubbles: measured 7.3 vs 8.4-9.5 ms), then a separate
    instrumented pass for the per-entry-point breakdown.  Returns (ms_per_step, {entry: ms per step}, {entry: launches})."""
    for i in range(warmup):
        step_fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(warmup + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    kern, launches = _event_pass(step_fn, warmup + steps, breakdown_steps, None)
    return ms, kern, launches


def _snapshot(model, opt):
    """Parameters, optimizer state and occupancy grid: the two graphed variants are timed from the SAME training state (the
    synthetic scene keeps training during a timing, which moves the grid and with it the sample count)."""
    est = model.occupancy_grid
    return (opt.flat.data.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_count, est.occs.clone(), est.binaries.clone())


def _restore(model, opt, snap):
    est = model.occupancy_grid
    with torch.no_grad():
        opt.flat.data.copy_(snap[0]); opt.exp_avg.copy_(snap[1]); opt.exp_avg_sq.copy_(snap[2])
        opt.step_count = snap[3]
        est.occs.copy_(snap[4]); est.binaries.copy_(snap[5])
    est._bump_grid_version()
    est._invalidate_occ_mean()


def _drop_autograd_graphs(holder):
    """Forget the outputs of the last EAGER step before a step of the same model is captured.  They own their autograd graph and with
    it the AccumulateGrad nodes of the small torch-side parameters, which stay bound to the stream they were created on (the default
    stream); a capture whose backward runs through such a stale node leaves the capturing stream -- torch warns, and the HIP runtime
    can crash in hipStreamEndCapture (lsenerf_amd.graph.GraphedTrainStep's docstring; found with tests/test_gpu_fullsize.py)."""
    import gc
    if isinstance(holder, dict):
        holder.clear()
    elif hasattr(holder, "last"):
        holder.last = None
    gc.collect()


def _graphed_timing(model, opt, col, prev, nxt, batch, steps, warmup, refresh_from=None, prefetch=False, cycle=None):
    """The same step as ONE replayed HIP graph (lsenerf_amd.graph.GraphedTrainStep: device-side sample counts, staged Adam
    scalars, jitter drawn inside the graph; the occupancy refresh stays an eager in-place call between replays).
    ``prefetch``: the graph also marches the NEXT step's rays on a side stream (here: the same rays, announced at every call; on the
    refresh steps the samples marched ahead are stale and the rays are marched again, inside the timing).
    ``cycle``: a list of (col, prev, nxt, batch) inputs of the captured composition, resident in HBM, that the replays cycle through
    (the graph copies the step's inputs into its static buffers: new rays every step, as in training).
    Returns {"ms_per_step", "host_ms_per_step"}."""
    from lsenerf_amd.graph import GraphedTrainStep
    step = GraphedTrainStep(model, opt, col, prev, nxt, batch, ray_grads=True, prefetch_march=prefetch,
                            prefetch_fork=os.environ.get("LSE_BENCH_PREFETCH_FORK", "hash_bwd"))
    cycle = cycle or [(col, prev, nxt, batch)]
    k = 0

    def run(n):
        nonlocal k
        for _ in range(n):
            if refresh_from is not None:
                model.update_occupancy_grid(refresh_from + k)
            c, p_, x, b = cycle[k % len(cycle)]
            kw = {"next_bundles": cycle[(k + 1) % len(cycle)][:3]} if prefetch else {}
            step(c, p_, x, b, **kw)
            k += 1
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    step.check_overflow()
    step.close()
    return {"ms_per_step": ms, "host_ms_per_step": host / steps * 1e3,
            "note": "one hipGraph replay per step (sampler with device-side counts -> field -> volume rendering -> loss epilogue -> "
                    "backward -> Adam); occupancy refresh eager between replays" +
                    ("; the marcher of the NEXT step's rays runs on a side stream inside the same graph (two graphs alternate "
                     "between two sample buffers)" if prefetch else "")}


def hash_bwd_request_floor(x01, meta):
    """Distinct 64-byte lines of the gradient table that each 64-sample window (one wave of the hash backward) touches, summed over the
    levels: the fewest float-atomic requests the kernel can send for these positions (a straight ray does not revisit fine cells, so a
    longer window merges 1 - 2 % more).  tcnn's indexing: dense levels x + y*res + z*res^2, hashed levels
    x ^ y*2654435761 ^ z*805459861 (mod level size); 8 entries of 2 floats per line."""
    n = x01.shape[0]
    total = 0
    for l in range(len(meta.scales)):
        res, size = meta.resolutions[l], meta.offsets[l + 1] - meta.offsets[l]
        p0 = (x01.double() * meta.scales[l] + 0.5).float().floor().to(torch.int64)      # fmaf(scale, x, 0.5): one rounding, as the kernels
        cols = []
        for c in range(8):
            qx, qy, qz = p0[:, 0] + (c & 1), p0[:, 1] + ((c >> 1) & 1), p0[:, 2] + ((c >> 2) & 1)
            idx = qx + qy * res + qz * res * res if res ** 3 <= size else (qx ^ (qy * 2654435761) ^ (qz * 805459861)) & 0xFFFFFFFF
            cols.append((idx % size) >> 3)
        ln = torch.stack(cols, 1)
        if n % 64:
            ln = torch.cat([ln, ln[-1:].expand(64 - n % 64, 8)])
        w = ln.reshape(-1, 512).sort(dim=1).values
        total += int((w[:, 1:] != w[:, :-1]).sum()) + w.shape[0]
        del ln, w, cols, p0
    return total


def sphere_rays(R, gen):
    """SURVEY.md 8d ray distribution: origins uniform on the sphere of radius 1.5, aimed at uniform targets in [-0.5, 0.5]^3."""
    o = torch.randn(R, 3, generator=gen)
    o = 1.5 * o / o.norm(dim=-1, keepdim=True)
    d = (torch.rand(R, 3, generator=gen) - 0.5) - o
    return o, d / d.norm(dim=-1, keepdim=True)


def context_default_config(device, steps=20, warmup=6):
    """The regime training runs in after the first few hundred steps -- the reference's DEFAULT configuration of the path:
    cone 0.004, alpha_thre 0.01, early-stop 1e-4, stratified jitter, visibility pre-pass (sigma_fn) on, 4-level 128^3 grid
    carved by the reference's own update rule, SURVEY 8d rays, grid refresh every 16 steps inside the timing.  Reported as
    context next to the contract number (a synthetic 'trained-like' field: density head biased so that a band is opaque)."""
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    from lsenerf_amd.optim import FlatAdam, FlatParams
    torch.manual_seed(96)
    model = LSENeRFModel(LSENeRFModelConfig(), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(device).train()
    with torch.no_grad():
        model.field.mlp_base_grid.params.mul_(3000.0)
        model.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
    flat = FlatParams(model.get_param_groups()["fields"])
    opt = FlatAdam(flat, lr=1e-3, eps=1e-15)
    g = torch.Generator().manual_seed(7)
    R = RAYS_PER_GPU
    o, d = sphere_rays(R, g)
    rb = RayBundle(origins=o.to(device).requires_grad_(True), directions=d.to(device).requires_grad_(True),
                   camera_indices=torch.zeros(R, 1, dtype=torch.long, device=device),
                   metadata={"appearance_id": torch.randint(0, 64, (R,), generator=g).to(device)})
    target = torch.rand(R, 3, generator=g).to(device)
    refresh = model.update_occupancy_grid
    for s in range(0, 64, 16):
        refresh(s)
    occ = float(model.occupancy_grid.binaries.float().mean())
    est = model.occupancy_grid

    def step(i):
        refresh(65 + i)
        opt.zero_grad()
        rb.origins.grad = rb.directions.grad = None
        out = model.exec_get_outputs(rb)
        model.fused_loss_dict({"col_out": out, "prev_out": None, "next_out": None},
                              {"col_batch": {"image": target}, "evs_batch": None})["rgb_loss"].backward()
        opt.step()
        step.last = out

    ms, kern, launches = _timed_steps(step, steps, warmup)
    kept = int(step.last["num_samples_per_ray"].sum())
    _drop_autograd_graphs(step)      # (the eager outputs own their autograd graph: see _drop_autograd_graphs)
    snap = _snapshot(model, opt)
    graphed = _graphed_timing(model, opt, rb, None, None, {"col_batch": {"image": target}, "evs_batch": None}, steps, warmup,
                              refresh_from=320)
    graphed["rays_per_s"] = R / (graphed["ms_per_step"] * 1e-3)
    _restore(model, opt, snap)
    graphed["marcher_prefetched"] = _graphed_timing(model, opt, rb, None, None, {"col_batch": {"image": target}, "evs_batch": None},
                                                    steps, warmup, refresh_from=320, prefetch=True)
    graphed["marcher_prefetched"]["rays_per_s"] = R / (graphed["marcher_prefetched"]["ms_per_step"] * 1e-3)
    # candidates before culling: one more marcher call (untimed)
    with torch.no_grad():
        cand = est.sampling(rb.origins.detach(), rb.directions.detach(), sigma_fn=None, near_plane=0.05, far_plane=1e3,
                            render_step_size=model.config.render_step_size, stratified=True, cone_angle=0.004,
                            alpha_thre=0.0, early_stop_eps=0.0, return_packed=True)[1].shape[0]
    return {"workload": "default-config: cone 0.004, alpha_thre 0.01, early_stop 1e-4, sigma_fn pre-pass on, carved 4-level 128^3 "
                        "grid, SURVEY-8d rays, grid refresh every 16 steps inside the timing", "rays": R, "steps": steps,
            "ms_per_step": ms, "rays_per_s": R / (ms * 1e-3), "occupied_fraction": occ,
            "candidate_samples_per_ray": cand / R, "samples_per_ray_after_culling": kept / R,
            # per-sample cost of the scatter in THIS regime (survivors spread over the uncontracted interior with growing steps:
            # ~2x the distinct table lines per sample of M-march), for comparison with the headline's
            # roofline.kernel_ms / samples_per_step
            "hash_bwd_ns_per_sample": kern.get("lse_hash_bwd", 0.0) * 1e6 / max(kept, 1),
            "graphed": graphed, "kernel_ms_per_step": kern, "launches_per_step": launches}


def context_inside_box(device, steps=12, warmup=4):
    """The round-1/2 headline workload as context: origins inside the box, 4-level grid, exactly 1024 samples per ray."""
    from lsenerf_amd.optim import FlatAdam, FlatParams
    model, sets, _ = build_workload(device, seed=1000, kind="inside")
    rb, target, jitter = sets[0]
    flat = FlatParams(model.get_param_groups()["fields"])
    opt = FlatAdam(flat, lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=200000)
    last = {}

    def step(i):
        last["n"], _ = train_step(model, rb, target, jitter, opt, 1)

    ms, kern, _ = _timed_steps(step, steps, warmup)
    n = int(last["n"])
    return {"workload": "M-march as benchmarked in rounds 1-2: origins in [-0.5,0.5]^3, 4-level 128^3 grid fully occupied, per-ray "
                        "t_max capped -> exactly 1024 samples per ray (65 % of them in the contracted shell, where consecutive "
                        "samples share fine cells: the friendlier input)",
            "samples_per_step": n, "ms_per_step": ms, "rays_per_s": RAYS_PER_GPU / (ms * 1e-3), "kernel_ms_per_step": kern,
            "hash_bwd_alg_GBps": HASH_BYTES_PER_SAMPLE * n / (kern.get("lse_hash_bwd", 1e9) * 1e-3) / 1e9}


def context_composition(device, kind, steps=16, warmup=6):
    """The step LSENeRF actually trains with (R:lse_nerf/lse_pipeline.py:110-145), as ONE packed pass
    (LSENeRFModel.train_step_bundles), in the reference's default configuration of the path (cone 0.004, alpha_thre 0.01,
    pre-pass on, carved 4-level 128^3 grid, refresh every 16 steps inside the timing; trained-like synthetic field):
      kind "cfg2": colour + previous-event + next-event bundle, 2316 / 597 / 597 rays (R:lse_nerf/lse_datamanager.py:135-144),
                   co_map routing, identity rgb mapper, powpow event mapper, learned ThreeToOne, rgb MSE + log-intensity event loss
                   (R:exp_configs/lsenerf_config.sh);
      kind "cfg3": cfg 2 + the per-frame appearance embedding of the EVIMOv2 preset (evs_emb: 512 x 32 table, one id per ray;
                   R:exp_configs/lsenerf_emb_config.sh:19, R:lse_nerf/lse_embeddings.py:19-44);
      kind "cfg4": BAD-NeRF: rgb only, 878 pixels x 4 virtual cameras = 3512 rays averaged per pixel (deblur), gradients w.r.t.
                   every ray's origin and direction for the pose optimiser (R:exp_configs/BADNERF_config.sh)."""
    from lsenerf_amd import LSEEmbeddingConfig, LSENeRFModel, LSENeRFModelConfig, RayBundle
    from lsenerf_amd.optim import FlatAdam, FlatParams
    torch.manual_seed(96)
    n_emb = 64
    if kind == "cfg2":
        cfg = LSENeRFModelConfig(use_mapping=True, mapping_method="identity", map_mode="co_map", evs_mapping_method="powpow")
        sizes = (2316, 597, 597)
    elif kind == "cfg3":     # R:exp_configs/lsenerf_emb_config.sh: cfg 2's routing + one learned 32-d embedding per training frame
        cfg = LSENeRFModelConfig(use_mapping=True, mapping_method="identity", map_mode="co_map", evs_mapping_method="powpow",
                                 embed_config=LSEEmbeddingConfig(embedding_type="evs_emb"))
        sizes, n_emb = (2316, 597, 597), 512
    else:
        cfg = LSENeRFModelConfig(use_mapping=False, map_mode="None", evs_mapping_method="None", rgb_loss_type="deblur")
        sizes = (878 * 4, 0, 0)
    model = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=n_emb).to(device).train()
    with torch.no_grad():
        model.field.mlp_base_grid.params.mul_(3000.0)
        model.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
    flat = FlatParams(model.get_param_groups()["fields"])
    opt = FlatAdam(flat, lr=1e-3, eps=1e-15)
    g = torch.Generator().manual_seed(7)

    def bundle(o, d):
        n = o.shape[0]
        return RayBundle(origins=o.to(device).requires_grad_(True), directions=d.to(device).requires_grad_(True),
                         camera_indices=torch.zeros(n, 1, dtype=torch.long, device=device),
                         metadata={"appearance_id": torch.randint(0, n_emb, (n,), generator=g).to(device)})
    if kind in ("cfg2", "cfg3"):
        o, d = sphere_rays(sizes[0], g)
        col = bundle(o, d)
        o, d = sphere_rays(sizes[1], g)
        prev = bundle(o, d)
        o2 = o + 0.01 * torch.randn(o.shape, generator=g)          # the same pixels from the neighbouring event camera
        d2 = d + 0.01 * torch.randn(d.shape, generator=g)
        nxt = bundle(o2, d2 / d2.norm(dim=-1, keepdim=True))
        batch = {"col_batch": {"image": torch.rand(sizes[0], 3, generator=g).to(device)},
                 "evs_batch": {"image": ((torch.rand(sizes[1], 1, generator=g) - 0.5) * 0.4).to(device)}}
    else:
        o, d = sphere_rays(878, g)
        o4 = (o[:, None, :] + 0.005 * torch.randn(878, 4, 3, generator=g)).reshape(-1, 3)
        d4 = d[:, None, :] + 0.005 * torch.randn(878, 4, 3, generator=g)
        col, prev, nxt = bundle(o4, (d4 / d4.norm(dim=-1, keepdim=True)).reshape(-1, 3)), None, None
        batch = {"col_batch": {"image": torch.rand(878, 3, generator=g).to(device)}, "evs_batch": None}
    for s_ in range(0, 64, 16):
        model.update_occupancy_grid(s_)
    last = {}

    def step(i):
        model.update_occupancy_grid(65 + i)
        opt.zero_grad()
        for b in (col, prev, nxt):
            if b is not None:
                b.origins.grad = b.directions.grad = None
        out, losses, _ = model.train_step_bundles(col, prev, nxt, batch)
        sum(losses.values()).backward()
        opt.step()
        last["out"] = out

    from lsenerf_amd import ops as _ops

    def timed(deferred):
        model.deferred_counts = deferred
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        _ops.SYNC_STATS.update(seconds=0.0, count=0)
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, host, _ops.SYNC_STATS["seconds"], _ops.SYNC_STATS["count"]
    # (a) the synchronising eager step (two read-backs of the sampler's counts), (b) the eager step with device-side counts
    ms_sync, host_sync, blocked, n_sync = timed(False)
    ms, host_issue, _, n_sync_deferred = timed(True)
    assert n_sync_deferred == 0
    kern, launches = _event_pass(step, warmup + steps, 8, None)
    kept = sum(int(v["num_samples_per_ray"].sum()) for v in last["out"].values() if v is not None)
    _drop_autograd_graphs(last)
    rays = sum(sizes)
    snap = _snapshot(model, opt)
    graphed = _graphed_timing(model, opt, col, prev, nxt, batch, steps, warmup, refresh_from=320)
    graphed["rays_per_s"] = rays / (graphed["ms_per_step"] * 1e-3)
    if kind in ("cfg2", "cfg3"):      # (cfg 4's rays come from poses the step itself updates: nothing to march ahead)
        _restore(model, opt, snap)
        graphed["marcher_prefetched"] = _graphed_timing(model, opt, col, prev, nxt, batch, steps, warmup, refresh_from=320,
                                                        prefetch=True)
        graphed["marcher_prefetched"]["rays_per_s"] = rays / (graphed["marcher_prefetched"]["ms_per_step"] * 1e-3)
    return {"workload": {"cfg2": "colour + prev + next event bundle (2316 / 597 / 597 rays), co_map routing, rgb + event loss",
                         "cfg3": "cfg 2 with a per-frame appearance embedding (evs_emb, 512 embeddings x 32, R:exp_configs/lsenerf_emb_config.sh): "
                                 "colour + prev + next event bundle (2316 / 597 / 597 rays), co_map routing, rgb + event loss",
                         "cfg4": "BAD-NeRF deblur: 878 pixels x 4 virtual cameras = 3512 rays, rgb loss on the 4-ray mean, pose gradients"}[kind]
                        + "; ONE packed pass per step (train_step_bundles), reference default sampler configuration, carved grid, "
                          "refresh every 16 steps inside the timing",
            "rays": rays, "steps": steps, "ms_per_step": ms, "rays_per_s": rays / (ms * 1e-3),
            "samples_per_ray_after_culling": kept / rays, "host_issue_ms_per_step": host_issue / steps * 1e3,
            "eager_synchronising": {"ms_per_step": ms_sync, "host_issue_ms_per_step": host_sync / steps * 1e3,
                                    "host_blocked_in_count_readbacks_ms_per_step": blocked / steps * 1e3,
                                    "count_readbacks_per_step": n_sync / steps,
                                    "host_python_ms_per_step": (host_sync - blocked) / steps * 1e3,
                                    "note": "LSENeRFModel.deferred_counts = False: the sampler reads its two counts back"},
            "launches_per_step": sum(launches.values()), "kernel_ms_sum_per_step": sum(kern.values()),
            "graphed": graphed, "kernel_ms_per_step": kern}


def context_m_packed(device, steps=12, warmup=4):
    """SURVEY.md 8d 'M-packed' (kernel roofline inputs): the estimator is bypassed; 4096 rays from the radius-1.5 sphere,
    ray_indices = repeat_interleave(arange(R), 1024), t_starts = 0.05 + k * step, t_ends = t_starts + step."""
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    from lsenerf_amd.optim import FlatAdam, FlatParams
    torch.manual_seed(96)
    cfg = LSENeRFModelConfig(cone_angle=0.0, alpha_thre=0.0)
    model = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(device).train()
    flat = FlatParams(model.get_param_groups()["fields"])
    opt = FlatAdam(flat, lr=1e-2, eps=1e-15)
    g = torch.Generator().manual_seed(96)
    R, S = RAYS_PER_GPU, SAMPLES_PER_RAY
    o, d = sphere_rays(R, g)
    rb = RayBundle(origins=o.to(device).requires_grad_(True), directions=d.to(device).requires_grad_(True),
                   camera_indices=torch.zeros(R, 1, dtype=torch.long, device=device),
                   metadata={"appearance_id": torch.randint(0, 64, (R,), generator=g).to(device)})
    target = torch.rand(R, 3, generator=g).to(device)
    dt = cfg.render_step_size
    ts = (0.05 + dt * torch.arange(S, dtype=torch.float32)).repeat(R).to(device)
    te = ts + dt
    ri = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(device)
    cnt = torch.full((R,), S, dtype=torch.long)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).contiguous().to(device)

    def step(i):
        opt.zero_grad()
        rb.origins.grad = rb.directions.grad = None
        out = model.render_packed(rb, ri, ts, te, packed)
        model.fused_loss_dict({"col_out": out, "prev_out": None, "next_out": None},
                              {"col_batch": {"image": target}, "evs_batch": None})["rgb_loss"].backward()
        opt.step()

    ms, kern, _ = _timed_steps(step, steps, warmup)
    n = R * S
    return {"workload": "M-packed (SURVEY 8d): estimator bypassed, radius-1.5-sphere rays, 1024 fixed-step samples per ray",
            "samples_per_step": n, "ms_per_step": ms, "rays_per_s": R / (ms * 1e-3), "kernel_ms_per_step": kern,
            "hash_fwd_alg_GBps": HASH_BYTES_PER_SAMPLE * n / (kern.get("lse_hash_fwd", 1e9) * 1e-3) / 1e9,
            "hash_bwd_alg_GBps": HASH_BYTES_PER_SAMPLE * n / (kern.get("lse_hash_bwd", 1e9) * 1e-3) / 1e9}


def cpu_baseline(seconds_budget=25.0):
    """The reference's torch-native CPU field (nerfstudio HashEncoding.pytorch_fwd + nn.Linear MLPs + torch volrend,
    restated in oracle/) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle.field import FieldOracle
    from oracle.model import ModelOracle, cpu_train_step_packed
    cores = min(os.cpu_count() or 1, 16)        # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(cores)
    f = FieldOracle("torch", num_embeddings=64, seed=96)
    m = ModelOracle(f, cone_angle=0.0, alpha_thre=0.0)
    # a bounded sample of the M workload itself ("the CPU run processes M in ray chunks with gradient accumulation"):
    # SURVEY-8d sphere rays x 1024 samples, processed in chunks of 128 rays (pytorch_fwd materialises 8 x [N,16,2] f32 per chunk =
    # 134 MB; larger chunks are faster on the CPU: 84 / 155 / 178 rays/s for chunks of 8 / 32 / 64 rays on 8 cores), gradients
    # accumulated over the chunks, one Adam step per pass.  How many of the 4096 rays a pass takes is set by a 256-ray probe pass so
    # that one pass costs ~8 s on this box's cores -- the whole batch where the CPU is fast enough (the MI355X box's 16-core share:
    # ~540 rays/s) -- and rays/s = rays / pass time (per-ray work is independent; the per-step constant, Adam over the 67 MB
    # torch-layout table, is paid once per pass)
    CHUNK = 128
    g = torch.Generator().manual_seed(0)
    o_all, d_all = sphere_rays(RAYS_PER_GPU, g)
    target_all = torch.rand(RAYS_PER_GPU, 3, generator=g)
    aid_all = torch.randint(0, 64, (RAYS_PER_GPU,), generator=g)
    step = m.render_step_size
    state = {}

    def one_pass(R):
        ri = torch.repeat_interleave(torch.arange(R), SAMPLES_PER_RAY)
        ts = (0.05 + step * torch.arange(SAMPLES_PER_RAY, dtype=torch.float32)).repeat(R)
        t0 = time.time()
        cpu_train_step_packed(m, o_all[:R], d_all[:R], ri, ts, ts + step, target_all[:R], aid_all[:R], state, ray_chunk=CHUNK)
        return time.time() - t0

    one_pass(CHUNK)                                           # warm-up (allocator, thread pool, optimizer state)
    probe = one_pass(2 * CHUNK)
    R = int(min(RAYS_PER_GPU, max(2 * CHUNK, (8.0 / probe) * 2 * CHUNK)) // CHUNK * CHUNK)
    times = []
    t_all = time.time()
    while len(times) < 3 and (not times or (time.time() - t_all) + times[-1] < seconds_budget):
        times.append(one_pass(R))
    med = sorted(times)[len(times) // 2]
    return {"value": R / med, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"{R} of the {RAYS_PER_GPU} SURVEY-8d sphere rays x {SAMPLES_PER_RAY} samples in chunks of {CHUNK} rays with gradient "
                      f"accumulation, {len(times)} timed train steps (median; {med:.1f} s each)"
                      + ("" if R == RAYS_PER_GPU else ", rate extrapolated linearly to the full batch")
                      + f"; torch-native field (hash pytorch_fwd + nn.Linear MLPs + torch volrend + Adam), fp32, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)         # "Warm-up 20 steps, time 100 steps"
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-context", action="store_true", help="skip the default-config / M-packed context runs")
    ap.add_argument("--no-atomic-floor", action="store_true",
                    help="skip roofline.atomic (its request count sorts 16 x 8 x N indices with torch kernels AFTER the timed region: "
                         "left out of rocprofv3 passes so that the kernel statistics hold the step's kernels only)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="self-launched ranks are terminated after this many seconds")
    args = ap.parse_args()

    # -- self-launch (R:train.py:171-234): `python bench.py --gpus N` without a launcher starts its own N ranks ------------------
    if IS_LAUNCHING_PARENT:
        sys.exit(_load_launcher().launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                               timeout=args.launch_timeout,
                                               stdout_filter=lambda line: line.lstrip().startswith("{")))

    from lsenerf_amd import _lib, ops, dist as ldist
    from lsenerf_amd.optim import FlatAdam, FlatParams
    import torch.distributed as tdist

    # RCCL ("nccl") over xGMI on a real multi-GPU node; LSE_BENCH_BACKEND=gloo lets two ranks share one GPU to rehearse
    # the multi-rank control path on a single-GPU box (the collective then stages through the host).
    backend = os.environ.get("LSE_BENCH_BACKEND", "nccl")
    rank, world, local = ldist.init_from_env(backend)
    if world != args.gpus:      # (a bare `--gpus N` never gets here: it became the launching parent above)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    ranks_seen = tdist.get_world_size() if tdist.is_initialized() else 1
    if ranks_seen != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {ranks_seen} rank(s)")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    _lib.load()

    model, sets, set_counts = build_workload(device, seed=1000 + rank, n_sets=N_RAY_SETS)
    rb, target, jitter = sets[0]
    flat = FlatParams(model.get_param_groups()["fields"], total_multiple=world * 64)
    ldist.broadcast_params(flat.data)
    opt = FlatAdam(flat, lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=200000)
    # N > 1: the gradient all-reduce of step k is hidden behind the ray marcher of step k+1 (GradPipeline).
    # LSE_BENCH_EXCHANGE=plain selects the single blocking all-reduce; =overlap / =split the two-launch hash backward of
    # dist.OverlappedGradExchange (measured at N = 1: the second launch costs 0.28 ms, more than the exchange it hides).
    # =sharded: dist.ShardedAdamExchange (reduce-scatter, Adam on 1/W of the buffer, all-gather).
    # =graphed: the step up to the backward pass replayed as ONE HIP graph, then a blocking all-reduce, then Adam.
    exchange, pipeline, sharded = None, None, None
    mode = os.environ.get("LSE_BENCH_EXCHANGE", "pipelined" if world > 1 else "plain")
    if mode == "sharded":
        sharded = ldist.ShardedAdamExchange(flat, lr=1e-2, eps=1e-15)
    if world > 1:      # rank 0's grid after every refresh (DDP's buffer broadcast, R:lse_nerf/lse_pipeline.py:97); M-march's grid is
        ldist.attach_grid_sync(model.occupancy_grid)      # fully occupied and never refreshed, the hook is part of the template
    if mode == "pipelined":
        pipeline = ldist.GradPipeline(opt, world).attach(model.occupancy_grid)
    if mode == "pipelined_sharded":     # reduce-scatter behind backward, Adam on 1/W + all-gather behind the next step's marcher
        pipeline = ldist.GradPipeline(opt, world, sharded=ldist.ShardedAdamExchange(flat, lr=1e-2, eps=1e-15)).attach(model.occupancy_grid)
    if mode in ("overlap", "split"):
        grid = model.field.mlp_base_grid
        exchange = ldist.OverlappedGradExchange(flat, grid.params, grid.meta.offsets, split_level=min(6, grid.meta.n_levels - 1))
        exchange.install()

    graphed = None
    if mode == "graphed":       # replay (sampler .. backward) -> all-reduce -> Adam; the pre-pass is off as SURVEY 8d prescribes
        from lsenerf_amd.graph import GraphedTrainStep
        model.sampler.density_fn = None
        graphed = GraphedTrainStep(model, opt, rb, None, None, {"col_batch": {"image": target}, "evs_batch": None}, ray_grads=True,
                                   jitter="input", optimizer_in_graph=False)
    exposed = [] if world > 1 else None
    if pipeline is not None:
        pipeline.exposed_events = exposed

    step_no = 0

    def one_step():
        # every step trains on the next of the N_RAY_SETS resident (rays, targets, stratified offsets) sets
        nonlocal step_no
        rb_k, target_k, jitter_k = sets[step_no % len(sets)]
        step_no += 1
        return train_step(model, rb_k, target_k, jitter_k, opt, world, exchange, pipeline, sharded, graphed, exposed)

    n_samples = 0
    for _ in range(args.warmup):
        n_samples, _ = one_step()
    if pipeline is not None:
        pipeline.flush()
    if exposed is not None:
        del exposed[:]

    # timed region: barrier + synchronize on both sides.  Only the two candidates for the dominant kernel (the hash
    # gather / scatter) carry HIP events here -- on the stream they are launched on -- because instrumenting every entry
    # point perturbs the step (see _timed_steps); the full per-kernel breakdown comes from a second, instrumented pass.
    _lib.TIMING = {"names": {"lse_hash_bwd", "lse_hash_fwd"}, "events": []}
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_samples, loss = one_step()
    if pipeline is not None:
        pipeline.flush()           # the K-th all-reduce + Adam belong to the timed K steps
    host_issue = time.perf_counter() - t0      # host time to ISSUE the K steps (the sampler's one sync per step included)
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
    elapsed = time.perf_counter() - t0
    timing = _lib.TIMING
    _lib.TIMING = None
    elapsed = ldist.max_over_ranks(elapsed, device)
    n_last = int(n_samples)                    # the device-side count of the last step, read once, after the timed region
    assert n_last == set_counts[(step_no - 1) % len(sets)], (n_last, set_counts)      # the timed steps marched what the set-up counted
    # samples per step = the mean over the sets the K timed steps actually cycled through (the counts differ by < 1 % between sets)
    timed_sets = [(args.warmup + i) % len(sets) for i in range(args.steps)]
    n_samples = int(round(sum(set_counts[k] for k in timed_sets) / max(len(timed_sets), 1)))
    model.occupancy_grid.check_deferred_overflow()
    # what the gradient exchange still costs per step after whatever overlap the mode has: stream time between the two events that
    # bracket the wait for (or, in the blocking modes, the whole of) the collective; max over ranks like the step time
    exchange_exposed_ms = None
    if exposed is not None:
        tot = sum(e0.elapsed_time(e1) for e0, e1 in exposed) / max(args.steps, 1)
        exchange_exposed_ms = ldist.max_over_ranks(tot, device)
        exposed = None
        if pipeline is not None:
            pipeline.exposed_events = None

    per_kernel = {}
    for name, e0, e1 in timing["events"]:
        per_kernel.setdefault(name, []).append(e0.elapsed_time(e1))
    dom_ms_all = {k: sum(v) / len(v) for k, v in per_kernel.items()}            # mean launch duration inside the timed region
    # per-entry-point breakdown: separate instrumented pass (same workload, not part of the timed K steps)
    bsteps = min(args.steps, 8)
    if graphed is not None:     # (a replayed graph has no host hooks between its kernels: the breakdown pass runs the eager step)
        graphed.close()
        graphed = None
        model.deferred_counts = True
    kern_ms, launches = _event_pass(lambda i: train_step(model, *sets[i % len(sets)], opt, world, exchange, pipeline, sharded),
                                    0, bsteps, None)
    if pipeline is not None:
        pipeline.flush()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays_per_s = world * RAYS_PER_GPU / (elapsed / args.steps)
        assert abs(n_samples / (RAYS_PER_GPU * SAMPLES_PER_RAY) - 1.0) < 0.01, f"workload drifted: {n_samples} samples"
        assert all(abs(c / (RAYS_PER_GPU * SAMPLES_PER_RAY) - 1.0) < 0.03 for c in set_counts), set_counts
        dom_src = "HIP events around the entry point inside the timed region"
        if not dom_ms_all:      # exchange mode "graphed": a replayed graph has no host hooks between its kernels
            dom_ms_all = {k: kern_ms[k] for k in ("lse_hash_bwd", "lse_hash_fwd") if k in kern_ms}
            dom_src = "HIP events in the separate instrumented eager pass (the timed region replays a HIP graph)"
        dom = max(("lse_hash_bwd", "lse_hash_fwd"), key=lambda k: dom_ms_all.get(k, 0.0))
        dom_ms = dom_ms_all[dom]
        achieved = HASH_BYTES_PER_SAMPLE * n_samples / (dom_ms * 1e-3) / 1e9
        b_step = n_samples * BYTES_PER_SAMPLE_STEP + 8 * 4 * flat.numel
        traffic, sq_busy, stale, sent_requests = None, None, None, None
        try:   # committed PMC summary of the same kernels (bench.py cannot run rocprofv3 on itself); see profiles/pmc_traffic.json.
            # Each group of counters is stamped with a digest of the kernel sources it was measured on (lsenerf_amd/provenance.py);
            # numbers from other sources than this tree's are NOT reported: null fields + "traffic_stale": true
            from lsenerf_amd import provenance
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            stale = {"hash": not provenance.counters_current(pmc, "hash"), "mlp": not provenance.counters_current(pmc, "mlp")}
            if not stale["hash"]:
                traffic = pmc.get(dom, {}).get("bytes")
                sent_requests = pmc.get("lse_hash_bwd", {}).get("atomic_requests")      # TCC_EA0_ATOMIC_sum per launch, same workload
            if not stale["mlp"]:
                sq_busy = pmc.get("matrix_core_busy")
        except OSError:
            pass
        atomic = None
        if dom == "lse_hash_bwd" and not args.no_atomic_floor and world == 1:      # (N = 1 only, like cpu_baseline: rank 0 alone would lag the others)
            # second roofline of the dominant kernel: it scatters with float atomics, which gfx950 executes at the memory side at a
            # chip-wide REQUEST rate (MI355X_MICROARCH.md "Global float atomics": ~1.3 TB/s of 64-byte requests = ~20 G requests/s;
            # tools/micro/atomic_gran.hip: 21 G/s), far below the HBM byte rate the contract's `frac` is priced against
            with torch.no_grad():
                cfg_ = model.config
                ri_, ts_, te_, pk_ = model.occupancy_grid.sampling(
                    rb.origins.detach(), rb.directions.detach(), near_plane=cfg_.near_plane, far_plane=cfg_.far_plane,
                    render_step_size=cfg_.render_step_size, stratified=True, jitter=jitter, return_packed=True)[:4]
                x01_ = ops.positions(rb.origins.detach(), rb.directions.detach(), ri_, ts_, te_, pk_, True, None)[0]
                req = hash_bwd_request_floor(x01_, model.field.mlp_base_grid.meta)
                del ri_, ts_, te_, pk_, x01_
            req = req / set_counts[0] * n_samples        # (counted on the first ray set; the launches average over all sets)
            atomic = {"bound": "memory-side float atomics (requests of <= 64 B)", "request_floor_per_launch": req,
                      "requests_per_sample": req / n_samples, "peak_requests_per_s": 21e9,
                      "achieved_requests_per_s": req / (dom_ms * 1e-3), "frac": req / (dom_ms * 1e-3) / 21e9,
                      # what the kernel SENDS (committed rocprofv3 --pmc TCC_EA0_ATOMIC_sum pass of this workload on this tree's
                      # kernel sources, null when the tree has moved on): the memory side is busy with these, not with the floor
                      "sent_requests_per_launch": sent_requests,
                      "sent_over_floor": (sent_requests / req) if sent_requests else None,
                      "sent_requests_per_s": (sent_requests / (dom_ms * 1e-3)) if sent_requests else None,
                      "sent_frac": (sent_requests / (dom_ms * 1e-3) / 21e9) if sent_requests else None,
                      "note": "floor = distinct 64-B table lines per 64-sample wave window, all 16 levels; `frac` prices the FLOOR against "
                              "the request rate (useful requests per second), `sent_frac` the requests the kernel actually sends "
                              "(utilisation of the memory-side atomic units; the cache-free kernel saturates them at 20.5 G/s, "
                              "profiles/r05_hash_bwd_ablation.txt)"}
        line = {
            "metric": "train-step rays/sec (4096-ray x 1024-sample batch)", "value": rays_per_s, "unit": "rays/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "backend": (tdist.get_backend() if tdist.is_initialized() else None),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "M-march (SURVEY 8d): 4096 rays x ~1024 samples per GPU, L=16 T=2^19 hash grid + 64-wide fused MLPs",
                       "step_roofline_frac": b_step / (ms_per_step * 1e-3) / HBM_PEAK, "roofline_frac": achieved / (HBM_PEAK / 1e9),
                       "workload_detail": "SURVEY 8d exactly: LSENeRF scene field (L=16 hash grid T=2^19 F=2, 64-wide fused MLPs, "
                                          "SH4, 32-d appearance embedding); 4096 rays/GPU from the radius-1.5 sphere aimed into "
                                          "[-0.5,0.5]^3, one-level 128^3 occupancy grid fully occupied, cone 0, no culling, constant step "
                                          "chosen for 1024 samples per ray on average; sampler + fwd + bwd + Adam, grads w.r.t. rays "
                                          "included; every step trains on the next of %d resident (rays, targets, stratified offsets) "
                                          "sets" % len(sets),
                       "ray_sets": len(sets), "samples_per_set": set_counts,
                       "rays_per_gpu": RAYS_PER_GPU, "samples_per_ray": n_samples / RAYS_PER_GPU, "samples_per_step": n_samples,
                       "render_step_size": model.config.render_step_size,
                       "parallelism": f"dp{world}", "grad_exchange": mode,
                       "exchange_exposed_ms_per_step": exchange_exposed_ms,
                       "exchange_bytes_per_step": 4 * flat.numel if world > 1 else 0},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / (HBM_PEAK / 1e9), "traffic": traffic,
                         "traffic_stale": None if stale is None else stale["hash"],
                         "kernel_ms": dom_ms, "kernel_ms_source": dom_src,
                         "algorithmic_bytes_per_launch": HASH_BYTES_PER_SAMPLE * n_samples,
                         "atomic": atomic},
            "step_roofline": {"algorithmic_bytes_per_step": b_step,
                              "achieved_GBps": b_step / (ms_per_step * 1e-3) / 1e9 * 1.0,
                              "frac_of_hbm_peak": b_step / (ms_per_step * 1e-3) / HBM_PEAK},
            "mfma": (lambda t_ms: {
                "kernels": "lse_mlp_fwd+lse_mlp_bwd", "ms": t_ms,
                "note": "f32-equivalent arithmetic on the bf16 matrix cores (three bf16 pieces per operand, six piece products, f32 "
                        "accumulate; the backward recomputes the hidden layers): `executed_bf16_*` prices the MFMA instructions "
                        "actually issued against the dense bf16 peak; `matrix_core_busy` is SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x "
                        "kernel cycles) from the committed rocprofv3 --pmc summary of the same kernels",
                "executed_bf16_flop_per_sample": MLP_BF16_FLOP_PER_SAMPLE,
                "executed_bf16_TFLOPs": MLP_BF16_FLOP_PER_SAMPLE * n_samples / (1e-3 * t_ms) / 1e12, "bf16_peak_TFLOPs": 2500.0,
                "executed_bf16_frac": MLP_BF16_FLOP_PER_SAMPLE * n_samples / (1e-3 * t_ms) / 2.5e15,
                "matrix_core_busy": sq_busy, "matrix_core_busy_stale": None if stale is None else stale["mlp"]})(
                    kern_ms.get("lse_mlp_fwd", 0) + kern_ms.get("lse_mlp_bwd", 0) + kern_ms.get("lse_mlp_wgrad", 0) + 1e-9),
            "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(kern_ms.items())},
            "kernel_ms_note": "per C-ABI entry point, from a separate instrumented pass of %d steps (event pairs around every "
                              "entry point perturb the step; the timed region instruments the two hash kernels only)" % bsteps,
            "host_issue_ms_per_step": host_issue / args.steps * 1e3,
            "host_issue_note": "host time to issue the K timed steps; no sample count is read back (device-side counts), so this is "
                               "Python + launch time and the host runs ahead of the GPU",
            "loss": float(loss.detach()),
        }
        del loss
        _drop_autograd_graphs({})
        if world == 1 and not args.no_context:
            # the headline workload as one replayed graph (pre-pass off as SURVEY 8d prescribes: no density_fn on the sampler)
            model.sampler.density_fn = None
            cyc = [(rb_k, None, None, {"col_batch": {"image": t_k}, "evs_batch": None}) for rb_k, t_k, _ in sets]
            line["m_march_graphed"] = _graphed_timing(model, opt, rb, None, None, {"col_batch": {"image": target}, "evs_batch": None},
                                                      min(args.steps, 20), 4, cycle=cyc)
            line["m_march_graphed"]["rays_per_s"] = RAYS_PER_GPU / (line["m_march_graphed"]["ms_per_step"] * 1e-3)
            pf = _graphed_timing(model, opt, rb, None, None, {"col_batch": {"image": target}, "evs_batch": None},
                                 min(args.steps, 20), 4, prefetch=True, cycle=cyc)
            pf["rays_per_s"] = RAYS_PER_GPU / (pf["ms_per_step"] * 1e-3)
            line["m_march_graphed"]["marcher_prefetched"] = pf
            del model, flat, opt, cyc
            torch.cuda.empty_cache()
            line["m_march_inside_box"] = context_inside_box(device)
            torch.cuda.empty_cache()
            line["default_config"] = context_default_config(device)
            torch.cuda.empty_cache()
            line["m_packed"] = context_m_packed(device)
            torch.cuda.empty_cache()
            line["cfg2_composition"] = context_composition(device, "cfg2")
            torch.cuda.empty_cache()
            line["cfg3_composition"] = context_composition(device, "cfg3")
            torch.cuda.empty_cache()
            line["cfg4_composition"] = context_composition(device, "cfg4")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
