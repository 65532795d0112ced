"""The whole training step as one replayed HIP graph (lsenerf_amd.graph.GraphedTrainStep): sampler with device-side counts,
visibility pre-pass, field, volume rendering, fused loss epilogue, backward, Adam with staged scalars.  The replayed step must
be the eager step: same losses and gradients on the same rays / targets / jitter, following in-place occupancy refreshes and
the learning-rate schedule from step to step."""
import copy

import pytest
import torch

from tests.util import nmax_err, random_binaries, random_rays

# torch's diagnosis of a captured backward that runs through an autograd node of another stream is an ERROR in this module
# (round 4: green with that warning, one step away from the hipStreamEndCapture crash; lsenerf_amd/graph.py)
pytestmark = [pytest.mark.gpu, pytest.mark.filterwarnings("error:The AccumulateGrad node's stream")]


@pytest.fixture(autouse=True)
def _every_stream_mismatch_counts():
    before = torch.is_warn_always_enabled()
    torch.set_warn_always(True)            # TORCH_WARN_ONCE would report the first occurrence of the process only
    yield
    torch.set_warn_always(before)


def _setup(ray_grads, mappers=("identity", "powpow")):
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    from lsenerf_amd.optim import FlatAdam, FlatParams
    torch.manual_seed(96)
    cfg = LSENeRFModelConfig(grid_levels=2, grid_resolution=32, log2_hashmap_size=15, use_mapping=True, mapping_method=mappers[0],
                             map_mode="co_map", evs_mapping_method=mappers[1])
    models, opts = [], []
    base = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 16)
    with torch.no_grad():
        base.field.mlp_base_grid.params.mul_(300.0)
    for _ in range(2):
        m = copy.deepcopy(base).cuda().train()
        m.occupancy_grid.binaries.copy_(random_binaries(2, 32, 0.5, 3).cuda())
        # occs.mean() = 0.006 < alpha_thre = 0.01: the cap `min(alpha_thre, occs.mean())` is the ACTIVE threshold of the culling
        m.occupancy_grid.occs.copy_(m.occupancy_grid.binaries.flatten().float() * 0.012)
        flat = FlatParams(m.get_param_groups()["fields"])
        models.append(m)
        opts.append(FlatAdam(flat, lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=50))     # a schedule that moves visibly
    g = torch.Generator().manual_seed(3)
    sizes = (200, 50, 50)

    def batch_of(seed):
        bundles = []
        for i, n in enumerate(sizes):
            o, d = random_rays(n, seed=seed + i)
            bundles.append(RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(n, 1, dtype=torch.long, device="cuda"),
                                     metadata={"appearance_id": torch.randint(0, 16, (n,), generator=g).cuda()}))
        batch = {"col_batch": {"image": torch.rand(sizes[0], 3, generator=g).cuda()},
                 "evs_batch": {"image": ((torch.rand(sizes[1], 1, generator=g) - 0.5) * 0.4).cuda()}}
        return bundles, batch, torch.rand(sum(sizes), generator=g).cuda()
    return models, opts, batch_of


def _eager_step(m, opt, bundles, batch, jit, ray_grads):
    opt.zero_grad()
    bs = bundles
    if ray_grads:
        from lsenerf_amd import RayBundle
        bs = [RayBundle(origins=b.origins.clone().requires_grad_(True), directions=b.directions.clone().requires_grad_(True),
                        camera_indices=b.camera_indices, metadata=b.metadata) for b in bundles]
    _, losses, _ = m.train_step_bundles(*bs, batch, jitter=jit)
    sum(losses.values()).backward()
    grad = opt.flat.grad.clone()
    opt.step()
    return {k: float(v) for k, v in losses.items()}, grad, bs


@pytest.mark.parametrize("ray_grads,mappers", [(False, ("identity", "powpow")), (True, ("identity", "powpow")), (True, ("rgb_mlp", "mlp"))])
def test_graphed_step_equals_eager_step(ray_grads, mappers, monkeypatch):
    """(The third case trains the two MLP intensity mappers of R:lse_nerf/intensity_mappers.py:28-62 inside the captured step: their
    1252 parameters are views of the flat buffer and the fused epilogue adds their gradients straight into the flat gradient.)"""
    from lsenerf_amd import model as M, ops
    from lsenerf_amd.graph import GraphedTrainStep
    monkeypatch.setattr(M.MLP_Mapper, "init_steps", 100)
    monkeypatch.setattr(M.RGB_MLP_Mapper, "init_steps", 100)
    (m_e, m_g), (o_e, o_g), batch_of = _setup(ray_grads, mappers)
    assert m_g._epilogue_desc() is not None                       # the fused epilogue, not the torch route
    b0, batch0, jit0 = batch_of(50)
    step = GraphedTrainStep(m_g, o_g, *b0, batch0, ray_grads=ray_grads, jitter="input")
    assert o_g.step_count == 0 and torch.equal(o_g.flat.data, o_e.flat.data)        # building the graph trains nothing
    ops.SYNC_STATS.update(seconds=0.0, count=0)
    spans = [(o, o + p.numel()) for p, o in zip(o_e.flat.params, o_e.flat.offsets)]
    for it in range(4):
        if it == 2:                      # an occupancy refresh between replays: in place, outside the graph
            for m in (m_e, m_g):
                m.update_occupancy_grid(0)
            # the refresh writes `occs` through raw pointers; the device-side copy of occs.mean() (the cap of the alpha threshold)
            # must follow it: here the mean drops below alpha_thre = 0.01, so a stale cap would change the culling
            mean_now = float(m_g.occupancy_grid.occs.mean())
            assert 0.0 < mean_now < 0.01 and abs(mean_now - 0.006) > 1e-5            # the refresh moved it
            assert float(m_g.occupancy_grid.__dict__["_occ_mean_dev"]) == mean_now    # (read as is: no recomputation here)
        bundles, batch, jit = batch_of(60 + 10 * it)
        sync_before = ops.SYNC_STATS["count"]
        l_g = {k: v for k, v in step(*bundles, batch, jitter=jit).items()}
        assert ops.SYNC_STATS["count"] == sync_before                                    # no sample count visited the host
        g_g = o_g.flat.grad.clone()
        l_g = {k: float(v) for k, v in l_g.items()}
        l_e, g_e, bs = _eager_step(m_e, o_e, bundles, batch, jit, ray_grads)
        assert set(l_g) == set(l_e) == {"rgb_loss", "event_loss"}
        for k in l_e:
            assert abs(l_g[k] - l_e[k]) <= 2e-5 * max(1.0, abs(l_e[k])), (it, k, l_g[k], l_e[k])
        if it == 0:      # identical parameters: the gradients agree to float-atomic noise, tensor by tensor
            for a, b in spans:
                assert nmax_err(g_g[a:b], g_e[a:b], 1e-12) < 3e-5, (a, b)
            if ray_grads:
                rg = step.ray_grads
                for key, be in zip(("col", "prev", "next"), bs):
                    assert nmax_err(rg[key][0], be.origins.grad, 1e-12) < 3e-5 and nmax_err(rg[key][1], be.directions.grad, 1e-12) < 3e-5
        assert o_g.step_count == o_e.step_count == it + 1
    step.check_overflow()
    # after four Adam steps (eps 1e-15 turns summation noise into +-lr steps) the parameters still agree
    # except for a small fraction of noise-dominated elements
    d = (o_g.flat.data - o_e.flat.data).abs()
    scale = float(o_e.flat.data.abs().max())
    assert float((d > 1e-5 * scale).float().mean()) < 0.02
    # composition changes are refused loudly
    bundles, batch, jit = batch_of(99)
    with pytest.raises(ValueError, match="fixed"):
        step(bundles[0], None, None, batch, jitter=jit)
    step.close()
    assert m_g.deferred_counts is True and m_g.deferred_max_slots == 1 << 24


def test_a_replay_survives_eager_passes_with_other_ray_counts_in_between():
    """A captured step bakes the ADDRESS of GlobalEmbedding's all-zero row-index vector into its row-bias / embedding-gradient
    kernels and keeps no reference of its own.  Round 4 cached ONE vector per module and replaced it when a training-mode call came
    with another ray count -- the old one went back to the allocator, and the next replay gathered and scattered embedding rows
    through whatever was allocated over it.  The vectors are grow-only now and never freed (field.GlobalEmbedding.ray_indices):
    capture, run eager training-mode passes with fewer AND more rays, poison the allocator with NaN and out-of-range integers,
    replay, and compare with the eager step."""
    from lsenerf_amd import RayBundle
    from lsenerf_amd.field import GlobalEmbedding
    from lsenerf_amd.graph import GraphedTrainStep
    (m_e, m_g), (o_e, o_g), batch_of = _setup(False)
    assert isinstance(m_g.field.embedding_appearance, GlobalEmbedding)
    b0, batch0, jit0 = batch_of(50)
    step = GraphedTrainStep(m_g, o_g, *b0, batch0, jitter="input")
    kept = m_g.field.embedding_appearance.__dict__["_zero_idx"][("cuda", torch.cuda.current_device())]
    captured_ptr = kept[-1].data_ptr()
    for n in (77, 3000):             # fewer rays: a prefix of the kept vector; more rays: a NEW, longer vector -- the old one stays
        o, d = random_rays(n, seed=n)
        rb = RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(n, 1, dtype=torch.long, device="cuda"),
                       metadata={"appearance_id": torch.zeros(n, dtype=torch.long, device="cuda")})
        with torch.no_grad():
            m_g.exec_get_outputs(rb)
    assert len(kept) == 2 and kept[0].data_ptr() == captured_ptr and kept[1].shape[0] >= 3000
    junk = [torch.full((1 << 18,), float("nan"), device="cuda") for _ in range(16)] + \
           [torch.full((k,), 0x7FFFFFF0, dtype=torch.int32, device="cuda") for k in (64, 300, 512, 1024, 4096, 1 << 14) for _ in range(8)]
    del junk
    for it in range(2):
        bundles, batch, jit = batch_of(70 + it)
        l_g = {k: float(v) for k, v in step(*bundles, batch, jitter=jit).items()}
        l_e, _, _ = _eager_step(m_e, o_e, bundles, batch, jit, False)
        for k in l_e:
            assert abs(l_g[k] - l_e[k]) <= 2e-5 * max(1.0, abs(l_e[k])), (it, k, l_g[k], l_e[k])
    assert bool(torch.isfinite(o_g.flat.data).all()) and int(kept[0].abs().max()) == 0
    step.check_overflow()
    step.close()


def test_graphed_step_feeds_a_pose_optimiser_outside_the_graph():
    """BASELINE config 4 with the captured step: the spline camera optimiser turns pixel samples into 4 virtual-camera rays per
    pixel OUTSIDE the graph (eager torch code, R:lse_nerf/ns_camera_optimizer.py), the graph renders them, averages the 4 renders
    per pixel (deblur), back-propagates to the rays, and ``step.ray_grads`` continues the backward pass eagerly into the pose
    parameters.  The control-tangent gradients must equal those of the fully eager step."""
    import numpy as np
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle, cameras as cam
    from lsenerf_amd.graph import GraphedTrainStep
    from lsenerf_amd.optim import FlatAdam, FlatParams
    from tests.test_cameras_cpu import gen_data
    torch.manual_seed(96)
    cfg = LSENeRFModelConfig(grid_levels=2, grid_resolution=32, log2_hashmap_size=15, rgb_loss_type="deblur", use_mapping=False)
    base = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 6)
    with torch.no_grad():
        base.field.mlp_base_grid.params.mul_(300.0)
    c2w, ts = gen_data(6, max_t=1.0, seed=2)
    c2w[:, :3, 3] *= 0.3
    g = torch.Generator().manual_seed(1)
    n_px = 48
    ci = torch.randint(1, 5, (n_px,), generator=g).cuda()
    coords = torch.randint(0, 32, (n_px, 2), generator=g).float().cuda()
    gt = torch.rand(n_px, 3, generator=g).cuda()
    jit = torch.rand(n_px * 4, generator=g).cuda()
    grads = []
    for graphed in (False, True):
        m = copy.deepcopy(base).cuda().train()
        m.occupancy_grid.binaries.copy_(random_binaries(2, 32, 0.5, 3).cuda())
        m.occupancy_grid.occs.copy_(m.occupancy_grid.binaries.flatten().float() * 0.5)
        opt = FlatAdam(FlatParams(m.get_param_groups()["fields"]), lr=1e-2, eps=1e-15)
        cams = cam.EdCameras(torch.from_numpy(c2w), 60.0, 60.0, 16.0, 16.0, 32, 32, times=torch.from_numpy(ts))
        spl = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="spline", exp_t=0.05).setup(num_cameras=6, device="cpu", cameras=cams,
                                                                                              dM=torch.eye(4)).to("cuda")
        spl.device = "cuda"
        cams.times = cams.times.cuda()
        rb = cam.generate_deblur_rays(cams, spl, ci, coords)                      # differentiable w.r.t. the spline's control tangents
        rb.metadata.setdefault("appearance_id", torch.zeros(len(rb), dtype=torch.long, device="cuda"))
        batch = {"col_batch": {"image": gt}, "evs_batch": None}
        if graphed:
            step = GraphedTrainStep(m, opt, rb, None, None, batch, ray_grads=True, jitter="input")
            loss = step(rb, None, None, batch, jitter=jit)["rgb_loss"]
            g_o, g_d = step.ray_grads["col"]
            torch.autograd.backward([rb.origins, rb.directions], [g_o, g_d])         # ... and on into the pose parameters, eagerly
            table_grad = opt.flat.grad.clone()
        else:
            opt.zero_grad()
            _, losses, _ = m.train_step_bundles(rb, None, None, batch, jitter=jit)
            loss = losses["rgb_loss"]
            loss.backward()
            table_grad = opt.flat.grad.clone()
        pose = [p.grad.clone() for p in spl.parameters() if p.grad is not None]
        assert pose and all(float(p.abs().max()) > 0 for p in pose)
        grads.append((float(loss), pose, table_grad))
    (l0, p0, t0), (l1, p1, t1) = grads
    assert abs(l0 - l1) < 2e-6 * max(1.0, abs(l0))
    assert nmax_err(t1, t0, 1e-12) < 3e-5
    for a, b in zip(p1, p0):
        assert nmax_err(a, b, 1e-12) < 1e-4


def test_graphed_step_with_the_next_steps_marcher_on_a_side_stream_equals_the_eager_step():
    """prefetch_march=True: each replay marches the NEXT step's rays on a side stream (two graphs alternating between two sample
    buffers).  The samples must be those the eager step draws: checked step by step through the losses, across an occupancy
    refresh (the samples marched ahead are stale then and the rays are marched again) and across a call that announces no rays."""
    from lsenerf_amd import ops
    from lsenerf_amd.graph import GraphedTrainStep
    (m_e, m_g), (o_e, o_g), batch_of = _setup(False)
    steps = [batch_of(60 + 10 * it) for it in range(7)]
    b0, batch0, jit0 = steps[0]
    step = GraphedTrainStep(m_g, o_g, *b0, batch0, jitter="input", prefetch_march=True)
    assert o_g.step_count == 0 and torch.equal(o_g.flat.data, o_e.flat.data)
    ops.SYNC_STATS.update(seconds=0.0, count=0)
    marched_again = []
    for it in range(6):
        if it == 3:                      # refresh between replays: grid_version moves, the samples marched during replay 2 are stale
            for m in (m_e, m_g):
                m.update_occupancy_grid(0)
        bundles, batch, jit = steps[it]
        nb, _, njit = steps[it + 1]
        announce = it != 4               # step 4 announces nothing: step 5 has to march its own rays
        before = (step._pm_version, m_g.occupancy_grid.grid_version)
        marched_again.append(before[0] is None or before[0] != before[1])
        sync_before = ops.SYNC_STATS["count"]
        l_g = step(*bundles, batch, jitter=jit, next_bundles=nb if announce else None, next_jitter=njit if announce else None)
        assert ops.SYNC_STATS["count"] == sync_before
        l_g = {k: float(v) for k, v in l_g.items()}
        l_e, _, _ = _eager_step(m_e, o_e, bundles, batch, jit, False)
        for k in l_e:
            assert abs(l_g[k] - l_e[k]) <= 2e-5 * max(1.0, abs(l_e[k])), (it, k, l_g[k], l_e[k])
        assert o_g.step_count == o_e.step_count == it + 1
        if it == 3:      # after four Adam steps: the bound of test_graphed_step_equals_eager_step (noise-dominated elements only)
            d = (o_g.flat.data - o_e.flat.data).abs()
            assert float((d > 1e-5 * float(o_e.flat.data.abs().max())).float().mean()) < 0.02
    assert marched_again == [True, False, False, True, False, True]        # first call, refresh, nothing announced
    assert step.remarched_unannounced == 0
    # rays that were NOT the announced ones: the call for steps[6] (announced by the last loop iteration) announces steps[0], and the
    # call after it brings steps[5] instead -- the samples marched ahead belong to other rays and must be dropped (the step must
    # equal the eager step on steps[5], not train on steps[0]'s samples)
    bundles, batch, jit = steps[6]
    step(*bundles, batch, jitter=jit, next_bundles=steps[0][0], next_jitter=steps[0][2])
    _eager_step(m_e, o_e, bundles, batch, jit, False)
    assert step.remarched_unannounced == 0
    bundles, batch, jit = steps[5]
    l_g = {k: float(v) for k, v in step(*bundles, batch, jitter=jit).items()}
    l_e, _, _ = _eager_step(m_e, o_e, bundles, batch, jit, False)
    assert step.remarched_unannounced == 1
    for k in l_e:
        assert abs(l_g[k] - l_e[k]) <= 2e-5 * max(1.0, abs(l_e[k])), (k, l_g[k], l_e[k])
    # The identity of the announced rays is the tensor OBJECTS + their version counters, and the step HOLDS them until the next call:
    # a caller that drops the announced bundle cannot get its storage back from the allocator for other rays at the same address and
    # version 0 (which an address-based signature would take for the announced ones and train on the wrong samples).
    import gc, weakref
    # (nine optimizer steps in, the two models have drifted apart in their noise-dominated elements -- Adam with eps 1e-15, see
    #  test_graphed_step_equals_eager_step: re-align the eager model so that the comparison below is about ONE step again)
    with torch.no_grad():
        o_e.flat.data.copy_(o_g.flat.data); o_e.exp_avg.copy_(o_g.exp_avg); o_e.exp_avg_sq.copy_(o_g.exp_avg_sq)
    o_e.step_count = o_g.step_count
    bundles, batch, jit = steps[2]
    ann_bundles, _, ann_jit = batch_of(990)
    alive = weakref.ref(ann_bundles[0].origins)
    step(*bundles, batch, jitter=jit, next_bundles=ann_bundles, next_jitter=ann_jit)
    _eager_step(m_e, o_e, bundles, batch, jit, False)
    del ann_bundles, ann_jit
    gc.collect()
    assert alive() is not None, "the announced tensors must stay alive until the next call has compared them"
    other, batch, jit = batch_of(991)          # different rays, brand-new objects
    before = step.remarched_unannounced
    l_g = {k: float(v) for k, v in step(*other, batch, jitter=jit).items()}
    l_e, _, _ = _eager_step(m_e, o_e, other, batch, jit, False)
    assert step.remarched_unannounced == before + 1
    for k in l_e:
        assert abs(l_g[k] - l_e[k]) <= 2e-5 * max(1.0, abs(l_e[k])), (k, l_g[k], l_e[k])
    step.check_overflow()
    step.close()


@pytest.mark.parametrize("prefetch", [False, True])
def test_graph_without_the_optimizer_leaves_exchange_and_update_to_the_caller(prefetch):
    """optimizer_in_graph=False (data parallel: the all-reduce of the flat gradient sits between the backward pass and Adam): the
    replay leaves this rank's gradient in opt.flat.grad and updates nothing; the caller's exchange + opt.step() finish the step.
    Same losses, gradients and step counts as the eager step."""
    from lsenerf_amd.graph import GraphedTrainStep
    (m_e, m_g), (o_e, o_g), batch_of = _setup(False)
    b0, batch0, jit0 = batch_of(50)
    step = GraphedTrainStep(m_g, o_g, *b0, batch0, jitter="input", optimizer_in_graph=False, prefetch_march=prefetch)
    spans = [(o, o + p.numel()) for p, o in zip(o_e.flat.params, o_e.flat.offsets)]
    steps = [batch_of(60 + 10 * it) for it in range(4)]
    for it in range(3):
        bundles, batch, jit = steps[it]
        before = o_g.flat.data.clone()
        kw = {"next_bundles": steps[it + 1][0], "next_jitter": steps[it + 1][2]} if prefetch else {}
        l_g = {k: float(v) for k, v in step(*bundles, batch, jitter=jit, **kw).items()}
        assert torch.equal(o_g.flat.data, before) and o_g.step_count == it          # the replay updated nothing
        g_g = o_g.flat.grad.clone()
        # (data parallel: lsenerf_amd.dist.allreduce_grads(o_g.flat.grad) here, then the update with grad_scale = 1 / world)
        o_g.step(grad_scale=1.0)
        l_e, g_e, _ = _eager_step(m_e, o_e, bundles, batch, jit, False)
        for k in l_e:
            assert abs(l_g[k] - l_e[k]) <= 2e-5 * max(1.0, abs(l_e[k])), (it, k, l_g[k], l_e[k])
        if it == 0:
            for a, b in spans:
                assert nmax_err(g_g[a:b], g_e[a:b], 1e-12) < 3e-5, (a, b)
        assert o_g.step_count == o_e.step_count == it + 1
    step.check_overflow()
    step.close()


def test_a_violated_capacity_inside_a_replayed_graph_is_reported_at_the_next_refresh():
    """The captured marcher ORs into the estimator's sticky overflow accumulator at every replay.  Capture with rays that yield
    nothing (so that construction sees no overflow), replay with rays that exceed a capacity forced to 24 slots: the replays run
    to completion on truncated rays and the next occupancy refresh raises."""
    from lsenerf_amd import RayBundle
    from lsenerf_amd.graph import GraphedTrainStep
    (_, m), (_, opt), batch_of = _setup(False)
    m.occupancy_grid._cap_per_ray = lambda *a, **k: 24
    (col, prev, nxt), batch, jit = batch_of(50)
    away = lambda b: RayBundle(origins=b.origins + 50.0, directions=b.directions, camera_indices=b.camera_indices, metadata=b.metadata)
    step = GraphedTrainStep(m, opt, away(col), away(prev), away(nxt), batch, jitter="input")
    step(away(col), away(prev), away(nxt), batch, jitter=jit)
    m.update_occupancy_grid(16)                      # nothing overflowed so far
    m.occupancy_grid.binaries.fill_(True)
    m.occupancy_grid._bump_grid_version()
    for _ in range(3):
        losses = step(col, prev, nxt, batch, jitter=jit)
    assert all(bool(torch.isfinite(v)) for v in losses.values())
    assert int(step.outputs["col_out"]["num_samples_per_ray"].max()) <= 24
    with pytest.raises(RuntimeError, match="more samples than"):
        m.update_occupancy_grid(32)
    step.check_overflow()                            # reported once
    step.close()


def test_replays_issued_far_ahead_of_the_device_take_their_own_steps_scalars():
    """The optimizer clock of a captured step is on the device (lse_adam_schedule_dev): 64 replays are issued back to back while
    the device is still busy with a long kernel queued in front of them -- the host is 64 steps ahead -- and every replay must
    see ITS step's learning rate and bias corrections.  (A per-step host -> device copy out of one pinned staging buffer, the
    round-3 scheme, hands early replays the scalars of later steps in exactly this situation.)"""
    import math
    from lsenerf_amd.optim import FlatAdam, FlatParams
    p = torch.nn.Parameter(torch.randn(1 << 16, device="cuda"))
    opt = FlatAdam(FlatParams([p]), lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=40)
    opt.step_count = 5                               # e.g. resumed from a checkpoint
    opt.flat.grad.normal_()
    n = 64
    log = torch.zeros(n + 8, 6, device="cuda")      # hyper = (lr, 1 - b1^t, 1 / sqrt(1 - b2^t), b1, b2, eps): all six live on the device
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.prepare_step(); opt.step_staged()
    torch.cuda.current_stream().wait_stream(side)
    opt.step_count = 5
    graph = torch.cuda.CUDAGraph()
    opt.prepare_step()
    with torch.cuda.graph(graph):
        opt.step_staged()
        log.index_copy_(0, opt._step_dev - 6, opt._hyper_dev[None])
    opt.step_count = 5
    big = torch.randn(8192, 8192, device="cuda")
    for _ in range(6):                               # ~ tens of milliseconds of device work queued in front of the replays
        big = big @ big * 1e-4
    for _ in range(n):
        opt.prepare_step()
        graph.replay()
    torch.cuda.synchronize()
    assert opt.step_count == 5 + n and int(opt._step_dev) == 5 + n
    got = log[:n].double().cpu()
    for i in range(n):
        t = 6 + i                                    # the step this replay takes; its lr is that of `t - 1` finished steps
        f = min((t - 1) / 40, 1.0)
        lr = math.exp(math.log(1e-2) * (1 - f) + math.log(1e-4) * f)
        want = (lr, 1 - 0.9 ** t, 1 / math.sqrt(1 - 0.999 ** t), 0.9, 0.999, 1e-15)
        for a, b in zip(got[i].tolist(), want):
            assert abs(a - b) <= 2e-7 * abs(b), (i, got[i].tolist(), want)
    # The schedule's constants are device-side state too (ABI 5: they were launch arguments, frozen into the captured graph):
    # a param group loaded from a checkpoint, or a manual learning-rate drop, reaches the replays of the SAME graph.
    opt.lr_init, opt.lr_final, opt.max_steps, opt.betas = 5e-3, 5e-5, 100, (0.8, 0.99)
    opt.prepare_step()
    graph.replay()
    torch.cuda.synchronize()
    t = 5 + n + 1
    f = (t - 1) / 100
    want = (math.exp(math.log(5e-3) * (1 - f) + math.log(5e-5) * f), 1 - 0.8 ** t, 1 / math.sqrt(1 - 0.99 ** t), 0.8, 0.99, 1e-15)
    got = opt._hyper_dev.double().cpu().tolist()
    for a, b in zip(got, want):
        assert abs(a - b) <= 2e-7 * abs(b), (got, want)
    assert abs(opt.current_lr() - math.exp(math.log(5e-3) * (1 - t / 100) + math.log(5e-5) * t / 100)) < 1e-12     # host mirror agrees


def test_capturing_a_step_survives_earlier_eager_graphs_of_the_same_model(capture_probe):
    """Eager losses that are still alive keep their autograd graph and the parameters' AccumulateGrad nodes, bound to the stream they
    were created on.  Until round 4 a capture whose backward handed a gradient to such a node (the loss epilogue's scalar parameters)
    crashed the HIP runtime in hipStreamEndCapture.  No parameter gradient goes through AccumulateGrad inside a step any more -- fast
    path and torch route of the loss (ops._direct_grad, ops.direct_grad_params) -- and the rays are fresh leaves per run.  The probe
    runs as a child process started by tests/conftest.py before this process touched the GPU (a regression may be a segfault), with
    torch's stream-mismatch warning as an error: two eager steps with retained graphs, then capture + two replays with ray gradients,
    for the cfg 2 and cfg 3 compositions at full size and for cfg 2 on the torch route of the loss."""
    if not capture_probe.get("launched"):
        pytest.skip(capture_probe.get("reason", "probe not launched"))
    assert capture_probe["returncode"] == 0, capture_probe["log_tail"]
    for kind in ("cfg2", "cfg3", "cfg2_torch_route"):
        assert f"{kind} captured and replayed with 2 eager graphs alive" in capture_probe["log_tail"], capture_probe["log_tail"]
