"""Training fast path without host read-backs of the sample counts (LSENeRFModel.deferred_counts -> LSEOccGridEstimator.sampling
(deferred=True) -> the n_dev argument of the per-sample entry points): the packed arrays have capacity extent, the counts stay on the device, every
per-sample kernel clamps to them.  Values must be those of the synchronising path: renders and sample counts bit for bit (same
kernels on the same samples), gradients up to the order of float-atomic summation."""
import pytest
import torch

from tests.util import TOL_GRAD, nmax_err, random_binaries, random_rays

pytestmark = pytest.mark.gpu


def _model(**cfg_kw):
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig
    torch.manual_seed(96)
    cfg = LSENeRFModelConfig(grid_levels=2, grid_resolution=32, log2_hashmap_size=15, **cfg_kw)
    m = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 16)
    with torch.no_grad():
        m.field.mlp_base_grid.params.mul_(300.0)
    m = m.cuda().train()
    m.occupancy_grid.binaries.copy_(random_binaries(2, 32, 0.5, 3).cuda())
    m.occupancy_grid.occs.copy_(m.occupancy_grid.binaries.flatten().float() * 0.5)
    return m


def _step(m, rb, jit, target):
    for p in m.parameters():
        p.grad = None
    rb.origins.grad = rb.directions.grad = None
    out = m.exec_get_outputs(rb, jitter=jit)
    loss = torch.nn.functional.mse_loss(out["rgb"], target) + 0.1 * out["depth"].mean() + 0.05 * out["accumulation"].mean()
    loss.backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    grads["origins"], grads["directions"] = rb.origins.grad.clone(), rb.directions.grad.clone()
    return {k: v.detach().clone() for k, v in out.items()}, grads, float(loss.detach())


@pytest.mark.parametrize("case", ["default_prepass_cone", "no_culling_constant_step"])
def test_deferred_counts_equal_the_synchronising_path(case):
    from lsenerf_amd import RayBundle, ops
    kw = {} if case == "default_prepass_cone" else dict(cone_angle=0.0, alpha_thre=0.0)
    m = _model(**kw)
    R = 300
    o, d = random_rays(R, seed=4)
    g = torch.Generator().manual_seed(1)
    rb = RayBundle(origins=o.cuda().requires_grad_(True), directions=d.cuda().requires_grad_(True),
                   camera_indices=torch.zeros(R, 1, dtype=torch.long, device="cuda"),
                   metadata={"appearance_id": torch.randint(0, 16, (R,), generator=g).cuda()})
    jit, target = torch.rand(R, generator=g).cuda(), torch.rand(R, 3, generator=g).cuda()
    ops.SYNC_STATS.update(seconds=0.0, count=0)
    assert m.deferred_counts is True            # the model's default
    m.deferred_counts = False
    out_s, g_s, l_s = _step(m, rb, jit, target)
    n_sync = ops.SYNC_STATS["count"]
    assert n_sync == 2          # marcher count + survivors of the cull (early_stop_eps > 0 keeps the pre-pass on in both cases)
    m.deferred_counts = True
    out_d, g_d, l_d = _step(m, rb, jit, target)
    assert ops.SYNC_STATS["count"] == n_sync, "the deferred path read a sample count back to the host"
    m.occupancy_grid.check_deferred_overflow()
    assert int(out_s["num_samples_per_ray"].sum()) > 20 * R
    for k in ("num_samples_per_ray", "rgb", "accumulation", "depth"):
        assert torch.equal(out_s[k], out_d[k]), k
    assert l_s == l_d
    assert set(g_s) == set(g_d)
    for k in g_s:
        assert nmax_err(g_d[k], g_s[k], 1e-12) < 3e-5, k               # float atomics: two runs of one path differ as much
    # a second deferred step on other rays (capacity buffers are recycled by the allocator: stale tails must not matter)
    o2, d2 = random_rays(R, seed=5)
    rb2 = RayBundle(origins=o2.cuda().requires_grad_(True), directions=d2.cuda().requires_grad_(True),
                    camera_indices=rb.camera_indices, metadata=rb.metadata)
    out_d2, g_d2, _ = _step(m, rb2, jit, target)
    m.deferred_counts = False
    out_s2, g_s2, _ = _step(m, rb2, jit, target)
    for k in ("num_samples_per_ray", "rgb", "accumulation", "depth"):
        assert torch.equal(out_s2[k], out_d2[k]), k
    for k in g_s2:
        assert nmax_err(g_d2[k], g_s2[k], 1e-12) < 3e-5, k


def test_deferred_counts_with_no_sample_at_all_and_capacity_bound():
    """Every ray misses every occupied cell: the device-side count is 0, nerfstudio's single fake sample is inserted on the device,
    the renders equal the synchronising path's and the gradients vanish -- nothing reads the (uninitialised) capacity buffers.  And the capacity is what
    _cap_per_ray promises: never smaller than the count of the synchronising path, for cone and constant steps."""
    from lsenerf_amd import RayBundle
    m = _model()
    m.deferred_counts = True
    m.occupancy_grid.binaries.zero_()
    R = 64
    # what torch.empty hands out next is what the caching allocator holds: NaN.  (Round 4: with no survivor the pre-pass features
    # parked for the main pass held nothing for slot 0, where the fake sample lands -- NaN renders, depending on the heap's history.)
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(24)] + \
           [torch.full((n,), float("nan"), device="cuda") for n in (64, 192, 1024, 4096, 1 << 14, 1 << 16, 1 << 18) for _ in range(4)]
    del junk
    o, d = random_rays(R, seed=2)
    rb = RayBundle(origins=o.cuda().requires_grad_(True), directions=d.cuda().requires_grad_(True),
                   camera_indices=torch.zeros(R, 1, dtype=torch.long, device="cuda"),
                   metadata={"appearance_id": torch.zeros(R, dtype=torch.long, device="cuda")})
    out = m.exec_get_outputs(rb)
    # (nerfstudio's single fake sample -- ray 0, t_start = t_end = 1, weight 0 -- is inserted on the device: lse_fake_sample_if_empty)
    assert out["num_samples_per_ray"].tolist() == [1] + [0] * (R - 1) and float(out["rgb"].detach().abs().max()) == 0.0
    m.deferred_counts = False
    out_s = m.exec_get_outputs(rb)
    m.deferred_counts = True
    for k in ("num_samples_per_ray", "rgb", "accumulation", "depth"):
        assert torch.equal(out[k], out_s[k]), k
    out["rgb"].sum().backward()
    assert float(m.field.mlp_base_grid.params.grad.abs().max()) == 0.0
    # capacity bound against fully occupied grids (the worst case for the count)
    m.occupancy_grid.binaries.fill_(True)
    est = m.occupancy_grid
    for cone in (0.0, 0.004, 0.02):
        cap = est._cap_per_ray(0.05, 1e3, m.config.render_step_size, cone)
        _, ts, _, packed = est.sampling(rb.origins.detach(), rb.directions.detach(), near_plane=0.05, far_plane=1e3,
                                        render_step_size=m.config.render_step_size, cone_angle=cone, return_packed=True)
        assert int(packed[:, 1].max()) <= cap and int(packed[:, 1].max()) > 100, (cone, cap, int(packed[:, 1].max()))
    est.check_deferred_overflow()


def test_train_step_bundles_deferred_equals_synced():
    from lsenerf_amd import RayBundle
    m = _model(use_mapping=True, mapping_method="identity", map_mode="co_map", evs_mapping_method="powpow")
    g = torch.Generator().manual_seed(3)
    sizes = (200, 50, 50)
    bundles, jit = [], []
    for i, n in enumerate(sizes):
        o, d = random_rays(n, seed=40 + i)
        bundles.append(RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(n, 1, dtype=torch.long, device="cuda"),
                                 metadata={"appearance_id": torch.randint(0, 16, (n,), generator=g).cuda()}))
        jit.append(torch.rand(n, generator=g).cuda())
    batch = {"col_batch": {"image": torch.rand(sizes[0], 3, generator=g).cuda()},
             "evs_batch": {"image": ((torch.rand(sizes[1], 1, generator=g) - 0.5) * 0.4).cuda()}}
    res = []
    for deferred in (False, True):
        m.deferred_counts = deferred
        for p in m.parameters():
            p.grad = None
        out, losses, _ = m.train_step_bundles(*bundles, batch, jitter=torch.cat(jit))
        sum(losses.values()).backward()
        res.append((out, {k: float(v) for k, v in losses.items()}, m.field.mlp_base_grid.params.grad.clone()))
    (o0, l0, g0), (o1, l1, g1) = res
    assert l0 == l1
    for k in ("col_out", "prev_out", "next_out"):
        assert torch.equal(o0[k]["rgb"], o1[k]["rgb"]) and torch.equal(o0[k]["num_samples_per_ray"], o1[k]["num_samples_per_ray"])
    assert nmax_err(g1, g0, 1e-12) < 3e-5


def test_samples_marched_ahead_of_the_step_are_the_steps_own_samples():
    """LSEOccGridEstimator.march_deferred / LSENeRFModel.premarch_bundles: the marcher launched ahead of its step (it reads rays and
    grid only) and handed back through ``premarched=`` must give the render of the step that marches for itself, bit for bit;
    reused buffers (``out=``) are overwritten in place and refused when they do not fit; ``grid_version`` moves with every change
    of the grid made through the estimator."""
    from lsenerf_amd import RayBundle, _lib, ops
    m = _model()
    est = m.occupancy_grid
    R = 200
    g = torch.Generator().manual_seed(2)

    def bundle(seed, n):
        o, d = random_rays(n, seed=seed)
        return RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(n, 1, dtype=torch.long, device="cuda"),
                         metadata={"appearance_id": torch.randint(0, 16, (n,), generator=g).cuda()})
    col, prev, nxt = bundle(4, R), bundle(5, 40), bundle(6, 40)
    jit = torch.rand(R + 80, generator=g).cuda()
    batch = {"col_batch": {"image": torch.rand(R, 3, generator=g).cuda()}, "evs_batch": {"image": torch.rand(40, 1, generator=g).cuda() - 0.5}}
    out_ref, loss_ref, _ = m.train_step_bundles(col, prev, nxt, batch, jitter=jit)
    pm = m.premarch_bundles(col, prev, nxt, jitter=jit)
    assert pm.n_rays == R + 80 and pm.grid_version == est.grid_version and int(pm.n_dev) > 20 * R
    out_pm, loss_pm, _ = m.train_step_bundles(col, prev, nxt, batch, premarched=pm)            # (jitter was applied by the marcher)
    for k in ("col_out", "prev_out", "next_out"):
        for key in ("rgb", "depth", "accumulation", "num_samples_per_ray"):
            assert torch.equal(out_pm[k][key], out_ref[k][key]), (k, key)
    assert all(torch.equal(loss_pm[k], loss_ref[k]) for k in loss_ref)
    # the same buffers filled again for other rays: in place, same tensors
    col2 = bundle(14, R)
    ptrs = [t.data_ptr() for t in pm[:6]]
    pm2 = m.premarch_bundles(col2, prev, nxt, jitter=jit, out=pm)
    assert [t.data_ptr() for t in pm2[:6]] == ptrs
    ref2, _, _ = m.train_step_bundles(col2, prev, nxt, batch, jitter=jit)
    got2, _, _ = m.train_step_bundles(col2, prev, nxt, batch, premarched=pm2)
    assert torch.equal(got2["col_out"]["rgb"], ref2["col_out"]["rgb"])
    with pytest.raises(_lib.LseHipError, match="not a result"):
        m.premarch_bundles(col, None, None, jitter=jit[:R], out=pm)                            # other ray count: other capacity
    with pytest.raises(ValueError, match="premarched samples are for"):
        m.train_step_bundles(col, None, None, {"col_batch": batch["col_batch"], "evs_batch": None}, premarched=pm2)
    # grid_version follows the grid
    v0 = est.grid_version
    m.update_occupancy_grid(0)
    assert est.grid_version == v0 + 1
    est.mark_all_occupied()
    assert est.grid_version == v0 + 2
    m.update_occupancy_grid(5)            # not a refresh step: nothing changes
    assert est.grid_version == v0 + 2
    est.check_deferred_overflow()


def test_a_violated_capacity_truncates_rays_in_bounds_and_is_reported_at_the_next_refresh():
    """The count-free sampler's capacity is a proven bound for unit-length directions.  When it is violated anyway (here: forced
    to 24 slots per ray; and once through directions of length 0.25) the marcher truncates the ray to its capacity -- the count it
    hands on never exceeds the slots, nothing is read from a neighbour's slots nor written past the packed arrays -- and ORs into
    the estimator's sticky accumulator, which the product path reads at the next occupancy refresh
    (LSENeRFModel.update_occupancy_grid) and which is never dropped unread, however many calls lie in between."""
    from lsenerf_amd import RayBundle
    m = _model(cone_angle=0.0, alpha_thre=0.0)
    m.occupancy_grid.binaries.fill_(True)
    est = m.occupancy_grid
    R = 96
    o, d = random_rays(R, seed=7)
    o, d = o.cuda(), d.cuda()
    step = m.config.render_step_size
    ref = est.sampling(o, d, near_plane=0.05, far_plane=1e3, render_step_size=step, return_packed=True)
    full_cnt = ref[3][:, 1]
    assert int(full_cnt.max()) > 100
    cap = 24
    est._cap_per_ray = lambda *a, **k: cap
    ri, ts, te, packed, n_dev = est.sampling(o, d, near_plane=0.05, far_plane=1e3, render_step_size=step, deferred=True)
    assert ts.shape[0] == R * cap
    cnt = packed[:, 1]
    assert torch.equal(cnt, full_cnt.clamp(max=cap))                     # truncated to the capacity, ray by ray
    n = int(n_dev)
    assert n == int(cnt.sum()) <= R * cap
    assert torch.equal(packed[:, 0], torch.cumsum(cnt, 0) - cnt)
    # the kept samples are the FIRST `cap` samples of each ray of the synchronising path, bit for bit
    keep = torch.cat([torch.arange(int(s), int(s) + int(c), device="cuda") for s, c in zip(ref[3][:, 0], cnt)])
    assert torch.equal(ts[:n], ref[1][keep]) and torch.equal(te[:n], ref[2][keep]) and torch.equal(ri[:n], ref[0][keep])
    # many more calls: the flag must survive them (until round 3 a Python list dropped its oldest 2048 entries unread)
    del est._cap_per_ray
    for _ in range(40):
        est.sampling(o, d, near_plane=0.05, far_plane=1e3, render_step_size=step, deferred=True)
    m.update_occupancy_grid(3)                                           # not a refresh step: nothing is read
    with pytest.raises(RuntimeError, match="more samples than"):
        m.update_occupancy_grid(16)
    m.update_occupancy_grid(32)                                          # reported once, then clear
    # directions shorter than 1 stretch the t-range inside the box beyond the bound: truncated, flagged, and named in the message
    est.sampling(o, d * 0.25, near_plane=0.05, far_plane=1e3, render_step_size=step, deferred=True)
    with pytest.raises(RuntimeError, match="shorter than 1"):
        est.check_deferred_overflow()
    est.check_deferred_overflow()
    # ... and the whole eager training step on truncated rays runs to completion (no fault) before the refresh reports it
    est._cap_per_ray = lambda *a, **k: cap
    rb = RayBundle(origins=o.clone().requires_grad_(True), directions=d.clone().requires_grad_(True),
                   camera_indices=torch.zeros(R, 1, dtype=torch.long, device="cuda"),
                   metadata={"appearance_id": torch.zeros(R, dtype=torch.long, device="cuda")})
    out = m.exec_get_outputs(rb)
    out["rgb"].sum().backward()
    assert int(out["num_samples_per_ray"].max()) <= cap and bool(torch.isfinite(out["rgb"]).all())
    with pytest.raises(RuntimeError, match="more samples than"):
        m.update_occupancy_grid(48)
