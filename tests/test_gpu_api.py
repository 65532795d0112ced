"""GPU tests of the reference-interface surface beyond the training fast path: eval mode, empty rays, the generic
RaySamples API of LSEField (what a nerfstudio caller would use), renderer modules, occupancy refresh callback."""
import pytest
import torch

from tests.util import TOL_FWD, TOL_GRAD, make_model_pair, nmax_err, random_rays

pytestmark = pytest.mark.gpu


def _bundle(o, d, **kw):
    from lsenerf_amd import RayBundle
    return RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(o.shape[0], 1, dtype=torch.long).cuda(), **kw)


@pytest.mark.parametrize("linear", [False, True])
def test_eval_mode_forward_matches_oracle(linear):
    """not training: no sigma_fn pre-pass, no stratified jitter (R: VolumetricSampler), zero-embedding eval mode
    (R:lse_nerf/lse_embeddings.py:51-55), eval clamp of RGBRenderer unless LinearRenderer (R:lse_nerf/lse_renderer.py)."""
    hip, orc = make_model_pair(grid_levels=2, grid_resolution=32, occupied_frac=0.5, param_scale=300.0, emb_type="evs_emb")
    if linear:
        from lsenerf_amd import LinearRenderer
        hip.renderer_rgb = LinearRenderer(background_color="random")
        orc.linear_renderer = True
    with torch.no_grad():   # push colours above 1 so the clamp matters
        hip.field.mlp_head.params[-16 * 64:].mul_(0.0)
        hip.field.mlp_base_mlp.params[-16 * 64:-15 * 64].add_(0.3)
    from tests.util import sync_params_to_oracle
    sync_params_to_oracle(hip, orc.field)
    hip.eval(); orc.training = False; orc.field.training = False
    o, d = random_rays(80, seed=5)
    with torch.no_grad():
        out = hip.exec_get_outputs(_bundle(o, d, metadata={"appearance_id": torch.zeros(80, dtype=torch.long).cuda()}))
        ref = orc.exec_get_outputs(o, d, torch.zeros(80, dtype=torch.long))
    assert torch.equal(out["num_samples_per_ray"].cpu(), ref["num_samples_per_ray"])       # sampler: bit-exact counts
    for k in ("rgb", "accumulation", "depth"):
        assert nmax_err(out[k], ref[k], 1e-3) < 5 * TOL_FWD, k
    final = hip.route_outputs(out, None)
    assert float(final["rgb"].max()) <= 1.0 and float(final["rgb"].min()) >= 0.0              # R:lse_nerf/lsenerf.py:372-373


def test_rays_without_samples_take_the_fake_sample_path():
    """All rays miss every grid -> VolumetricSampler inserts one fake sample (ray 0, t = 1); outputs stay finite."""
    hip, orc = make_model_pair(grid_levels=1, grid_resolution=16, occupied_frac=0.5, param_scale=300.0)
    hip.train(); orc.training = True
    o = torch.tensor([[5.0, 5.0, 5.0], [6.0, 5.0, 5.0], [7.0, 5.0, 5.0]])
    d = torch.tensor([[1.0, 0, 0]] * 3)
    out = hip.exec_get_outputs(_bundle(o, d), jitter=torch.zeros(3).cuda())
    ref = orc.exec_get_outputs(o, d, None, jitter=torch.zeros(3))
    assert out["num_samples_per_ray"].tolist() == ref["num_samples_per_ray"].tolist() == [1, 0, 0]
    assert nmax_err(out["rgb"], ref["rgb"], 1e-3) < 5 * TOL_FWD and bool(torch.isfinite(out["depth"]).all())
    out["rgb"].sum().backward()                                                              # backward through an (almost) empty batch
    assert all(torch.isfinite(p.grad).all() for p in hip.field.parameters() if p.grad is not None)
    # an occupancy grid with nothing in it: same path
    hip.occupancy_grid.binaries.zero_(); orc.grid.binaries.zero_()
    o2, d2 = random_rays(16, seed=1)
    out2 = hip.exec_get_outputs(_bundle(o2, d2), jitter=torch.zeros(16).cuda())
    assert int(out2["num_samples_per_ray"].sum()) == 1


def test_generic_raysamples_api_of_lsefield():
    """LSEField.get_density / get_outputs / forward / density_fn with plain per-sample tensors of any batch shape (no
    packed bookkeeping) -- what a nerfstudio caller passes -- against the oracle field."""
    from lsenerf_amd import Frustums, RaySamples
    from lsenerf_amd.field import FieldHeadNames
    hip, orc = make_model_pair(grid_levels=1, grid_resolution=16, param_scale=300.0, emb_type="evs_emb")
    fld = hip.field
    fld.train(); orc.field.training = True
    g = torch.Generator().manual_seed(0)
    B, S = 6, 11
    o = (torch.rand(B, S, 3, generator=g) - 0.5) * 3
    d = torch.nn.functional.normalize(torch.randn(B, S, 3, generator=g), dim=-1)
    st = torch.rand(B, S, 1, generator=g)
    en = st + 0.01
    aid = torch.randint(0, 8, (B, S), generator=g)
    rs = RaySamples(Frustums(o.cuda(), d.cuda(), st.cuda(), en.cuda()), camera_indices=torch.zeros(B, S, 1, dtype=torch.long).cuda(),
                    metadata={"appearance_id": aid.cuda()})
    outs = fld(rs)
    dens, rgb = outs[FieldHeadNames.DENSITY], outs[FieldHeadNames.RGB]
    assert dens.shape == (B, S, 1) and rgb.shape == (B, S, 3)
    pos = (o + d * (st + en) / 2).reshape(-1, 3)
    dref, geo = orc.field.get_density(pos)
    rref = orc.field.get_outputs(d.reshape(-1, 3), geo, aid.reshape(-1))
    assert nmax_err(dens.reshape(-1), dref.reshape(-1)) < TOL_FWD and nmax_err(rgb.reshape(-1, 3), rref) < TOL_FWD
    dfn = fld.density_fn(pos.cuda().view(B, S, 3))
    assert dfn.shape == (B, S, 1) and nmax_err(dfn.reshape(-1), dref.reshape(-1)) < TOL_FWD
    (dens.sum() + rgb.sum()).backward()
    (dref.sum() + rref.sum()).backward()
    assert nmax_err(fld.mlp_base_grid.params.grad, orc.field.params["grid"].grad, 1e-12) < TOL_GRAD
    assert nmax_err(fld.embedding_appearance.embedding.weight.grad, orc.field.params["embedding"].grad, 1e-12) < TOL_GRAD
    # tcnn-style encoding output [N, L*F]
    enc = fld.mlp_base_grid(torch.rand(50, 3, generator=g).cuda())
    assert enc.shape == (50, 32)


def test_renderer_modules_match_oracle():
    from lsenerf_amd import AccumulationRenderer, DepthRenderer, Frustums, LinearRenderer, RaySamples, RGBRenderer
    from oracle import volrend as ovr
    g = torch.Generator().manual_seed(1)
    cnt = torch.tensor([3, 0, 70, 1])
    ri = torch.repeat_interleave(torch.arange(4), cnt)
    N = int(cnt.sum())
    w = torch.rand(N, 1, generator=g) * 0.05
    rgb = torch.rand(N, 3, generator=g) * 40           # linear radiance far above 1
    ts = torch.rand(N, 1, generator=g)
    rs = RaySamples(Frustums(torch.zeros(N, 3).cuda(), torch.zeros(N, 3).cuda(), ts.cuda(), (ts + 0.01).cuda()))
    for Renderer, training, bg in ((RGBRenderer, True, "random"), (RGBRenderer, False, "random"), (LinearRenderer, False, "random"),
                                   (RGBRenderer, True, "white")):
        r = Renderer(background_color=bg).cuda()
        r.train(training)
        got = r(rgb.cuda(), w.cuda(), ri.cuda(), 4)
        eff = True if Renderer is LinearRenderer else training
        assert nmax_err(got, ovr.render_rgb(rgb, w, ri, 4, eff, bg)) < TOL_FWD
    assert nmax_err(AccumulationRenderer()(w.cuda(), ri.cuda(), 4), ovr.render_accumulation(w, ri, 4)) < TOL_FWD
    assert nmax_err(DepthRenderer("expected")(w.cuda(), rs, ri.cuda(), 4),
                    ovr.render_depth_expected(w, ts[:, 0], ts[:, 0] + 0.01, ri, 4)) < TOL_FWD


def test_training_callback_refreshes_occupancy_grid():
    hip, _ = make_model_pair(grid_levels=2, grid_resolution=16, param_scale=300.0)
    hip.train()
    hip.occupancy_grid.binaries.zero_(); hip.occupancy_grid.occs.zero_()
    (cb,) = hip.get_training_callbacks()
    cb(0)                                                 # step 0 < warmup: every cell evaluated with field.density_fn * step
    occs = hip.occupancy_grid.occs
    assert float(occs.min()) >= 0 and float(occs.max()) > 0
    thre = min(float(occs.mean()), 0.01)
    assert torch.equal(hip.occupancy_grid.binaries.flatten(), occs > thre)
    before = occs.clone()
    cb(3)                                                 # not a multiple of 16: no update
    assert torch.equal(before, hip.occupancy_grid.occs)
    cb(320)                                               # sampled-cells branch
    assert bool(torch.isfinite(hip.occupancy_grid.occs).all())


def test_flat_params_receive_gradients_in_place_and_adam_matches_oracle():
    """optim.FlatParams keeps every .grad as a view of one flat buffer; the hash / MLP backward kernels then accumulate
    straight into it (ops.DIRECT_PARAM_GRADS).  Same gradients as the autograd-accumulated path, accumulation across two
    backward passes, and two FlatAdam steps track the oracle's Adam."""
    from lsenerf_amd import ops
    from lsenerf_amd.optim import FlatAdam, FlatParams
    hip, orc = make_model_pair(grid_levels=2, grid_resolution=32, occupied_frac=0.5, param_scale=300.0)
    hip.train()
    flat = FlatParams(hip.get_param_groups()["fields"])
    o, d = random_rays(200, seed=9)
    tgt = torch.rand(200, 3, generator=torch.Generator().manual_seed(3)).cuda()
    jit = torch.rand(200, generator=torch.Generator().manual_seed(4)).cuda()

    def backward_once():
        out = hip.exec_get_outputs(_bundle(o, d), jitter=jit)
        ((out["rgb"] - tgt) ** 2).mean().backward()

    grads = {}
    for direct in (True, False):
        ops.DIRECT_PARAM_GRADS = direct
        try:
            flat.zero_grad()
            backward_once()
            grads[direct] = flat.grad.clone()
        finally:
            ops.DIRECT_PARAM_GRADS = True
    assert float(grads[True].abs().max()) > 0
    assert nmax_err(grads[True], grads[False]) < TOL_GRAD
    backward_once()                                   # second pass without zeroing: accumulates
    assert nmax_err(flat.grad, 2 * grads[False]) < TOL_GRAD
    for p, off in zip(flat.params, flat.offsets):     # .grad are still views of the flat buffer
        assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * off
    # FlatAdam == torch.optim.Adam on the same gradients
    ref_p = flat.data.clone().requires_grad_(True)
    ref_opt = torch.optim.Adam([ref_p], lr=1e-2, eps=1e-15)
    opt = FlatAdam(flat, lr=1e-2, eps=1e-15)
    for _ in range(2):
        flat.zero_grad()
        backward_once()
        ref_p.grad = flat.grad.clone()
        opt.step()
        ref_opt.step()
        # keep the reference on the same trajectory (its parameters drive nothing; compare the update only)
        assert nmax_err(flat.data, ref_p.detach()) < 1e-5
        with torch.no_grad():
            ref_p.copy_(flat.data)


def test_main_pass_reuses_the_prepass_features_of_the_survivors():
    """Default configuration: the visibility pre-pass encodes every candidate; the survivors' unit-cube positions, selector and
    hash features are compacted (lse_compact_features) and picked up by the main pass instead of a second hash forward.
    Outputs and every gradient must be those of the path that encodes twice (bitwise for the forward: same kernels, same
    inputs), and the hand-over must refuse anything but exactly the samples / parameters it was made for."""
    from lsenerf_amd import _lib
    hip, _ = make_model_pair(grid_levels=2, grid_resolution=32, occupied_frac=0.5, param_scale=300.0, emb_type="evs_emb")
    hip.train()
    o, d = random_rays(200, seed=21)
    aid = torch.randint(0, 8, (200,), generator=torch.Generator().manual_seed(2)).cuda()
    jit = torch.rand(200, generator=torch.Generator().manual_seed(3)).cuda()
    res = {}
    calls = {}
    orig_call = _lib.call

    def counting_call(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return orig_call(name, *a)

    for reuse in (True, False):
        hip.field.reuse_prepass = reuse
        og, dg = o.clone().cuda().requires_grad_(True), d.clone().cuda().requires_grad_(True)
        rb = _bundle(og, dg, metadata={"appearance_id": aid})
        rb.origins, rb.directions = og, dg
        for p in hip.parameters():
            p.grad = None
        calls.clear()
        _lib.call = counting_call
        try:
            out = hip.exec_get_outputs(rb, jitter=jit)
        finally:
            _lib.call = orig_call
        n_hash_fwd = calls.get("lse_hash_fwd", 0)
        assert n_hash_fwd == (1 if reuse else 2), (reuse, calls)
        assert calls.get("lse_compact_features", 0) == (1 if reuse else 0)
        (out["rgb"].sum() + out["depth"].sum() + out["accumulation"].sum()).backward()
        res[reuse] = (out, {n: p.grad.clone() for n, p in hip.named_parameters() if p.grad is not None}, og.grad.clone(), dg.grad.clone())
        assert hip.field._prepass is None            # consumed
    a, b = res[True], res[False]
    assert int(a[0]["num_samples_per_ray"].sum()) > 1000
    for k in ("rgb", "depth", "accumulation", "num_samples_per_ray"):
        assert torch.equal(a[0][k], b[0][k]), k
    for n in a[1]:
        assert nmax_err(a[1][n], b[1][n], 1e-12) < TOL_GRAD, n           # float atomics: not bitwise
    assert nmax_err(a[2], b[2], 1e-12) < TOL_GRAD and nmax_err(a[3], b[3], 1e-12) < TOL_GRAD
    # the hand-over is keyed: other samples, or parameters that changed in between, do not take it
    hip.field.reuse_prepass = True
    rb = _bundle(o.clone().cuda(), d.clone().cuda(), metadata={"appearance_id": aid})
    rs, _ = hip.sampler(ray_bundle=rb, near_plane=0.05, far_plane=1e3, render_step_size=hip.config.render_step_size,
                        alpha_thre=0.01, cone_angle=0.004, jitter=jit)
    assert hip.field._prepass is not None
    with torch.no_grad():
        hip.field.mlp_base_grid.params.mul_(1.0)       # in-place update bumps the version: features are stale
    ts, te = rs.frustums.starts[..., 0], rs.frustums.ends[..., 0]
    assert hip.field._take_prepass(rb.origins, rb.directions, rs.ray_indices, ts) is None


def test_the_model_hands_its_sample_regime_to_the_hash_backward(monkeypatch):
    """The hash backward's two path thresholds (few_runs, stage_max) have different optima for a constant step and for steps that grow
    with the distance (profiles/r05_hash_bwd_thresholds.txt).  They are CALL ARGUMENTS (lse_hash_bwd_opts): the model tells the grid
    which regime its sampler produces (cone_angle), the grid's meta carries the pair to lse_hash_bwd_ex.  Values do not depend on
    it (every threshold pair is held against the oracle in test_hash_bwd_metric_regime_every_kernel_variant_vs_oracle)."""
    import ctypes
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle, _lib, ops
    seen = []
    real_call = _lib.call

    def spy(name, *args):
        if name == "lse_hash_bwd_ex":
            o = ctypes.cast(args[11], ctypes.POINTER(_lib.HashBwdOpts)).contents
            seen.append((o.few_runs, o.stage_max))
        return real_call(name, *args)
    monkeypatch.setattr(_lib, "call", spy)
    o, d = random_rays(64, seed=2)
    for cone, want_dense in ((0.0, True), (0.004, False)):
        torch.manual_seed(0)
        m = LSENeRFModel(LSENeRFModelConfig(cone_angle=cone, grid_levels=1, grid_resolution=32, log2_hashmap_size=14),
                         torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4).cuda().train()
        m.occupancy_grid.mark_all_occupied()
        rb = RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(64, 1, dtype=torch.long, device="cuda"))
        m.exec_get_outputs(rb)["rgb"].sum().backward()
        assert m.field.mlp_base_grid.dense_steps is want_dense
        assert m.field.mlp_base_grid.meta.bwd_tuning == (ops.HASH_BWD_DENSE_STEPS if want_dense else ops.HASH_BWD_DEFAULT)
    assert seen == [(3, 56), (8, 48)], seen
