"""BASELINE.json configs 2-4 as model-level parity cases on the GPU (scaled-down ray counts, same composition):
  cfg 2  LSENeRF scene: colour + prev/next event bundles, co_map routing (identity rgb mapper, powpow event mapper,
         learned ThreeToOne), log-intensity event loss            R:exp_configs/lsenerf_config.sh
  cfg 3  + per-event-frame appearance embedding (evs_emb)          R:exp_configs/lsenerf_emb_config.sh
  cfg 4  BAD-NeRF: rgb only, 4 virtual cameras per pixel averaged (deblur), gradients w.r.t. per-ray poses
                                                                    R:exp_configs/BADNERF_config.sh
The HIP model renders; the oracle re-renders the same packed samples on the CPU and applies its own restatement of the
routing/losses (oracle/losses.py); losses, routed outputs and all gradients are compared."""
import pytest
import torch

from tests.util import TOL_FWD, TOL_GRAD, nmax_err, per_ray_grad_check, random_rays, rel_l2

pytestmark = pytest.mark.gpu


def _pair(cfg_kw, emb_type="global_emb", n_train=16):
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, LSEEmbeddingConfig
    from oracle.field import FieldOracle
    from oracle.model import ModelOracle
    from tests.util import random_binaries, sync_params_to_oracle
    torch.manual_seed(96)
    cfg = LSENeRFModelConfig(grid_levels=2, grid_resolution=32, alpha_thre=0.0, log2_hashmap_size=15,
                             embed_config=LSEEmbeddingConfig(embedding_type=emb_type), **cfg_kw)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    hip = LSENeRFModel(cfg, aabb, n_train)
    with torch.no_grad():
        hip.field.mlp_base_grid.params.mul_(300.0)
    hip = hip.cuda().train()
    n_emb = hip.field.embedding_appearance.embedding.weight.shape[0]
    f = FieldOracle("tcnn", log2_hashmap_size=15, num_embeddings=n_emb, seed=1)
    sync_params_to_oracle(hip, f)
    orc = ModelOracle(f, grid_resolution=32, grid_levels=2, alpha_thre=0.0)
    b = random_binaries(2, 32, 0.5, 3)
    hip.occupancy_grid.binaries.copy_(b.cuda())
    orc.grid.binaries = b.clone()
    return hip, orc


def _render_both(hip, orc, o, d, aid, ev_out, routing_kw, seed):
    from lsenerf_amd import RayBundle
    from oracle.losses import route_outputs
    R = o.shape[0]
    og, dg = o.clone().cuda().requires_grad_(True), d.clone().cuda().requires_grad_(True)
    meta = {"appearance_id": aid.cuda()} if aid is not None else {}
    rb = RayBundle(origins=og, directions=dg, camera_indices=torch.zeros(R, 1, dtype=torch.long, device="cuda"), metadata=meta)
    jit = torch.rand(R, generator=torch.Generator().manual_seed(seed))
    rs, li = hip.sampler(ray_bundle=rb, near_plane=hip.config.near_plane, far_plane=hip.config.far_plane,
                         render_step_size=hip.config.render_step_size, alpha_thre=hip.config.alpha_thre,
                         cone_angle=hip.config.cone_angle, jitter=jit.cuda())
    ts, te = rs.frustums.starts[..., 0].contiguous(), rs.frustums.ends[..., 0].contiguous()
    raw = hip.render_packed(rb, rs.ray_indices, ts, te, rs.packed_info)
    out = hip.route_outputs(raw, rb, ev_out=ev_out)
    out["_raw"] = raw                                  # the un-routed render: input of the fused epilogue
    oc, dc = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ref_raw = orc.render_samples(oc, dc, li.cpu(), ts.cpu(), te.cpu(), aid)
    ref = route_outputs(ref_raw["rgb"], training=True, ev_out=ev_out, **routing_kw)
    return out, ref, (og, dg), (oc, dc)


def _event_cameras(n_cam=6, seed=4):
    """A short trajectory of pinhole cameras on a sphere of radius 1.4 looking at the scene centre (OpenGL frame: -z forward),
    with a per-camera appearance id as the data parser attaches it (R:lse_nerf/lse_parser.py)."""
    import numpy as np
    from lsenerf_amd import cameras as cam
    rng = np.random.default_rng(seed)
    ang = np.linspace(0.2, 1.1, n_cam) + 0.02 * rng.normal(size=n_cam)
    pos = 1.4 * np.stack([np.cos(ang), 0.3 * np.sin(2 * ang), np.sin(ang)], -1)
    c2w = np.zeros((n_cam, 3, 4), dtype=np.float32)
    for i, p in enumerate(pos):
        fwd = -p / np.linalg.norm(p)
        right = np.cross(fwd, np.array([0.0, 1.0, 0.0])); right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        c2w[i, :, 0], c2w[i, :, 1], c2w[i, :, 2], c2w[i, :, 3] = right, up, -fwd, p
    return cam.EdCameras(torch.from_numpy(c2w), 40.0, 40.0, 16.0, 16.0, 32, 32, times=torch.linspace(0, 1, n_cam),
                         metadata={"appearance_id": torch.arange(n_cam) % 16})


@pytest.mark.parametrize("emb_type,generator", [("global_emb", "consec"), ("evs_emb", "prevnext")])     # cfg 2 / cfg 3
def test_config2_3_rgb_plus_events(emb_type, generator):
    from lsenerf_amd import cameras as cam
    from oracle.losses import loss_dict
    hip, orc = _pair(dict(use_mapping=True, mapping_method="identity", map_mode="co_map", evs_mapping_method="powpow",
                          ev_one_dim="learned"), emb_type=emb_type)
    with torch.no_grad():
        hip.evs_mapper.pow_coeff.fill_(0.8)
        hip.rgb_to_one.weights.copy_(torch.tensor([[0.2, 0.5, 0.3]]))
    pw = hip.evs_mapper.pow_coeff.detach().cpu().clone().requires_grad_(True)
    tw = hip.rgb_to_one.weights.detach().cpu().clone().requires_grad_(True)
    routing = dict(use_mapping=True, map_mode="co_map", rgb_loss_type="linspace", rgb_mapper=lambda x: x,
                   evs_mapper=lambda x: x ** pw, three_to_one_w=tw)
    g = torch.Generator().manual_seed(5)
    n_col, n_ev = 232, 60                                       # 2316 / 597 / 597 scaled by 10 (R:lse_datamanager.py:135-144)
    o, d = random_rays(n_col, seed=10)
    bundles = [(o, d, torch.randint(0, 16, (n_col,), generator=g) if emb_type == "evs_emb" else None)]
    # prev / next event bundles from the reference's generators (R:lse_nerf/lse_ray_generator.py:36-100): the same pixels
    # seen from camera c and c + 1 (ConsecRayGenerator) or from a start-of-window / end-of-window camera set
    # (PrevNextRayGenerator), corrected by PrevNextCamOptimizer like the data manager does
    cams = _event_cameras()
    px = torch.stack([torch.randint(0, 5, (n_ev,), generator=g), torch.randint(0, 32, (n_ev,), generator=g),
                      torch.randint(0, 32, (n_ev,), generator=g)], -1)
    if generator == "consec":
        rb_prev, rb_next = cam.ConsecRayGenerator(cams)(px)
    else:
        cams_next = _event_cameras(seed=9)
        rb_prev, rb_next = cam.PrevNextRayGenerator(cams, cams_next)(px)
        pn = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="prevnext").setup(num_cameras=len(cams), device="cpu")
        with torch.no_grad():
            pn.prev_optim.pose_adjustment.normal_(0, 0.01, generator=g)
            pn.next_optim.pose_adjustment.normal_(0, 0.01, generator=g)
        pn.apply_to_raybundle(rb_prev); pn.apply_to_raybundle(rb_next)
    for rbe in (rb_prev, rb_next):
        aid = rbe.metadata["appearance_id"] if emb_type == "evs_emb" else None
        bundles.append((rbe.origins.detach().contiguous(), rbe.directions.detach().contiguous(), aid))
    assert not torch.equal(bundles[1][0], bundles[2][0])
    outs, refs, hin, cin = [], [], [], []
    for i, (o, d, aid) in enumerate(bundles):
        a, b, hi, ci = _render_both(hip, orc, o, d, aid, ev_out=(i > 0), routing_kw=routing, seed=20 + i)
        outs.append(a); refs.append(b); hin.append(hi); cin.append(ci)
    col_gt = torch.rand(n_col, 3, generator=g)
    evs_gt = (torch.rand(n_ev, 1, generator=g) - 0.5) * 0.4
    hl = hip.get_loss_dict({"col_out": outs[0], "prev_out": outs[1], "next_out": outs[2]},
                           {"col_batch": {"image": col_gt.cuda()}, "evs_batch": {"image": evs_gt.cuda()}})
    rl = loss_dict(refs[0], refs[1], refs[2], col_gt, evs_gt, use_mapping=True)
    assert set(hl) == set(rl) == {"rgb_loss", "event_loss"}
    for k in hl:
        assert abs(float(hl[k].detach()) - float(rl[k].detach())) < 5e-5 * max(1.0, abs(float(rl[k].detach()))), k
    assert nmax_err(outs[1]["ev_out"], refs[1]["ev_out"], 1e-3) < 5 * TOL_FWD
    # the training step's route: the same three renders through the fused epilogue kernels
    fl = hip.fused_loss_dict({"col_out": outs[0]["_raw"], "prev_out": outs[1]["_raw"], "next_out": outs[2]["_raw"]},
                             {"col_batch": {"image": col_gt.cuda()}, "evs_batch": {"image": evs_gt.cuda()}})
    assert set(fl) == set(hl)
    for k in hl:
        assert abs(float(fl[k].detach()) - float(hl[k].detach())) < 1e-5 * max(1.0, abs(float(hl[k].detach()))), k
    for p in hip.parameters():
        p.grad = None
    for p in orc.field.parameters():
        p.grad = None
    sum(fl.values()).backward()                       # backward through lse_loss_epilogue_bwd into the HIP render graph
    sum(rl.values()).backward()
    fld = hip.field
    for k, p in {"grid": fld.mlp_base_grid.params, "base": fld.mlp_base_mlp.params, "head": fld.mlp_head.params,
                 "embedding": fld.embedding_appearance.embedding.weight}.items():
        assert nmax_err(p.grad, orc.field.params[k].grad, 1e-12) < TOL_GRAD, k
    # scalar mapper parameters: sums over all event rays with cancellation -> absolute floor
    assert nmax_err(hip.evs_mapper.pow_coeff.grad, pw.grad, 1e-3) < TOL_GRAD
    assert nmax_err(hip.rgb_to_one.weights.grad, tw.grad, 1e-3) < TOL_GRAD
    # ray gradients: d(field)/d(x) is piecewise constant in every hash cell, so a sample whose GPU- and CPU-computed position
    # straddle a cell face of a fine level moves ONE ray's gradient by a visible step: bound the worst ray loosely and the
    # whole tensor (relative L2) tightly
    from tests.util import rel_l2
    assert nmax_err(hin[0][0].grad, cin[0][0].grad) < 2 * TOL_GRAD and nmax_err(hin[1][1].grad, cin[1][1].grad) < 2 * TOL_GRAD
    assert rel_l2(hin[0][0].grad, cin[0][0].grad) < TOL_GRAD and rel_l2(hin[1][1].grad, cin[1][1].grad) < TOL_GRAD


@pytest.mark.parametrize("case", ["co_map_powpow_learned", "evs_rgb_gt_gray", "rgb_evs_powpow", "plain_rgb_key", "deblur_co_map",
                                  "co_map_rgb_mlp_mlp_learned", "rgb_evs_rgb_mlp", "deblur_evs_rgb_rgb_mlp_gray",
                                  "co_map_powpow_rgb_mlp_events", "enerf_co_map_powpow_learned", "enerf_rgb_evs_gt",
                                  "enerf_co_map_rgb_mlp_mlp_learned"])
def test_fused_loss_epilogue_matches_torch_routing_and_oracle(case, monkeypatch):
    """lse_loss_epilogue_fwd / _bwd (one launch each way) against (i) the model's own torch routing + losses and (ii) the
    oracle's restatement (oracle/losses.py), values and every gradient: rendered radiance of the three bundles, powpow
    coefficients, ThreeToOne weights, the 593 / 659 parameters of the MLP mappers (R:lse_nerf/intensity_mappers.py:28-62; their
    weight gradients are summed on the f32 matrix core, csrc/epilogue.hip) -- over the map modes / mappers / one-dim choices of
    R:lse_nerf/lsenerf.py:329-439."""
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig
    from lsenerf_amd import model as M
    from oracle.losses import loss_dict, mlp_mapper, route_outputs
    monkeypatch.setattr(M.MLP_Mapper, "init_steps", 60)          # a partly fitted mapper: far from the identity, ReLUs on both sides
    monkeypatch.setattr(M.RGB_MLP_Mapper, "init_steps", 60)
    kw = {"co_map_powpow_learned": dict(use_mapping=True, mapping_method="powpow", map_mode="co_map", evs_mapping_method="powpow", ev_one_dim="learned"),
          "evs_rgb_gt_gray": dict(use_mapping=True, mapping_method="gt", map_mode="evs_rgb", ev_one_dim="gt"),
          "rgb_evs_powpow": dict(use_mapping=True, mapping_method="powpow", map_mode="rgb_evs", ev_one_dim=False),
          "plain_rgb_key": dict(use_mapping=False, ev_one_dim=False, evs_loss_weight=0.7),
          "deblur_co_map": dict(use_mapping=True, mapping_method="identity", map_mode="co_map", evs_mapping_method="gt",
                                ev_one_dim="learned", rgb_loss_type="deblur"),
          "co_map_rgb_mlp_mlp_learned": dict(use_mapping=True, mapping_method="rgb_mlp", map_mode="co_map", evs_mapping_method="mlp",
                                             ev_one_dim="learned"),
          "rgb_evs_rgb_mlp": dict(use_mapping=True, mapping_method="rgb_mlp", map_mode="rgb_evs", ev_one_dim=False),
          "deblur_evs_rgb_rgb_mlp_gray": dict(use_mapping=True, mapping_method="rgb_mlp", map_mode="evs_rgb", ev_one_dim="gt",
                                              rgb_loss_type="deblur"),
          "co_map_powpow_rgb_mlp_events": dict(use_mapping=True, mapping_method="powpow", map_mode="co_map",
                                               evs_mapping_method="rgb_mlp", ev_one_dim=False),
          # enerf_norm_loss (R:lse_nerf/lsenerf.py:406-419) inside the fused epilogue: closed-form and MLP mappers
          "enerf_co_map_powpow_learned": dict(use_mapping=True, mapping_method="powpow", map_mode="co_map", evs_mapping_method="powpow",
                                              ev_one_dim="learned", event_loss_type="enerf_norm_loss", evs_loss_weight=3.0),
          "enerf_rgb_evs_gt": dict(use_mapping=True, mapping_method="gt", map_mode="rgb_evs", ev_one_dim=False,
                                   event_loss_type="enerf_norm_loss"),
          "enerf_co_map_rgb_mlp_mlp_learned": dict(use_mapping=True, mapping_method="rgb_mlp", map_mode="co_map", evs_mapping_method="mlp",
                                                   ev_one_dim="learned", event_loss_type="enerf_norm_loss")}[case]
    torch.manual_seed(0)
    cfg = LSENeRFModelConfig(grid_levels=1, grid_resolution=16, num_levels=4, log2_hashmap_size=12, **kw)
    m = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 4).cuda().train()
    g = torch.Generator().manual_seed(1)
    G = 4 if cfg.rgb_loss_type == "deblur" else 1
    n_col, n_ev = 1000, 333
    raw = {k: (torch.rand(n, 3, generator=g) * 1.2 - 0.02) for k, n in (("col", n_col * G), ("prev", n_ev), ("next", n_ev))}
    raw["prev"][:5] = 0.0                                   # below the 1e-5 clamp: zero gradient there
    col_gt, evs_gt = torch.rand(n_col, 3, generator=g), (torch.rand(n_ev, 1, generator=g) - 0.5) * 0.4
    pw = {}
    with torch.no_grad():
        if isinstance(getattr(m, "rgb_mapper", None), torch.nn.Module) and hasattr(m.rgb_mapper, "pow_coeff"):
            m.rgb_mapper.pow_coeff.fill_(0.6)
            pw["rgb"] = torch.tensor([0.6], requires_grad=True)
        if m.evs_mapper is not None and hasattr(m.evs_mapper, "pow_coeff"):
            m.evs_mapper.pow_coeff.fill_(0.8)
            pw["evs"] = torch.tensor([0.8], requires_grad=True)
        if cfg.ev_one_dim == "learned":
            m.rgb_to_one.weights.copy_(torch.tensor([[0.2, 0.5, 0.3]]))
    tw = torch.tensor([[0.2, 0.5, 0.3]], requires_grad=True) if cfg.ev_one_dim == "learned" else None
    e_thresh = 0.15 + 0.1 * torch.rand(n_ev, 1, generator=g)          # per event ray, R:lse_nerf/lse_pixel_sampler.py:36-37
    batch = {"col_batch": {"image": col_gt.cuda()}, "evs_batch": {"image": evs_gt.cuda(), "e_thresh": e_thresh.cuda()}}

    def leaves():
        return {k: v.clone().cuda().requires_grad_(True) for k, v in raw.items()}

    # (a) fused kernels
    assert m._epilogue_desc() is not None
    la = leaves()
    fused = m.fused_loss_dict({"col_out": {"rgb": la["col"]}, "prev_out": {"rgb": la["prev"]}, "next_out": {"rgb": la["next"]}}, batch)
    (fused["rgb_loss"] * 1.3 + fused["event_loss"] * 0.6).backward()
    fused_param_grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    # (b) the model's torch routing
    lb = leaves()
    routed = {"col_out": m.route_outputs({"rgb": lb["col"]}, None), "prev_out": m.route_outputs({"rgb": lb["prev"]}, None, ev_out=True),
              "next_out": m.route_outputs({"rgb": lb["next"]}, None, ev_out=True)}
    tl = m.get_loss_dict(routed, batch)
    (tl["rgb_loss"] * 1.3 + tl["event_loss"] * 0.6).backward()
    # (c) the oracle
    lc = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    mapper = {"identity": (lambda x: x), "gt": (lambda x: x ** (1 / 2.4)), "powpow": None}
    method = cfg.mapping_method if cfg.use_mapping else "identity"        # (the config's default "mlp" is only read with use_mapping)
    omlp = {}                                                             # the oracle's copies of the MLP mappers' parameters
    for side, mod in (("rgb", getattr(m, "rgb_mapper", None)), ("evs", m.evs_mapper)):
        if isinstance(mod, (M.MLP_Mapper, M.RGB_MLP_Mapper)):
            omlp[side] = [p.detach().cpu().clone().requires_grad_(True) for p in mod.parameters()]
    if method in ("mlp", "rgb_mlp"):
        rgb_mapper = mlp_mapper(omlp["rgb"])
    else:
        rgb_mapper = (lambda x: x ** pw["rgb"]) if method == "powpow" else mapper[method]
    evs_mapper = None
    if cfg.evs_mapping_method in ("mlp", "rgb_mlp"):
        evs_mapper = mlp_mapper(omlp["evs"])
    elif cfg.evs_mapping_method is not None:
        evs_mapper = (lambda x: x ** pw["evs"]) if cfg.evs_mapping_method == "powpow" else mapper[cfg.evs_mapping_method]
    gray_w = torch.log(torch.tensor([[0.2989, 0.5870, 0.1140]]))        # softmax(log w) == w: ToGrayGT through the same formula
    rkw = dict(training=True, use_mapping=cfg.use_mapping, map_mode=cfg.map_mode, rgb_loss_type=cfg.rgb_loss_type,
               rgb_mapper=rgb_mapper, evs_mapper=evs_mapper,
               three_to_one_w=tw if cfg.ev_one_dim == "learned" else (gray_w if cfg.ev_one_dim == "gt" else None))
    oref = [route_outputs(lc[k], ev_out=(k != "col"), **rkw) for k in ("col", "prev", "next")]
    rl = loss_dict(oref[0], oref[1], oref[2], col_gt, evs_gt, use_mapping=cfg.use_mapping, evs_loss_weight=cfg.evs_loss_weight,
                   event_loss=cfg.event_loss_type, e_thresh=e_thresh)
    (rl["rgb_loss"] * 1.3 + rl["event_loss"] * 0.6).backward()
    for k in ("rgb_loss", "event_loss"):
        assert abs(float(fused[k]) - float(tl[k])) < 1e-5 * max(1.0, abs(float(tl[k]))), (k, float(fused[k]), float(tl[k]))
        assert abs(float(fused[k]) - float(rl[k])) < 1e-5 * max(1.0, abs(float(rl[k]))), (k, float(fused[k]), float(rl[k]))
    for k in raw:
        assert nmax_err(la[k].grad, lb[k].grad) < 1e-5, k
        assert nmax_err(la[k].grad, lc[k].grad) < 1e-5, k
    assert float(la["prev"].grad[:5].abs().max()) == 0.0
    torch_param_grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert set(fused_param_grads) == set(torch_param_grads)
    # enerf_norm_loss divides the log-intensity change by its own norm, and a powpow exponent on the one-channel event side only
    # SCALES that change (log s^p = p log s): the loss does not depend on the exponent but for the two EPS terms, its gradient is a sum
    # of cancelling O(1e-3) terms that ends at ~6e-6 -- compared on the scale of the summands there, not of the residue
    floor_of = lambda n: 1e-4 if (cfg.event_loss_type == "enerf_norm_loss" and n == "evs_mapper.pow_coeff") else 1e-6
    for n, gten in fused_param_grads.items():
        assert nmax_err(gten, torch_param_grads[n], floor_of(n)) < 1e-4, n
    if "evs" in pw:
        assert nmax_err(fused_param_grads["evs_mapper.pow_coeff"], pw["evs"].grad, floor_of("evs_mapper.pow_coeff")) < 1e-4
    if "rgb" in pw and pw["rgb"].grad is not None:
        assert nmax_err(fused_param_grads["rgb_mapper.pow_coeff"], pw["rgb"].grad, 1e-6) < 1e-4
    if tw is not None:
        assert nmax_err(fused_param_grads["rgb_to_one.weights"], tw.grad, 1e-6) < 1e-4
    for side, plist in omlp.items():                      # every layer of an MLP mapper against the oracle's autograd
        names = [f"{side}_mapper.mlp.layers.{i}.{k}" for i in range(4) for k in ("weight", "bias")]
        assert any(float(p.grad.abs().max()) > 0 for p in plist if p.grad is not None), side
        for n, p in zip(names, plist):
            assert p.grad is not None and nmax_err(fused_param_grads[n], p.grad, 1e-6) < 1e-4, n


def test_config4_badnerf_deblur_pose_gradients():
    """rgb_frac = 1, 4 virtual cameras per pixel: rgb = mean over the 4 renders; the loss gradient must reach the per-ray
    origins/directions (what ns_camera_optimizer consumes, R:lse_nerf/ns_camera_optimizer.py:322-329)."""
    import torch.nn.functional as F
    hip, orc = _pair(dict(rgb_loss_type="deblur", use_mapping=False))
    n_px = 88                                                    # 878 pixels x 4 = 3512 rays, scaled by 10
    o, d = random_rays(n_px, seed=3)
    g = torch.Generator().manual_seed(7)
    o4 = (o[:, None, :] + 0.01 * torch.randn(n_px, 4, 3, generator=g)).reshape(-1, 3).contiguous()
    d4 = d[:, None, :] + 0.01 * torch.randn(n_px, 4, 3, generator=g)
    d4 = (d4 / d4.norm(dim=-1, keepdim=True)).reshape(-1, 3).contiguous()
    routing = dict(use_mapping=False, map_mode="evs_rgb", rgb_loss_type="deblur")
    out, ref, (og, dg), (oc, dc) = _render_both(hip, orc, o4, d4, None, ev_out=False, routing_kw=routing, seed=1)
    assert out["rgb"].shape == (n_px, 3) and nmax_err(out["rgb"], ref["rgb"], 1e-3) < 5 * TOL_FWD
    gt = torch.rand(n_px, 3, generator=g)
    hl = hip.get_loss_dict({"col_out": out, "prev_out": None, "next_out": None},
                           {"col_batch": {"image": gt.cuda()}, "evs_batch": None})
    rl = F.mse_loss(gt, ref["rgb"])
    assert abs(float(hl["rgb_loss"]) - float(rl)) < 5e-5
    hl["rgb_loss"].backward()
    rl.backward()
    assert float(oc.grad.abs().max()) > 0
    per_ray_grad_check(og.grad, oc.grad)         # (all rays within TOL_GRAD but at most one: tests/util.py explains the one)
    per_ray_grad_check(dg.grad, dc.grad)


def test_config4_pose_parameters_receive_gradients_through_the_hot_path():
    """cfg 4 end to end: spline deblur cameras -> rays -> HIP field/sampler/renderer -> deblur mean -> loss; the gradient
    w.r.t. the spline control tangents must equal the oracle's (ray generation is shared torch code, checked vs scipy)."""
    import numpy as np
    from lsenerf_amd import cameras as cam
    from tests.test_cameras_cpu import gen_data
    hip, orc = _pair(dict(rgb_loss_type="deblur", use_mapping=False))
    c2w, ts = gen_data(6, max_t=1.0, seed=2)
    c2w[:, :3, 3] *= 0.3                                         # cameras inside the scene box

    def make(device):
        cams = cam.EdCameras(torch.from_numpy(c2w), 60.0, 60.0, 16.0, 16.0, 32, 32, times=torch.from_numpy(ts))
        spl = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="spline", exp_t=0.05).setup(
            num_cameras=6, device="cpu", cameras=cams, dM=torch.eye(4))
        spl = spl.to(device)
        spl.device = device
        cams.times = cams.times.to(device)
        return cams, spl
    g = torch.Generator().manual_seed(1)
    n_px = 48
    ci = torch.randint(1, 5, (n_px,), generator=g)
    coords = torch.randint(0, 32, (n_px, 2), generator=g).float()
    gt = torch.rand(n_px, 3, generator=g)
    jit = torch.rand(n_px * 4, generator=g)
    routing = dict(use_mapping=False, map_mode="evs_rgb", rgb_loss_type="deblur")
    # --- HIP
    cams_g, spl_g = make("cuda")
    rb = cam.generate_deblur_rays(cams_g, spl_g, ci.cuda(), coords.cuda())
    rb.origins.retain_grad(); rb.directions.retain_grad()
    rs, li = hip.sampler(ray_bundle=rb, near_plane=0.05, far_plane=1e3, render_step_size=hip.config.render_step_size,
                         alpha_thre=0.0, cone_angle=hip.config.cone_angle, jitter=jit.cuda())
    ts_, te_ = rs.frustums.starts[..., 0].contiguous(), rs.frustums.ends[..., 0].contiguous()
    out = hip.route_outputs(hip.render_packed(rb, rs.ray_indices, ts_, te_, rs.packed_info), rb)
    loss = torch.nn.functional.mse_loss(out["rgb"], gt.cuda())
    loss.backward()
    # --- oracle on the same samples AND the same ray values (GPU- and CPU-evaluated splines differ by ~1e-7, which would
    # flip a few samples across fine hash-cell faces where the field gradient is piecewise constant)
    from oracle.losses import route_outputs
    oc = rb.origins.detach().cpu().clone().requires_grad_(True)
    dc = rb.directions.detach().cpu().clone().requires_grad_(True)
    ref = route_outputs(orc.render_samples(oc, dc, li.cpu(), ts_.cpu(), te_.cpu(), None)["rgb"], training=True, ev_out=False,
                        **routing)
    lref = torch.nn.functional.mse_loss(ref["rgb"], gt)
    lref.backward()
    assert abs(float(loss.detach()) - float(lref.detach())) < 5e-5
    assert nmax_err(rb.origins.grad, oc.grad) < TOL_GRAD and nmax_err(rb.directions.grad, dc.grad) < TOL_GRAD
    # chain the oracle's ray gradients through the CPU camera graph: must reproduce the GPU's control-tangent gradient
    cams_c, spl_c = make("cpu")
    rbc = cam.generate_deblur_rays(cams_c, spl_c, ci, coords)
    assert nmax_err(rb.origins, rbc.origins) < 1e-5 and nmax_err(rb.directions, rbc.directions) < 1e-5
    torch.autograd.backward([rbc.origins, rbc.directions], [oc.grad, dc.grad])
    assert float(spl_c.ctrl_tangents.grad.abs().max()) > 0
    assert nmax_err(spl_g.ctrl_tangents.grad, spl_c.ctrl_tangents.grad) < 2 * TOL_GRAD


@pytest.mark.parametrize("emb_type", ["global_emb", "evs_emb"])
def test_train_step_bundles_equals_three_separate_passes(emb_type):
    """The reference's training step (R:lse_nerf/lse_pipeline.py:110-145: colour, previous-event and next-event bundle, one
    model forward each) as ONE packed pass (LSENeRFModel.train_step_bundles): with the same stratified offsets every ray gets
    the same samples and the same rendered values as in a pass of its own -- asserted exactly for the sample counts, to
    rounding for the renders -- the losses agree, and every gradient (table, MLPs, embedding, mapper scalars, per-ray poses)
    agrees within the gradient tolerance (one hash scatter over all bundles sums in a different order than three).  Default
    configuration of the path: cone 0.004, alpha_thre 0.01 -> visibility pre-pass on."""
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, LSEEmbeddingConfig, RayBundle
    from tests.util import random_binaries
    torch.manual_seed(96)
    cfg = LSENeRFModelConfig(grid_levels=2, grid_resolution=32, log2_hashmap_size=15, use_mapping=True, mapping_method="identity",
                             map_mode="co_map", evs_mapping_method="powpow", ev_one_dim="learned",
                             embed_config=LSEEmbeddingConfig(embedding_type=emb_type))
    hip = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), 16)
    with torch.no_grad():
        hip.field.mlp_base_grid.params.mul_(300.0)
        hip.evs_mapper.pow_coeff.fill_(0.8)
    hip = hip.cuda().train()
    hip.occupancy_grid.binaries.copy_(random_binaries(2, 32, 0.5, 3).cuda())
    hip.occupancy_grid.occs.copy_(hip.occupancy_grid.binaries.flatten().float() * 0.5)
    g = torch.Generator().manual_seed(11)
    sizes = (232, 60, 60)                                       # 2316 / 597 / 597 scaled by 10
    bundles, jit = [], []
    for i, n in enumerate(sizes):
        o, d = random_rays(n, seed=30 + i)
        meta = {"appearance_id": torch.randint(0, 16, (n,), generator=g).cuda()}
        bundles.append(RayBundle(origins=o.cuda().requires_grad_(True), directions=d.cuda().requires_grad_(True),
                                 camera_indices=torch.zeros(n, 1, dtype=torch.long, device="cuda"), metadata=meta))
        jit.append(torch.rand(n, generator=g).cuda())
    batch = {"col_batch": {"image": torch.rand(sizes[0], 3, generator=g).cuda()},
             "evs_batch": {"image": ((torch.rand(sizes[1], 1, generator=g) - 0.5) * 0.4).cuda()}}
    params = {"grid": hip.field.mlp_base_grid.params, "base": hip.field.mlp_base_mlp.params, "head": hip.field.mlp_head.params,
              "emb": hip.field.embedding_appearance.embedding.weight, "pow": hip.evs_mapper.pow_coeff, "w31": hip.rgb_to_one.weights}

    def grads():
        out = {k: p.grad.clone() for k, p in params.items()}
        out.update({f"o{i}": b.origins.grad.clone() for i, b in enumerate(bundles)})
        out.update({f"d{i}": b.directions.grad.clone() for i, b in enumerate(bundles)})
        for p in params.values():
            p.grad = None
        for b in bundles:
            b.origins.grad = b.directions.grad = None
        return out

    # (i) the reference's composition: three forwards
    raws = [hip.exec_get_outputs(b, jitter=j) for b, j in zip(bundles, jit)]
    l3 = hip.fused_loss_dict({"col_out": raws[0], "prev_out": raws[1], "next_out": raws[2]}, batch)
    sum(l3.values()).backward()
    g3 = grads()
    # (ii) one packed pass
    out, l1, metrics = hip.train_step_bundles(bundles[0], bundles[1], bundles[2], batch, jitter=torch.cat(jit), with_metrics=True)
    sum(l1.values()).backward()
    g1 = grads()
    assert set(l1) == set(l3) == {"rgb_loss", "event_loss"}
    for k, r in zip(("col_out", "prev_out", "next_out"), raws):
        assert torch.equal(out[k]["num_samples_per_ray"], r["num_samples_per_ray"]), k
        assert int(r["num_samples_per_ray"].sum()) > 20 * r["rgb"].shape[0]
        for key in ("rgb", "accumulation", "depth"):
            assert nmax_err(out[k][key], r[key]) < 1e-6, (k, key)
    for k in l3:
        assert abs(float(l1[k]) - float(l3[k])) < 1e-6 * max(1.0, abs(float(l3[k]))), k
    for k in g3:
        assert nmax_err(g1[k], g3[k], 1e-12) < TOL_GRAD, k
    assert set(metrics) == {"col_psnr", "col_num_samples_per_batch"}
    assert int(metrics["col_num_samples_per_batch"]) == int(raws[0]["num_samples_per_ray"].sum())
    # colour-only step (BASELINE config 4's composition: no event bundles)
    out_c, l_c, _ = hip.train_step_bundles(bundles[0], None, None, {"col_batch": batch["col_batch"], "evs_batch": None}, jitter=jit[0])
    assert set(l_c) == {"rgb_loss"} and out_c["prev_out"] is None and abs(float(l_c["rgb_loss"]) - float(l3["rgb_loss"])) < 1e-6
    with pytest.raises(ValueError, match="pairs"):
        hip.train_step_bundles(bundles[0], bundles[1], None, batch)
