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

from tests.util import TOL_FWD, TOL_GRAD, nmax_err, random_rays

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
    out = hip.route_outputs(hip.render_packed(rb, rs.ray_indices, ts, te, rs.packed_info), rb, ev_out=ev_out)
    oc, dc = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ref_raw = orc.render_samples(oc, dc, li.cpu(), ts.cpu(), te.cpu(), aid)
    ref = route_outputs(ref_raw["rgb"], training=True, ev_out=ev_out, **routing_kw)
    return out, ref, (og, dg), (oc, dc)


@pytest.mark.parametrize("emb_type", ["global_emb", "evs_emb"])     # cfg 2 / cfg 3
def test_config2_3_rgb_plus_events(emb_type):
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
    bundles = []
    for i, n in enumerate((n_col, n_ev, n_ev)):
        o, d = random_rays(n, seed=10 + i)
        aid = torch.randint(0, 16, (n,), generator=g) if emb_type == "evs_emb" else None
        bundles.append((o, d, aid))
    # prev / next: the same pixels seen from two nearby poses (two times)
    po, pd = bundles[1][0], bundles[1][1]
    nd = pd + 0.05 * torch.randn(n_ev, 3, generator=g)
    bundles[2] = (po + 0.03 * torch.randn(n_ev, 3, generator=g), nd / nd.norm(dim=-1, keepdim=True), bundles[2][2])
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
    for p in hip.parameters():
        p.grad = None
    for p in orc.field.parameters():
        p.grad = None
    sum(hl.values()).backward()
    sum(rl.values()).backward()
    fld = hip.field
    for k, p in {"grid": fld.mlp_base_grid.params, "base": fld.mlp_base_mlp.params, "head": fld.mlp_head.params,
                 "embedding": fld.embedding_appearance.embedding.weight}.items():
        assert nmax_err(p.grad, orc.field.params[k].grad, 1e-12) < TOL_GRAD, k
    # scalar mapper parameters: sums over all event rays with cancellation -> absolute floor
    assert nmax_err(hip.evs_mapper.pow_coeff.grad, pw.grad, 1e-3) < TOL_GRAD
    assert nmax_err(hip.rgb_to_one.weights.grad, tw.grad, 1e-3) < TOL_GRAD
    assert nmax_err(hin[0][0].grad, cin[0][0].grad) < TOL_GRAD and nmax_err(hin[1][1].grad, cin[1][1].grad) < TOL_GRAD


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
    assert nmax_err(og.grad, oc.grad) < TOL_GRAD and nmax_err(dg.grad, dc.grad) < TOL_GRAD


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
