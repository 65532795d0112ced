"""The outer drop-in boundary: the reference model's own call sequence against ``lsenerf_amd``.

``_ReferenceCallSites`` issues the calls of R:lse_nerf/lsenerf.py:158-228 (``populate_modules``) and :278-326
(``exec_get_outputs``) with the reference's argument names and values -- ``LSEField(..., spatial_distortion=
SceneContraction(order=inf), implementation="tcnn")``, ``VolumetricSampler(occupancy_grid=, density_fn=self.field.
density_fn)``, ``ray_samples.metadata = ...; self.field(ray_samples)``, ``nerfacc.pack_info`` / ``render_weight_from_density``,
the three renderers with ``ray_indices`` / ``num_rays`` -- after the three-line import swap INTEGRATION.md section A
describes.  The ray containers handed in are stand-ins named like nerfstudio's and carry ONLY nerfstudio's attributes
(no packed bookkeeping), as a stock caller's would.  Checked against the CPU oracle on the same samples (renders and
every parameter / ray gradient); parity unpinned (tests/util.py).
"""
from dataclasses import dataclass, field as dc_field
from typing import Dict, Optional

import pytest
import torch

from tests.util import (TOL_FWD, TOL_GRAD, make_model_pair, nmax_err, random_binaries, random_rays, rel_l2,
                        sync_params_to_oracle)

pytestmark = pytest.mark.gpu


# ---- stand-ins with nerfstudio's names and nerfstudio's attributes only ---------------------------------------
class SceneContraction:
    """nerfstudio.field_components.spatial_distortions.SceneContraction: just ``order`` (the field must not call it)."""

    def __init__(self, order=None):
        self.order = order

    def __call__(self, positions):
        raise AssertionError("the fused position kernel applies the contraction; the module must not be evaluated")


@dataclass
class SceneBox:
    aabb: torch.Tensor


@dataclass
class Frustums:
    origins: torch.Tensor
    directions: torch.Tensor
    starts: torch.Tensor
    ends: torch.Tensor
    pixel_area: Optional[torch.Tensor] = None

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def get_positions(self):
        return self.origins + self.directions * (self.starts + self.ends) / 2


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[torch.Tensor] = None
    deltas: Optional[torch.Tensor] = None
    metadata: Optional[Dict[str, torch.Tensor]] = None
    times: Optional[torch.Tensor] = None

    def __len__(self):
        return self.frustums.origins.shape[0]


@dataclass
class RayBundle:
    origins: torch.Tensor
    directions: torch.Tensor
    pixel_area: Optional[torch.Tensor] = None
    camera_indices: Optional[torch.Tensor] = None
    nears: Optional[torch.Tensor] = None
    fars: Optional[torch.Tensor] = None
    metadata: Dict[str, torch.Tensor] = dc_field(default_factory=dict)
    times: Optional[torch.Tensor] = None

    def __len__(self):
        return self.origins.shape[0]


# ---- the reference's call sites, after INTEGRATION.md's import swap ----------------------------------------------
import lsenerf_amd.nerfacc_compat as nerfacc  # noqa: E402   (reference: `import nerfacc`)
from lsenerf_amd import (AccumulationRenderer, DepthRenderer, LSEField, LSEOccGridEstimator, RGBRenderer,  # noqa: E402
                         VolumetricSampler)
from lsenerf_amd.field import FieldHeadNames  # noqa: E402


class _ReferenceCallSites(torch.nn.Module):
    def __init__(self, config, scene_box, num_train_data):
        super().__init__()
        self.config, self.scene_box, self.num_train_data = config, scene_box, num_train_data
        self.populate_modules()

    def populate_modules(self):                                   # R:lse_nerf/lsenerf.py:163-199
        scene_contraction = None if self.config.disable_scene_contraction else SceneContraction(order=float("inf"))
        self.field = LSEField(aabb=self.scene_box.aabb, num_images=self.num_train_data,
                              log2_hashmap_size=self.config.log2_hashmap_size, max_res=self.config.max_res,
                              spatial_distortion=scene_contraction, embd_config=self.config.embed_config,
                              implementation="tcnn")
        self.scene_aabb = torch.nn.Parameter(self.scene_box.aabb.flatten(), requires_grad=False)
        if self.config.render_step_size is None:
            self.config.render_step_size = ((self.scene_aabb[3:] - self.scene_aabb[:3]) ** 2).sum().sqrt().item() / 1000
        self.occupancy_grid = LSEOccGridEstimator(roi_aabb=self.scene_aabb, resolution=self.config.grid_resolution,
                                                  levels=self.config.grid_levels)
        self.sampler = VolumetricSampler(occupancy_grid=self.occupancy_grid, density_fn=self.field.density_fn)
        self.renderer_rgb = RGBRenderer(background_color=self.config.background_color)
        self.renderer_accumulation = AccumulationRenderer()
        self.renderer_depth = DepthRenderer(method="expected")

    def exec_get_outputs(self, ray_bundle):                       # R:lse_nerf/lsenerf.py:278-326
        assert self.field is not None
        num_rays = len(ray_bundle)
        ray_samples, ray_indices = self.sampler(ray_bundle=ray_bundle, near_plane=self.config.near_plane,
                                                far_plane=self.config.far_plane,
                                                render_step_size=self.config.render_step_size,
                                                alpha_thre=self.config.alpha_thre, cone_angle=self.config.cone_angle)
        metadata = {k: v[ray_indices] for k, v in ray_bundle.metadata.items()}
        ray_samples.metadata = metadata
        field_outputs = self.field(ray_samples)
        packed_info = nerfacc.pack_info(ray_indices, num_rays)
        weights = nerfacc.render_weight_from_density(t_starts=ray_samples.frustums.starts[..., 0],
                                                     t_ends=ray_samples.frustums.ends[..., 0],
                                                     sigmas=field_outputs[FieldHeadNames.DENSITY][..., 0],
                                                     packed_info=packed_info)[0]
        weights = weights[..., None]
        rgb = self.renderer_rgb(rgb=field_outputs[FieldHeadNames.RGB], weights=weights, ray_indices=ray_indices,
                                num_rays=num_rays)
        depth = self.renderer_depth(weights=weights, ray_samples=ray_samples, ray_indices=ray_indices, num_rays=num_rays)
        accumulation = self.renderer_accumulation(weights=weights, ray_indices=ray_indices, num_rays=num_rays)
        return {"rgb": rgb, "accumulation": accumulation, "depth": depth, "num_samples_per_ray": packed_info[:, 1],
                "_ray_samples": ray_samples, "_ray_indices": ray_indices, "_weights": weights}


def _reference_style_model(emb_type="evs_emb", n_emb=16, seed=96):
    from lsenerf_amd import LSEEmbeddingConfig, LSENeRFModelConfig
    from oracle.field import FieldOracle
    from oracle.model import ModelOracle
    torch.manual_seed(seed)
    cfg = LSENeRFModelConfig(grid_levels=2, grid_resolution=32, embed_config=LSEEmbeddingConfig(embedding_type=emb_type))
    box = SceneBox(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]))
    m = _ReferenceCallSites(cfg, box, n_emb)
    with torch.no_grad():
        m.field.mlp_base_grid.params.mul_(300.0)
    m = m.cuda()
    ne = m.field.embedding_appearance.embedding.weight.shape[0]
    f = FieldOracle("tcnn", num_embeddings=ne, contraction=True, aabb=box.aabb, seed=seed)
    sync_params_to_oracle(m, f)
    orc = ModelOracle(f, grid_resolution=32, grid_levels=2)
    b = random_binaries(2, 32, 0.5, seed)
    m.occupancy_grid.binaries.copy_(b.cuda())
    m.occupancy_grid.occs.copy_((b.float().flatten() * 0.5).cuda())
    orc.grid.binaries, orc.grid.occs = b.clone(), b.float().flatten() * 0.5
    return m, orc


def test_reference_constructor_surface():
    """R:lse_nerf/lse_field.py:124-160: every keyword of the reference is accepted, the reference's buffers are in the
    state dict, switched-off heads add nothing, switched-on heads carry the reference's module names, and a contraction that is
    not L-infinity is refused."""
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    f = LSEField(aabb=aabb, num_images=7, num_layers=2, hidden_dim=64, geo_feat_dim=15, num_levels=16, base_res=16,
                 max_res=2048, log2_hashmap_size=19, num_layers_color=3, num_layers_transient=2, features_per_level=2,
                 hidden_dim_color=64, hidden_dim_transient=64, appearance_embedding_dim=32, embd_config=None,
                 transient_embedding_dim=16, use_transient_embedding=False, use_semantics=False, num_semantic_classes=100,
                 pass_semantic_gradients=False, use_pred_normals=False, use_average_appearance_embedding=False,
                 spatial_distortion=SceneContraction(order=float("inf")), average_init_density=1.0, implementation="tcnn")
    sd = f.state_dict()
    assert int(sd["max_res"]) == 2048 and int(sd["num_levels"]) == 16 and int(sd["log2_hashmap_size"]) == 19
    assert f.spatial_distortion.order == float("inf")
    # the heads the reference's model leaves off build the reference's modules under the reference's names when switched on
    full = LSEField(aabb=aabb, num_images=5, use_transient_embedding=True, use_semantics=True, use_pred_normals=True,
                    num_semantic_classes=11)
    keys = set(full.state_dict())
    assert {"embedding_transient.embedding.weight", "mlp_transient.params", "field_head_transient_uncertainty.net.weight",
            "field_head_transient_rgb.net.bias", "field_head_transient_density.net.weight", "mlp_semantics.params",
            "field_head_semantics.net.weight", "mlp_pred_normals.params", "field_head_pred_normals.net.bias"} <= keys
    assert full.field_head_semantics.net.weight.shape == (11, 64) and not any(k.startswith("mlp_transient") for k in sd)
    with pytest.raises(NotImplementedError):
        LSEField(aabb=aabb, num_images=1, spatial_distortion=SceneContraction(order=2))
    with pytest.raises(NotImplementedError):
        LSEField(aabb=aabb, num_images=1, implementation="torch")


def _side_head_params(fld):
    names = ["embedding_transient.embedding.weight", "mlp_transient.params", "mlp_semantics.params", "mlp_pred_normals.params"]
    for h in ("field_head_transient_uncertainty", "field_head_transient_rgb", "field_head_transient_density", "field_head_semantics",
              "field_head_pred_normals"):
        names += [h + ".net.weight", h + ".net.bias"]
    sd = dict(fld.named_parameters())
    return {n: sd[n] for n in names}


@pytest.mark.parametrize("packed", [False, True])
def test_side_heads_match_oracle_and_feed_the_base_gradient(packed):
    """R:lse_nerf/lse_field.py:313-345 with every optional head switched on: uncertainty / transient rgb / transient density,
    semantics (gradients passed on) and predicted normals against the oracle, on stock per-sample RaySamples and on the packed
    samples of this package's sampler (one camera index per ray); the heads' gradients reach the hash table through the geometry
    features together with the fused RGB head's."""
    from oracle.field import FieldOracle, SideHeadsOracle
    from lsenerf_amd import RayBundle as HipRayBundle
    torch.manual_seed(5)
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    fld = LSEField(aabb=aabb, num_images=7, use_transient_embedding=True, use_semantics=True, pass_semantic_gradients=True,
                   use_pred_normals=True, num_semantic_classes=11, spatial_distortion=SceneContraction(order=float("inf")))
    with torch.no_grad():
        fld.mlp_base_grid.params.mul_(300.0)
    fld = fld.cuda().train()
    g = torch.Generator().manual_seed(6)
    if packed:
        R = 64
        o, d = random_rays(R, seed=3)
        cam_ray = torch.randint(0, 7, (R, 1), generator=g)
        occ = LSEOccGridEstimator(roi_aabb=aabb.flatten(), resolution=16, levels=1).cuda()
        occ.binaries.fill_(True)
        sampler = VolumetricSampler(occupancy_grid=occ, density_fn=None)
        rb = HipRayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=cam_ray.cuda())
        rs, ri = sampler(ray_bundle=rb, near_plane=0.05, far_plane=4.0, render_step_size=0.05, alpha_thre=0.0, cone_angle=0.0)
        N = len(rs)
        assert N > 500 and rs.ray_indices is not None
        ri_c = ri.cpu().long()
        o_s, d_s = o[ri_c], d[ri_c]
        st, en = rs.frustums.starts.detach().cpu(), rs.frustums.ends.detach().cpu()
        cams = cam_ray[ri_c]
    else:
        N = 777
        o_s = (torch.rand(N, 3, generator=g) - 0.5) * 3
        d_s = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
        st = torch.rand(N, 1, generator=g)
        en = st + 0.02
        cams = torch.randint(0, 7, (N, 1), generator=g)
        rs = RaySamples(Frustums(o_s.cuda(), d_s.cuda(), st.cuda(), en.cuda()), camera_indices=cams.cuda(), metadata={})
    outs = fld(rs)
    assert set(outs) == {FieldHeadNames.RGB, FieldHeadNames.DENSITY, FieldHeadNames.UNCERTAINTY, FieldHeadNames.TRANSIENT_RGB,
                         FieldHeadNames.TRANSIENT_DENSITY, FieldHeadNames.SEMANTICS, FieldHeadNames.PRED_NORMALS}
    # oracle on the same samples and parameters
    f = FieldOracle("tcnn", num_embeddings=1, contraction=True, aabb=aabb, seed=5)

    class _M:       # sync_params_to_oracle reads `.field`
        field = fld
    sync_params_to_oracle(_M, f)
    side_p = {n: v.detach().cpu().clone().requires_grad_(True) for n, v in _side_head_params(fld).items()}
    side = SideHeadsOracle(side_p)
    pos = o_s + d_s * (st + en) / 2
    dref, geo = f.get_density(pos)
    rref = f.get_outputs(d_s, geo, torch.zeros(N, dtype=torch.long))
    u, trgb, tden = side.transient(geo, cams)
    sem = side.semantics(geo, pass_gradients=True)
    nrm = side.pred_normals(pos, geo)
    want = {FieldHeadNames.RGB: rref, FieldHeadNames.DENSITY: dref, FieldHeadNames.UNCERTAINTY: u, FieldHeadNames.TRANSIENT_RGB: trgb,
            FieldHeadNames.TRANSIENT_DENSITY: tden, FieldHeadNames.SEMANTICS: sem, FieldHeadNames.PRED_NORMALS: nrm}
    for k, w in want.items():
        assert outs[k].shape == w.shape, (k, outs[k].shape, w.shape)
        assert nmax_err(outs[k].reshape(N, -1), w.reshape(N, -1)) < TOL_FWD, k
    # one scalar through every head: gradients of the side parameters and of the shared trunk
    coef = {k: torch.randn(w.shape, generator=g) for k, w in want.items()}
    sum((outs[k] * coef[k].cuda()).sum() for k in want).backward()
    sum((want[k] * coef[k]).sum() for k in want).backward()
    for n, p in _side_head_params(fld).items():
        assert nmax_err(p.grad, side_p[n].grad, 1e-12) < TOL_GRAD, n
    assert nmax_err(fld.mlp_base_grid.params.grad, f.params["grid"].grad, 1e-12) < TOL_GRAD
    assert nmax_err(fld.mlp_base_mlp.params.grad, f.params["base"].grad, 1e-12) < TOL_GRAD
    assert nmax_err(fld.mlp_head.params.grad, f.params["head"].grad, 1e-12) < TOL_GRAD
    # evaluation: the transient group is a training-only output (R:lse_nerf/lse_field.py:313)
    fld.eval()
    with torch.no_grad():
        ev = fld(rs)
    assert FieldHeadNames.UNCERTAINTY not in ev and FieldHeadNames.SEMANTICS in ev and FieldHeadNames.PRED_NORMALS in ev
    # semantics detached from the trunk by default
    fld2 = LSEField(aabb=aabb, num_images=1, use_semantics=True, spatial_distortion=SceneContraction(order=float("inf"))).cuda().train()
    rs2 = RaySamples(Frustums(o_s[:64].cuda(), d_s[:64].cuda(), st[:64].cuda(), en[:64].cuda()),
                     camera_indices=torch.zeros(64, 1, dtype=torch.long).cuda(), metadata={})
    fld2(rs2)[FieldHeadNames.SEMANTICS].sum().backward()
    assert fld2.mlp_semantics.params.grad is not None and fld2.mlp_base_grid.params.grad is None


@pytest.mark.parametrize("emb_type", ["global_emb", "evs_emb"])
def test_reference_call_sites_match_oracle(emb_type):
    m, orc = _reference_style_model(emb_type)
    m.train(); orc.training = True
    R = 160
    o, d = random_rays(R, seed=11)
    aid = torch.randint(0, 16, (R,), generator=torch.Generator().manual_seed(5))
    og, dg = o.clone().cuda().requires_grad_(True), d.clone().cuda().requires_grad_(True)
    rb = RayBundle(origins=og, directions=dg, camera_indices=torch.zeros(R, 1, dtype=torch.long).cuda(),
                   metadata={"appearance_id": aid.cuda()})
    torch.manual_seed(3)                              # the stratified jitter comes from the global RNG, as in the reference
    out = m.exec_get_outputs(rb)
    ri = out["_ray_indices"]
    assert ri.dtype == torch.int64                     # nerfstudio's VolumetricSampler returns int64 ray indices
    rs = out["_ray_samples"]
    ts, te = rs.frustums.starts[..., 0], rs.frustums.ends[..., 0]
    assert ts.shape[0] > 1000
    oc, dc = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ref = orc.render_samples(oc, dc, ri.cpu(), ts.detach().cpu(), te.detach().cpu(), aid if emb_type == "evs_emb" else None)
    assert torch.equal(out["num_samples_per_ray"].cpu(), ref["num_samples_per_ray"])
    for k in ("rgb", "accumulation", "depth"):
        assert out[k].shape == ref[k].shape
        assert nmax_err(out[k], ref[k], 1e-3) < 5 * TOL_FWD, k
    g = torch.Generator().manual_seed(1)
    wr, wa, wd = torch.rand(R, 3, generator=g), torch.rand(R, 1, generator=g), torch.rand(R, 1, generator=g)
    ((out["rgb"] * wr.cuda()).sum() + (out["accumulation"] * wa.cuda()).sum() + (out["depth"] * wd.cuda()).sum()).backward()
    ((ref["rgb"] * wr).sum() + (ref["accumulation"] * wa).sum() + (ref["depth"] * wd).sum()).backward()
    fld = m.field
    for k, p in {"grid": fld.mlp_base_grid.params, "base": fld.mlp_base_mlp.params, "head": fld.mlp_head.params,
                 "embedding": fld.embedding_appearance.embedding.weight}.items():
        assert nmax_err(p.grad, orc.field.params[k].grad, 1e-12) < TOL_GRAD, k
        assert rel_l2(p.grad, orc.field.params[k].grad) < TOL_GRAD, k
    assert nmax_err(og.grad, oc.grad, 1e-12) < TOL_GRAD and nmax_err(dg.grad, dc.grad, 1e-12) < TOL_GRAD


def test_stock_raysamples_take_the_generic_path():
    """R:lse_nerf/lsenerf.py:292-297 with containers that have nerfstudio's attributes and nothing else: per-sample
    origins / directions / camera indices / metadata, no packed bookkeeping -> no AttributeError, same values as the
    packed route, parameter gradients equal to the oracle's."""
    m, orc = _reference_style_model("evs_emb")
    m.train(); orc.training = True
    R = 96
    o, d = random_rays(R, seed=2)
    aid = torch.randint(0, 16, (R,), generator=torch.Generator().manual_seed(6))
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(R, 1, dtype=torch.long).cuda(),
                   metadata={"appearance_id": aid.cuda()})
    torch.manual_seed(4)
    packed_samples, ri = m.sampler(ray_bundle=rb, near_plane=m.config.near_plane, far_plane=m.config.far_plane,
                                   render_step_size=m.config.render_step_size, alpha_thre=m.config.alpha_thre,
                                   cone_angle=m.config.cone_angle)
    fr = packed_samples.frustums
    stock = RaySamples(frustums=Frustums(origins=rb.origins[ri], directions=rb.directions[ri], starts=fr.starts, ends=fr.ends),
                       camera_indices=rb.camera_indices[ri])
    stock.metadata = {k: v[ri] for k, v in rb.metadata.items()}
    assert not hasattr(stock, "ray_indices") and not hasattr(stock, "ray_bundle") and not hasattr(stock, "packed_info")
    outs = m.field(stock)
    packed_samples.metadata = stock.metadata
    outs_packed = m.field(packed_samples)
    for k in (FieldHeadNames.DENSITY, FieldHeadNames.RGB):
        assert outs[k].shape == outs_packed[k].shape
        assert nmax_err(outs[k], outs_packed[k]) < TOL_FWD
    pos = (o[ri.cpu()] + d[ri.cpu()] * (fr.starts.cpu() + fr.ends.cpu()) / 2)
    dref, geo = orc.field.get_density(pos)
    rref = orc.field.get_outputs(d[ri.cpu()], geo, aid[ri.cpu()])
    assert nmax_err(outs[FieldHeadNames.DENSITY].reshape(-1), dref.reshape(-1)) < TOL_FWD
    assert nmax_err(outs[FieldHeadNames.RGB], rref) < TOL_FWD
    (outs[FieldHeadNames.DENSITY].sum() + outs[FieldHeadNames.RGB].sum()).backward()
    (dref.sum() + rref.sum()).backward()
    fld = m.field
    for k, p in {"grid": fld.mlp_base_grid.params, "base": fld.mlp_base_mlp.params, "head": fld.mlp_head.params,
                 "embedding": fld.embedding_appearance.embedding.weight}.items():
        assert nmax_err(p.grad, orc.field.params[k].grad, 1e-12) < TOL_GRAD, k
    # a caller that copies the geometry features (e.g. `.contiguous()`, arithmetic) still gets the right colours
    dens, geo_feat = m.field.get_density(stock)
    rgb_a = m.field.get_outputs(stock, density_embedding=geo_feat)[FieldHeadNames.RGB]
    rgb_b = m.field.get_outputs(stock, density_embedding=geo_feat.contiguous() * 1.0)[FieldHeadNames.RGB]
    assert nmax_err(rgb_a, rgb_b) < TOL_FWD


def test_sampler_with_a_foreign_density_fn_and_alpha_fn():
    """VolumetricSampler(density_fn=<any positions -> density callable>) takes nerfstudio's generic sigma_fn route and gives
    the samples of the packed route; the estimator's alpha_fn branch (R:lse_nerf/lse_grid_estimator.py:128-138) culls like
    sigma_fn when fed the equivalent opacities."""
    m, _ = _reference_style_model("global_emb")
    m.train()
    R = 64
    o, d = random_rays(R, seed=8)
    rb = RayBundle(origins=o.cuda(), directions=d.cuda(), camera_indices=torch.zeros(R, 1, dtype=torch.long).cuda())
    kw = dict(near_plane=0.05, far_plane=1e3, render_step_size=m.config.render_step_size, alpha_thre=0.01, cone_angle=0.004)
    torch.manual_seed(9)
    a, ri_a = m.sampler(ray_bundle=rb, **kw)
    foreign = VolumetricSampler(occupancy_grid=m.occupancy_grid, density_fn=lambda pos: m.field.density_fn(pos))
    foreign.train()
    assert foreign._packed_field is None
    torch.manual_seed(9)
    b, ri_b = foreign(ray_bundle=rb, **kw)
    assert torch.equal(ri_a, ri_b) and torch.equal(a.frustums.starts, b.frustums.starts)
    # alpha_fn branch
    est = m.occupancy_grid
    jit = torch.rand(R, generator=torch.Generator().manual_seed(1)).cuda()
    common = dict(rays_o=rb.origins, rays_d=rb.directions, near_plane=0.05, far_plane=1e3,
                  render_step_size=m.config.render_step_size, alpha_thre=0.01, stratified=True, cone_angle=0.004, jitter=jit)

    def sigma_fn(ts, te, ri):
        return m.field.density_packed(rb.origins, rb.directions, ri.to(torch.int32), ts, te, None)[0]

    def alpha_fn(ts, te, ri):
        return 1.0 - torch.exp(-sigma_fn(ts, te, ri) * (te - ts))

    ri_s, ts_s, te_s = est.sampling(sigma_fn=sigma_fn, **common)
    ri_al, ts_al, te_al = est.sampling(alpha_fn=alpha_fn, **common)
    assert ri_s.dtype == torch.int64 and ts_s.shape[0] > 100
    # identical up to visibility-threshold flips on values within rounding of the threshold
    assert abs(ts_s.shape[0] - ts_al.shape[0]) <= max(2, ts_s.shape[0] // 2000)
    if ts_s.shape[0] == ts_al.shape[0]:
        assert torch.equal(ri_s, ri_al) and torch.equal(ts_s, ts_al)


def test_visibility_from_alpha_keeps_a_fully_opaque_sample():
    """nerfacc.render_visibility_from_alpha (R:lse_nerf/lse_grid_estimator.py:133): T_k = prod_{i<k} (1 - alpha_i), exclusive.
    alpha == 1.0f (what sigma * dt > ~17 rounds to) in the middle of a ray: that sample itself is still visible (its own
    transmittance is finite), everything behind it is culled; a ray that starts with alpha == 1 keeps exactly its first sample.
    Checked against a float64 exclusive cumprod (ragged rays, one spanning several 64-sample chunks)."""
    from lsenerf_amd import nerfacc_compat as nc
    g = torch.Generator().manual_seed(4)
    cnts = [5, 0, 200, 1, 70]
    alphas, want = [], []
    for r, c in enumerate(cnts):
        a = (torch.rand(c, generator=g) * 0.02).float()
        if r == 0:
            a[2] = 1.0
        if r == 2:
            a[130] = 1.0
        if r == 4:
            a[0] = 1.0
        T = torch.cat([torch.ones(1, dtype=torch.float64), torch.cumprod(1.0 - a.double(), 0)[:-1]]) if c else torch.zeros(0, dtype=torch.float64)
        alphas.append(a)
        want.append((T >= 1e-4) & (a >= 0.005).to(torch.bool))
    al = torch.cat(alphas).cuda()
    cnt = torch.tensor(cnts, dtype=torch.long)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).cuda().contiguous()
    got = nc.render_visibility_from_alpha(al, packed_info=packed, early_stop_eps=1e-4, alpha_thre=0.005).cpu()
    want = torch.cat(want)
    assert got.dtype == torch.bool and torch.equal(got, want)
    off = [0, 5, 5, 205, 206]
    assert bool(got[off[0] + 2]) and not got[off[0] + 3:off[0] + 5].any()       # the opaque sample survives, the rest does not
    assert bool(got[off[2] + 130]) and not got[off[2] + 131:off[2] + 200].any()
    assert bool(got[off[4]]) and not got[off[4] + 1:].any()
