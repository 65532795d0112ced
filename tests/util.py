"""Shared helpers for the parity tests: seeded inputs, HIP-model/oracle pairs with identical parameters, error metrics.

The oracle (``oracle/``) is the checker only; everything named ``hip_*`` runs through the C-ABI on the GPU.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

# stated tolerances (normalised max error: max|a-b| / max(floor, max|b|))
TOL_FWD = 2e-5     # forward activations / renders, fp32 with different summation orders
TOL_GRAD = 3e-4    # gradients: long sums (N up to 1e5 terms) in different orders, float atomics
TOL_GRAD_BLOCK = 1e-3   # the same per block (hash level / MLP row / embedding row), each scaled by its OWN maximum


def nmax_err(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-6) -> float:
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / max(floor, float(b.abs().max())))


def rel_l2(a: torch.Tensor, b: torch.Tensor, floor: float = 1e-30) -> float:
    """||a - b||_2 / ||b||_2 in float64."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).norm() / max(floor, float(b.norm())))


def per_ray_grad_check(got: torch.Tensor, ref: torch.Tensor, tol: float = None, max_outliers: int = 1, outlier_tol: float = 1e-2):
    """Per-ray gradients [R, 3]: every ray within ``tol`` of the reference (error of the ray's worst component over the global
    maximum), except at most ``max_outliers`` rays which may be off by up to ``outlier_tol``.
    Why outliers exist at all: a ReLU gate is a step function of its pre-activation h.  Two correct float32 evaluations of h
    differ in the last bits, so when |h| < ~1e-7 * scale for some (sample, neuron) one of them opens the gate and the other does
    not, and that ONE ray's gradient changes by an O(1e-3) amount while every other ray agrees to 1e-5.  With ~1e7
    (sample, neuron) pairs per test this happens to about one pair per seed; which pair depends on the summation order of the
    implementation (f32 MFMA chain, bf16-piece products, torch's GEMM), not on its correctness."""
    tol = TOL_GRAD if tol is None else tol
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    e = (got - ref).abs().amax(-1) / ref.abs().max().clamp_min(1e-30)
    bad = int((e > tol).sum())
    assert bad <= max_outliers, f"{bad} of {e.numel()} rays beyond {tol}: worst {float(e.max()):.3e}"
    assert float(e.max()) < outlier_tol, f"outlier ray error {float(e.max()):.3e} >= {outlier_tol}"
    return float(e.max()), bad


def blockwise_nmax_err(a: torch.Tensor, b: torch.Tensor, bounds, rel_floor: float = 1e-4) -> float:
    """max over blocks [bounds[i], bounds[i+1]) of  max|a - b| / max(max|b| in the block, rel_floor * max|b| overall).

    The global-max normalisation of ``nmax_err`` lets a block whose values sit orders of magnitude below the tensor's
    maximum (a fine hash level, an MLP row, an embedding row) be entirely wrong and still pass; this one scales every block
    by its own magnitude.  ``rel_floor`` keeps blocks that are numerically empty from amplifying summation noise."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    assert a.shape == b.shape, (a.shape, b.shape)
    gmax = float(b.abs().max()) if b.numel() else 0.0
    if gmax == 0.0:
        return float(a.abs().max()) if a.numel() else 0.0
    worst = 0.0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        if hi <= lo:
            continue
        scale = max(float(b[lo:hi].abs().max()), rel_floor * gmax)
        worst = max(worst, float((a[lo:hi] - b[lo:hi]).abs().max()) / scale)
    return worst


def hash_level_bounds(meta, n_features: int = 2):
    """Flat-parameter boundaries of the hash-grid levels (tcnn layout: level after level, F floats per entry)."""
    return [int(o) * n_features for o in meta.offsets]


def row_bounds(n_rows: int, row_len: int):
    return [r * row_len for r in range(n_rows + 1)]


def mlp_row_bounds(mlp):
    """Row boundaries of a tcnn-layout MLP parameter vector: every output neuron of every layer is one block."""
    out, pos = [0], 0
    for (o, i) in mlp.shapes:
        for _ in range(o):
            pos += i
            out.append(pos)
    return out


def grad_errors(name: str, got: torch.Tensor, ref: torch.Tensor, bounds=None) -> Dict[str, float]:
    """The three views of a gradient comparison used throughout the GPU tests."""
    res = {f"d_{name}": nmax_err(got, ref, 1e-12), f"d_{name}_l2": rel_l2(got, ref)}
    if bounds is not None:
        res[f"d_{name}_blk"] = blockwise_nmax_err(got, ref, bounds)
    return res


def random_rays(n: int, seed: int = 0, inside: bool = False, device="cpu"):
    """origins on the radius-1.5 sphere aimed at random targets in [-0.5,0.5]^3
    (``inside``: origins uniformly in [-0.5,0.5]^3, random unit directions)."""
    g = torch.Generator().manual_seed(seed)
    if inside:
        o = torch.rand(n, 3, generator=g) - 0.5
        d = torch.randn(n, 3, generator=g)
    else:
        o = torch.randn(n, 3, generator=g)
        o = 1.5 * o / o.norm(dim=-1, keepdim=True)
        tgt = torch.rand(n, 3, generator=g) - 0.5
        d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    return o.to(device).contiguous(), d.to(device).contiguous()


def random_binaries(levels: int, res: int, frac: float, seed: int = 0) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand((levels, res, res, res), generator=g) < frac


def make_model_pair(grid_levels=4, grid_resolution=128, seed=96, occupied_frac=0.3, num_levels=16, hidden=64,
                    emb_type="global_emb", num_train_data=8, contraction=True, alpha_thre=0.01, cone_angle=0.004,
                    log2_hashmap_size=19, param_scale: float = 1.0, emb_dim: int = 32):
    """(HIP model on cuda:0, ModelOracle on CPU) sharing parameters, occupancy grid and hyper-parameters."""
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, LSEEmbeddingConfig
    from oracle.field import FieldOracle
    from oracle.model import ModelOracle

    torch.manual_seed(seed)
    cfg = LSENeRFModelConfig(grid_levels=grid_levels, grid_resolution=grid_resolution, num_levels=num_levels,
                             hidden_dim=hidden, hidden_dim_color=hidden, alpha_thre=alpha_thre, cone_angle=cone_angle,
                             log2_hashmap_size=log2_hashmap_size, disable_scene_contraction=not contraction,
                             embed_config=LSEEmbeddingConfig(embedding_type=emb_type, emb_dim=emb_dim))
    aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    hip = LSENeRFModel(cfg, aabb, num_train_data)
    if param_scale != 1.0:   # larger table values make the hash contribution visible above fp32 noise
        with torch.no_grad():
            hip.field.mlp_base_grid.params.mul_(param_scale)
    hip = hip.cuda()
    n_emb = hip.field.embedding_appearance.embedding.weight.shape[0]
    f = FieldOracle("tcnn", num_levels=num_levels, hidden_dim=hidden, hidden_dim_color=hidden,
                    log2_hashmap_size=log2_hashmap_size, num_embeddings=n_emb, contraction=contraction, aabb=aabb,
                    seed=seed, appearance_embedding_dim=emb_dim)
    sync_params_to_oracle(hip, f)
    orc = ModelOracle(f, grid_resolution=grid_resolution, grid_levels=grid_levels, alpha_thre=alpha_thre,
                      cone_angle=cone_angle)
    b = random_binaries(grid_levels, grid_resolution, occupied_frac, seed)
    occs = b.float().flatten() * 0.5
    hip.occupancy_grid.binaries.copy_(b.cuda())
    hip.occupancy_grid.occs.copy_(occs.cuda())
    hip.occupancy_grid._occ_mean_host = None
    orc.grid.binaries = b.clone()
    orc.grid.occs = occs.clone()
    assert abs(orc.render_step_size - cfg.render_step_size) < 1e-12
    return hip, orc


def sync_params_to_oracle(hip, field_oracle):
    fld = hip.field
    src = {"grid": fld.mlp_base_grid.params, "base": fld.mlp_base_mlp.params, "head": fld.mlp_head.params,
           "embedding": fld.embedding_appearance.embedding.weight}
    for k, v in src.items():
        assert field_oracle.params[k].shape == v.shape, (k, field_oracle.params[k].shape, v.shape)
        field_oracle.params[k] = v.detach().cpu().clone().requires_grad_(True)
    # the oracle's own level table must agree with the product's (independent restatements of tcnn's constructor)
    m, mo = fld.mlp_base_grid.meta, field_oracle.meta
    assert list(m.offsets) == list(mo.offsets) and list(m.resolutions) == list(mo.resolutions)
    assert np.allclose(np.float32(m.scales), np.float32(mo.scales), rtol=0, atol=0)


def compare_model_outputs(hip, orc, o, d, appearance_id: Optional[torch.Tensor], check_grads=True,
                          jitter: Optional[torch.Tensor] = None) -> Dict[str, float]:
    """Runs hip.exec_get_outputs on the GPU, re-renders THE SAME packed samples through the oracle, compares renders
    and (optionally) every parameter gradient + ray gradients.  Sampler parity is tested separately (bit-exact)."""
    from lsenerf_amd import RayBundle
    hip.train()
    orc.training = True
    R = o.shape[0]
    og = o.clone().cuda().requires_grad_(True)
    dg = d.clone().cuda().requires_grad_(True)
    meta = {}
    if appearance_id is not None:
        meta["appearance_id"] = appearance_id.cuda()
    rb = RayBundle(origins=og, directions=dg, camera_indices=torch.zeros(R, 1, dtype=torch.long, device="cuda"),
                   metadata=meta)
    if jitter is None:
        jitter = torch.rand(R, generator=torch.Generator().manual_seed(7))
    # sampler on the GPU; the same samples go to the oracle
    rs, li = hip.sampler(ray_bundle=rb, near_plane=hip.config.near_plane, far_plane=hip.config.far_plane,
                         render_step_size=hip.config.render_step_size, alpha_thre=hip.config.alpha_thre,
                         cone_angle=hip.config.cone_angle, jitter=jitter.cuda())
    ts, te = rs.frustums.starts[..., 0].contiguous(), rs.frustums.ends[..., 0].contiguous()
    out = hip.render_packed(rb, rs.ray_indices, ts, te, rs.packed_info)

    oc = o.clone().requires_grad_(True)
    dc = d.clone().requires_grad_(True)
    for p in orc.field.parameters():
        p.grad = None
    ref = orc.render_samples(oc, dc, li.cpu(), ts.cpu(), te.cpu(), appearance_id)
    res = {
        "n_samples": float(ts.shape[0]),
        "rgb": nmax_err(out["rgb"], ref["rgb"], 1e-3),
        "acc": nmax_err(out["accumulation"], ref["accumulation"], 1e-3),
        "depth": nmax_err(out["depth"], ref["depth"], 1e-3),
    }
    assert torch.equal(out["num_samples_per_ray"].cpu(), ref["num_samples_per_ray"])
    assert res["rgb"] < TOL_FWD * 5 and res["acc"] < TOL_FWD * 5 and res["depth"] < TOL_FWD * 5, res
    if check_grads:
        g = torch.Generator().manual_seed(3)
        wr = torch.rand(R, 3, generator=g)
        wa = torch.rand(R, 1, generator=g)
        wd = torch.rand(R, 1, generator=g)
        for p in hip.parameters():
            p.grad = None
        loss = (out["rgb"] * wr.cuda()).sum() + (out["accumulation"] * wa.cuda()).sum() + (out["depth"] * wd.cuda()).sum()
        loss.backward()
        lref = (ref["rgb"] * wr).sum() + (ref["accumulation"] * wa).sum() + (ref["depth"] * wd).sum()
        lref.backward()
        fld = hip.field
        pairs = {"grid": fld.mlp_base_grid.params, "base": fld.mlp_base_mlp.params, "head": fld.mlp_head.params,
                 "embedding": fld.embedding_appearance.embedding.weight}
        emb_w = fld.embedding_appearance.embedding.weight
        bounds = {"grid": hash_level_bounds(fld.mlp_base_grid.meta), "base": mlp_row_bounds(fld.mlp_base_mlp),
                  "head": mlp_row_bounds(fld.mlp_head), "embedding": row_bounds(emb_w.shape[0], emb_w.shape[1])}
        for k, p in pairs.items():
            gref = orc.field.params[k].grad
            assert p.grad is not None and gref is not None, k
            # global-max normalised, relative L2, and per hash level / per MLP row / per embedding row
            res.update(grad_errors(k, p.grad, gref, bounds[k]))
            assert res["d_" + k] < TOL_GRAD and res[f"d_{k}_l2"] < TOL_GRAD and res[f"d_{k}_blk"] < TOL_GRAD_BLOCK, (k, res)
        res["d_origins"] = nmax_err(og.grad, oc.grad, 1e-12)
        res["d_directions"] = nmax_err(dg.grad, dc.grad, 1e-12)
        assert res["d_origins"] < TOL_GRAD and res["d_directions"] < TOL_GRAD, res
    return res
