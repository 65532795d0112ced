"""Oracle: packed volume rendering + compositing (nerfacc 0.5.2 / nerfstudio 0.3.2 restated).
TEST INFRASTRUCTURE.  Parity unpinned.  SURVEY.md App. A.6 / A.8; call sites R:lse_nerf/lsenerf.py:300-318,
R:lse_nerf/lse_renderer.py:4-10, R:lse_nerf/lse_grid_estimator.py:120-127.
All functions are plain differentiable torch so backward goldens come from autograd.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def pack_info(ray_indices: torch.Tensor, n_rays: int) -> torch.Tensor:
    """nerfacc.pack_info -> [n_rays, 2] int64 (start, count)."""
    cnt = torch.zeros(n_rays, dtype=torch.int64)
    cnt.index_add_(0, ray_indices, torch.ones_like(ray_indices))
    start = torch.cumsum(cnt, 0) - cnt
    return torch.stack([start, cnt], dim=-1)


def exclusive_sum(x: torch.Tensor, packed_info: torch.Tensor) -> torch.Tensor:
    """Segmented exclusive prefix sum (differentiable: built from cumsum + per-ray offsets)."""
    if x.numel() == 0:
        return x.clone()
    starts, cnts = packed_info[:, 0], packed_info[:, 1]
    cs = torch.cumsum(x, 0)
    excl = cs - x
    # subtract the running total at each ray start
    nz = cnts > 0
    base_per_ray = torch.zeros(packed_info.shape[0], dtype=x.dtype)
    base_per_ray[nz] = excl[starts[nz]]
    ray_of = torch.repeat_interleave(torch.arange(packed_info.shape[0]), cnts)
    return excl - base_per_ray[ray_of]


def exclusive_sum_seq(x: torch.Tensor, packed_info: torch.Tensor) -> torch.Tensor:
    """Strictly sequential per-ray fp32 accumulation (the order a serial scan produces)."""
    out = torch.zeros_like(x)
    xs = x.tolist()
    import numpy as np
    for s, c in packed_info.tolist():
        acc = np.float32(0)
        for i in range(s, s + c):
            out[i] = float(acc)
            acc = np.float32(acc + np.float32(xs[i]))
    return out


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info):
    sigmas_dt = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sigmas_dt)
    trans = torch.exp(-exclusive_sum(sigmas_dt, packed_info))
    return trans, alphas


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info) -> Tuple[torch.Tensor, ...]:
    """nerfacc.render_weight_from_density -> (weights, transmittance, alphas).  R:lse_nerf/lsenerf.py:301-306."""
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info)
    return trans * alphas, trans, alphas


@torch.no_grad()
def render_visibility_from_density(t_starts, t_ends, sigmas, packed_info, early_stop_eps=1e-4, alpha_thre=0.0):
    """nerfacc.render_visibility_from_density.  R:lse_nerf/lse_grid_estimator.py:120-127."""
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis


def accumulate_along_rays(weights: torch.Tensor, values: Optional[torch.Tensor], ray_indices: torch.Tensor,
                          n_rays: int) -> torch.Tensor:
    """nerfacc.accumulate_along_rays: zeros(n_rays, C).index_add_(0, ray_indices, w[:,None]*values)."""
    src = weights[..., None] if values is None else weights[..., None] * values
    out = torch.zeros((n_rays, src.shape[-1]), dtype=src.dtype)
    return out.index_add(0, ray_indices, src)


def render_rgb(rgb, weights, ray_indices, num_rays, training=True, background="random"):
    """nerfstudio ``RGBRenderer.forward`` with packed samples; ``LinearRenderer`` == training=True always
    (R:lse_nerf/lse_renderer.py:6-10).  background "random": returned un-blended (App. A.8)."""
    if not training:
        rgb = torch.nan_to_num(rgb)
    comp = accumulate_along_rays(weights[..., 0], rgb, ray_indices, num_rays)
    if background not in ("random", "last_sample"):
        acc = accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays)
        bg = {"black": 0.0, "white": 1.0}[background]
        comp = comp + bg * (1.0 - acc)
    if not training:
        comp = torch.clamp(comp, 0.0, 1.0)
    return comp


def render_accumulation(weights, ray_indices, num_rays):
    return accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays)


def render_depth_expected(weights, t_starts, t_ends, ray_indices, num_rays):
    """nerfstudio ``DepthRenderer("expected")``: sum w*(s+e)/2 / (sum w + 1e-10), clipped to [min,max] of steps."""
    eps = 1e-10
    steps = (t_starts + t_ends) / 2
    depth = accumulate_along_rays(weights[..., 0], steps[..., None], ray_indices, num_rays)
    acc = accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays)
    depth = depth / (acc + eps)
    if steps.numel() > 0:
        depth = torch.clip(depth, steps.min(), steps.max())
    return depth
