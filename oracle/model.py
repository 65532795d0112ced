"""Oracle: the hot-path driver ``LSENeRFModel.exec_get_outputs`` (R:lse_nerf/lsenerf.py:278-326) and one
training step, CPU torch.  TEST INFRASTRUCTURE (also the ``cpu_baseline`` leg of bench.py).  Parity unpinned.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch

from . import volrend as vr
from .field import FieldOracle, frustum_positions
from .sampling import OccGridOracle


class ModelOracle:
    """Wiring of R:lse_nerf/lsenerf.py:158-228 with the InstantNGPModelConfig defaults of SURVEY.md App. A.9."""

    def __init__(self, field: FieldOracle, grid_resolution=128, grid_levels=4, alpha_thre=0.01, cone_angle=0.004,
                 near_plane=0.05, far_plane=1e3, render_step_size: Optional[float] = None,
                 linear_renderer=False, background="random"):
        self.field = field
        self.scene_aabb = field.aabb.flatten()
        if render_step_size is None:
            render_step_size = ((self.scene_aabb[3:] - self.scene_aabb[:3]) ** 2).sum().sqrt().item() / 1000
        self.render_step_size = render_step_size
        self.grid = OccGridOracle(self.scene_aabb, grid_resolution, grid_levels)
        self.alpha_thre, self.cone_angle = alpha_thre, cone_angle
        self.near_plane, self.far_plane = near_plane, far_plane
        self.linear_renderer = linear_renderer
        self.background = background
        self.training = True

    def sample(self, origins, directions, jitter=None, t_max=None):
        """nerfstudio ``VolumetricSampler.forward`` -> estimator.sampling (sigma_fn only when training)."""
        sigma_fn = None
        if self.training:
            def sigma_fn(t_starts, t_ends, ray_indices):
                pos = origins[ray_indices] + directions[ray_indices] * ((t_starts + t_ends) / 2.0)[:, None]
                return self.field.density_fn(pos).squeeze(-1)
        return self.grid.sampling(origins, directions, sigma_fn=sigma_fn, near_plane=self.near_plane,
                                  far_plane=self.far_plane, t_max=t_max, render_step_size=self.render_step_size,
                                  alpha_thre=self.alpha_thre, stratified=self.training, cone_angle=self.cone_angle,
                                  jitter=jitter)

    def render_samples(self, origins, directions, ray_indices, t_starts, t_ends, appearance_id=None
                       ) -> Dict[str, torch.Tensor]:
        """R:lse_nerf/lsenerf.py:292-326 given the sampler's output."""
        num_rays = origins.shape[0]
        self.field.training = self.training
        pos = frustum_positions(origins[ray_indices], directions[ray_indices], t_starts[:, None], t_ends[:, None])
        density, geo = self.field.get_density(pos)
        if appearance_id is None:   # GlobalEmbedding: index camera_indices * 0 (R:lse_nerf/lse_embeddings.py:80-82)
            aid = torch.zeros(ray_indices.shape[0], dtype=torch.int64)
        else:
            aid = appearance_id[ray_indices]
        rgb = self.field.get_outputs(directions[ray_indices], geo, aid)
        packed_info = vr.pack_info(ray_indices, num_rays)
        weights = vr.render_weight_from_density(t_starts, t_ends, density[..., 0], packed_info)[0][..., None]
        training = True if self.linear_renderer else self.training
        out_rgb = vr.render_rgb(rgb, weights, ray_indices, num_rays, training, self.background)
        depth = vr.render_depth_expected(weights, t_starts, t_ends, ray_indices, num_rays)
        acc = vr.render_accumulation(weights, ray_indices, num_rays)
        return {"rgb": out_rgb, "accumulation": acc, "depth": depth, "num_samples_per_ray": packed_info[:, 1],
                "weights": weights, "density": density, "sample_rgb": rgb}

    def exec_get_outputs(self, origins, directions, appearance_id=None, jitter=None):
        ri, ts, te = self.sample(origins, directions, jitter)
        if ri.numel() == 0:  # VolumetricSampler's fake sample
            ri = torch.zeros(1, dtype=torch.int64)
            ts = torch.ones(1)
            te = torch.ones(1)
        return self.render_samples(origins, directions, ri, ts, te, appearance_id)


def adam_step(params, state, lr=1e-2, eps=1e-15, betas=(0.9, 0.999)):
    """torch.optim.Adam semantics (R:lse_nerf/lse_config.py:31), written out for flat-buffer parity tests."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    for i, p in enumerate(params):
        if p.grad is None:
            continue
        m = state.setdefault(("m", i), torch.zeros_like(p))
        v = state.setdefault(("v", i), torch.zeros_like(p))
        m.mul_(betas[0]).add_(p.grad, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(p.grad, p.grad, value=1 - betas[1])
        bc1 = 1 - betas[0] ** t
        bc2 = 1 - betas[1] ** t
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.data.addcdiv_(m, denom, value=-lr / bc1)


def cpu_train_step_packed(model: ModelOracle, origins, directions, ray_indices, t_starts, t_ends, target,
                          appearance_id, opt_state, ray_chunk=256):
    """One training step on pre-packed samples (metric workload), processed in ray chunks with gradient
    accumulation so the torch path's 8 x [N,L,F] temporaries fit."""
    R = origins.shape[0]
    params = model.field.parameters()
    for p in params:
        p.grad = None
    cnt = torch.bincount(ray_indices, minlength=R)
    starts = torch.cumsum(cnt, 0) - cnt
    total = 0.0
    for r0 in range(0, R, ray_chunk):
        r1 = min(R, r0 + ray_chunk)
        s0, s1 = int(starts[r0]), int(starts[r1 - 1] + cnt[r1 - 1])
        out = model.render_samples(origins[r0:r1], directions[r0:r1], ray_indices[s0:s1] - r0, t_starts[s0:s1],
                                   t_ends[s0:s1], None if appearance_id is None else appearance_id[r0:r1])
        loss = ((out["rgb"] - target[r0:r1]) ** 2).sum() / (R * 3)
        loss.backward()
        total += float(loss)
    adam_step(params, opt_state)
    return total
