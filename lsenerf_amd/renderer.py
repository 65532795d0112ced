"""Renderers with the reference's interface, running on the packed volume-rendering kernel.

  * ``RGBRenderer`` / ``AccumulationRenderer`` / ``DepthRenderer("expected")``: nerfstudio 0.3.2 signatures as
    called at R:lse_nerf/lsenerf.py:309-318;
  * ``LinearRenderer``: R:lse_nerf/lse_renderer.py:4-10 (forces the training branch so linear radiance is neither
    nan_to_num'ed nor clamped to [0,1]).

The standalone renderers composite with ``ops.volume_render`` fed with *weights*; the model's fast path
(``lsenerf_amd.model``) calls ``ops.volume_render`` once for weights + all three composites.
"""
from __future__ import annotations

from typing import Optional, Union

import torch
from torch import Tensor, nn

from . import ops


class _WeightedSumFn(torch.autograd.Function):
    """out[r] = sum_{i in ray r} w_i * v_i over packed, ray-sorted samples (nerfacc.accumulate_along_rays)."""

    @staticmethod
    def forward(ctx, weights, values, packed_info):
        wv = weights[:, None] * values if values is not None else weights[:, None]
        wv = wv.contiguous()
        R = packed_info.shape[0]
        out = torch.zeros((R, wv.shape[1]), dtype=torch.float32, device=wv.device)
        from . import _lib
        import ctypes
        assert wv.shape[1] <= 64
        _lib.call("lse_segment_sum_rows", ctypes.c_void_p(wv.data_ptr()), wv.shape[1],
                  ctypes.c_void_p(packed_info.data_ptr()), R, ctypes.c_void_p(out.data_ptr()), ops._stream())
        ctx.save_for_backward(weights, values, packed_info)
        return out

    @staticmethod
    def backward(ctx, g):
        weights, values, packed_info = ctx.saved_tensors
        cnt = packed_info[:, 1]
        g_per = torch.repeat_interleave(g, cnt, dim=0)
        if values is None:
            return g_per[:, 0], None, None
        return (g_per * values).sum(-1), g_per * weights[:, None], None


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor], ray_indices: Tensor, n_rays: int,
                          packed_info: Optional[Tensor] = None) -> Tensor:
    """weights [N] (or [N,1]); values [N,C] or None; samples must be ray-sorted (they are: the sampler packs them)."""
    w = weights.reshape(-1)
    if packed_info is None:
        cnt = torch.bincount(ray_indices.long(), minlength=n_rays)
        packed_info = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], dim=-1).contiguous()
    return _WeightedSumFn.apply(w, values, packed_info)


class RGBRenderer(nn.Module):
    def __init__(self, background_color: Union[str, Tensor] = "random") -> None:
        super().__init__()
        self.background_color = background_color

    def combine_rgb(self, rgb, weights, ray_indices, num_rays, packed_info=None):
        comp = accumulate_along_rays(weights[..., 0], rgb, ray_indices, num_rays, packed_info)
        bg = self.background_color
        if isinstance(bg, str) and bg in ("random", "last_sample"):
            return comp          # "random": returned un-blended
        acc = accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays, packed_info)
        if isinstance(bg, str):
            bg = {"black": 0.0, "white": 1.0}[bg]
        return comp + bg * (1.0 - acc)

    def forward(self, rgb: Tensor, weights: Tensor, ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None,
                packed_info: Optional[Tensor] = None) -> Tensor:
        if not self.training:
            rgb = torch.nan_to_num(rgb)
        out = self.combine_rgb(rgb, weights, ray_indices, num_rays, packed_info)
        if not self.training:
            torch.clamp_(out, min=0.0, max=1.0)
        return out


class LinearRenderer(RGBRenderer):
    """R:lse_nerf/lse_renderer.py:4-10."""

    def forward(self, rgb, weights, ray_indices, num_rays, packed_info=None) -> Tensor:
        tmp = self.training
        self.training = True
        out = super().forward(rgb, weights, ray_indices, num_rays, packed_info)
        self.training = tmp
        return out


class AccumulationRenderer(nn.Module):
    def forward(self, weights: Tensor, ray_indices: Optional[Tensor] = None, num_rays: Optional[int] = None,
                packed_info: Optional[Tensor] = None) -> Tensor:
        return accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays, packed_info)


class DepthRenderer(nn.Module):
    def __init__(self, method: str = "expected") -> None:
        super().__init__()
        assert method == "expected", "only the method the reference uses (R:lse_nerf/lsenerf.py:199)"
        self.method = method

    def forward(self, weights: Tensor, ray_samples, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None, packed_info: Optional[Tensor] = None) -> Tensor:
        eps = 1e-10
        steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
        depth = accumulate_along_rays(weights[..., 0], steps, ray_indices, num_rays, packed_info)
        acc = accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays, packed_info)
        depth = depth / (acc + eps)
        return torch.clip(depth, steps.min(), steps.max())


def finish_depth(depth_num: Tensor, acc: Tensor, t_starts: Tensor, t_ends: Tensor) -> Tensor:
    """DepthRenderer("expected") epilogue on the fused kernel's outputs: num/(acc+1e-10), clipped to the step range."""
    depth = depth_num / (acc + 1e-10)
    if t_starts.numel() > 0:
        steps = (t_starts + t_ends) / 2
        depth = torch.clip(depth, steps.min(), steps.max())
    return depth
