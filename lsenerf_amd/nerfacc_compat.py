"""The nerfacc 0.5.2 functions the reference calls by name on the hot path, on the gfx950 kernels.

``import lsenerf_amd.nerfacc_compat as nerfacc`` lets R:lse_nerf/lsenerf.py:300-306 and
R:lse_nerf/lse_grid_estimator.py:120-138 run unchanged:

  * ``pack_info(ray_indices, n_rays)``                               R:lse_nerf/lsenerf.py:300
  * ``render_weight_from_density(t_starts, t_ends, sigmas, ...)``    R:lse_nerf/lsenerf.py:301-306
  * ``render_visibility_from_density`` / ``render_visibility_from_alpha``   R:lse_nerf/lse_grid_estimator.py:120-138
  * ``accumulate_along_rays``                                        (what nerfstudio's renderers call)

Samples must be ray-sorted (they are: the sampler emits them packed).  This is the generic composition route; the
model's fast path (``LSENeRFModel.render_packed``) fuses weights + all three renderers into one kernel.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .renderer import accumulate_along_rays  # noqa: F401  (re-export, nerfacc.accumulate_along_rays)


def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """``[n_rays, 2]`` int64 (start, count) of every ray's contiguous sample segment."""
    assert ray_indices.dim() == 1, "ray_indices must be a 1D tensor"
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1 if ray_indices.numel() else 0
    cnt = torch.bincount(ray_indices.long(), minlength=n_rays)
    if not cnt.is_cuda:
        return torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], dim=-1)
    return ops.pack_info_from_counts(cnt.contiguous())[0]


def _packed(packed_info, ray_indices, n_rays) -> Tensor:
    if packed_info is None:
        assert ray_indices is not None, "either packed_info or ray_indices (+ n_rays) is needed"
        packed_info = pack_info(ray_indices, n_rays)
    return packed_info.contiguous()


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Optional[Tensor] = None,
                               ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                               prefix_trans: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """``(weights, transmittance, alphas)``: ``w_k = T_k (1 - exp(-sigma_k dt_k))``, ``T_k = exp(-sum_{i<k} sigma_i dt_i)``."""
    assert prefix_trans is None, "prefix_trans is not used by the reference and not implemented"
    return ops.render_weight_from_density(t_starts.reshape(-1), t_ends.reshape(-1), sigmas.reshape(-1),
                                          _packed(packed_info, ray_indices, n_rays))


@torch.no_grad()
def _visibility(values: Tensor, t_starts, t_ends, packed_info, early_stop_eps, alpha_thre, from_alpha) -> Tensor:
    n = values.shape[0]
    dev = values.device
    if n == 0:
        return torch.zeros(0, dtype=torch.bool, device=dev)
    dummy_ri = torch.zeros(n, dtype=torch.int32, device=dev)
    z = values if from_alpha else None
    _, _, _, _, mask = ops.visibility_compact(dummy_ri, (z if from_alpha else t_starts).contiguous(),
                                              (z if from_alpha else t_ends).contiguous(), values.contiguous(), packed_info,
                                              early_stop_eps, alpha_thre, from_alpha=from_alpha)
    return mask.bool()


def render_visibility_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Optional[Tensor] = None,
                                   ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None,
                                   early_stop_eps: float = 1e-4, alpha_thre: float = 0.0) -> Tensor:
    """Boolean mask ``T >= early_stop_eps`` (and ``alpha >= alpha_thre`` when ``alpha_thre > 0``)."""
    return _visibility(sigmas.reshape(-1), t_starts.reshape(-1), t_ends.reshape(-1), _packed(packed_info, ray_indices, n_rays),
                       early_stop_eps, alpha_thre, False)


def render_visibility_from_alpha(alphas: Tensor, packed_info: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                                 n_rays: Optional[int] = None, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0) -> Tensor:
    return _visibility(alphas.reshape(-1), None, None, _packed(packed_info, ray_indices, n_rays), early_stop_eps,
                       alpha_thre, True)
