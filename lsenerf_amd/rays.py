"""Minimal duck-typed stand-ins for nerfstudio's ``RayBundle`` / ``Frustums`` / ``RaySamples``
(nerfstudio 0.3.2 ``cameras/rays.py``; not importable here).  Only the attributes the hot path touches
(R:lse_nerf/lsenerf.py:278-326, R:lse_nerf/lse_field.py:264-360) are modelled, with the same names and shapes."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Union

import torch
from torch import Tensor


class SceneContraction(torch.nn.Module):
    """nerfstudio ``SceneContraction`` (field_components/spatial_distortions.py) as the reference builds it at
    R:lse_nerf/lsenerf.py:163-166: ``x`` if ``|x| < 1`` else ``(2 - 1/|x|) x/|x|`` with ``|.|`` the given norm order.
    ``LSEField`` only reads ``.order`` (the contraction itself is fused into ``lse_positions_fwd``); calling the module
    evaluates the same map with torch ops for callers that use it directly."""

    def __init__(self, order: Optional[Union[float, int]] = None) -> None:
        super().__init__()
        self.order = order

    def forward(self, positions: Tensor) -> Tensor:
        mag = torch.linalg.norm(positions, ord=self.order, dim=-1)[..., None]
        return torch.where(mag < 1, positions, (2 - (1 / mag)) * (positions / mag))


@dataclass
class SceneBox:
    """nerfstudio ``SceneBox``: ``aabb`` [2,3] (min corner, max corner)."""
    aabb: Tensor


@dataclass
class RayBundle:
    origins: Tensor                       # [R,3]
    directions: Tensor                    # [R,3] unit
    pixel_area: Optional[Tensor] = None   # [R,1]
    camera_indices: Optional[Tensor] = None  # [R,1] int
    nears: Optional[Tensor] = None        # [R,1]
    fars: Optional[Tensor] = None         # [R,1]
    times: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)

    def __len__(self) -> int:
        return self.origins.shape[0]

    @staticmethod
    def cat(bundles, alias_blocks: bool = False) -> "RayBundle":
        """One bundle holding the rays of ``bundles`` in order (autograd flows back to every part).  An optional field must be
        present in all parts or in none; metadata keys likewise.  The result owns fresh tensors (``torch.cat``) -- in-place work on
        it never reaches the parts.  ``alias_blocks=True`` is the opt-in of callers whose parts are THEIR OWN consecutive row blocks
        of one buffer per field (lsenerf_amd.graph's static inputs): the join is then that buffer itself, no copy, no autograd node
        -- and an alias of the parts, so such callers must not write to the joined bundle in place."""
        bundles = list(bundles)
        if len(bundles) == 1:
            return bundles[0]

        def join_vals(vals):
            base = _whole_base(vals) if alias_blocks else None
            return base if base is not None else torch.cat(vals, dim=0)

        def join(name):
            vals = [getattr(b, name) for b in bundles]
            if all(v is None for v in vals):
                return None
            if any(v is None for v in vals):
                raise ValueError(f"RayBundle.cat: '{name}' is set on some bundles only")
            return join_vals(vals)
        keys = set(bundles[0].metadata)
        if any(set(b.metadata) != keys for b in bundles):
            raise ValueError("RayBundle.cat: metadata keys differ between bundles")
        meta = {k: join_vals([b.metadata[k] for b in bundles]) for k in keys}
        return RayBundle(origins=join("origins"), directions=join("directions"), pixel_area=join("pixel_area"),
                         camera_indices=join("camera_indices"), nears=join("nears"), fars=join("fars"), times=join("times"),
                         metadata=meta)


def _whole_base(vals):
    """If ``vals`` are consecutive dim-0 slices of ONE contiguous tensor that together cover it, that tensor; else None.
    (lsenerf_amd.graph keeps the static inputs of a captured step that way: the three bundles of a step are row blocks of one buffer
    per field, so joining them costs no kernel -- and, for rays that carry gradients, the joined tensor is the leaf itself.)"""
    base = vals[0]._base
    if base is None or not base.is_contiguous() or any(v._base is not base for v in vals):
        return None
    expect = base.storage_offset()
    for v in vals:
        if v.dim() != base.dim() or v.shape[1:] != base.shape[1:] or v.dtype != base.dtype or not v.is_contiguous() \
                or v.storage_offset() != expect:
            return None
        expect += v.numel()
    return base if expect == base.storage_offset() + base.numel() else None


@dataclass
class Frustums:
    origins: Tensor      # [N,3]
    directions: Tensor   # [N,3]
    starts: Tensor       # [N,1]
    ends: Tensor         # [N,1]
    pixel_area: Optional[Tensor] = None

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def get_positions(self) -> Tensor:
        """nerfstudio: origins + directions * (starts + ends) / 2."""
        return self.origins + self.directions * (self.starts + self.ends) / 2


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[Tensor] = None   # [N,1]
    times: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    # packed-sample bookkeeping the HIP path adds (absent upstream; optional for callers)
    ray_indices: Optional[Tensor] = None      # [N] int32, sorted
    packed_info: Optional[Tensor] = None      # [R,2] int64
    ray_bundle: Optional[RayBundle] = None

    def __len__(self) -> int:
        return self.frustums.origins.shape[0]
