"""Oracle: multiresolution hash-grid encodings (torch, autograd-capable).  TEST INFRASTRUCTURE.

Two layouts, selected by the caller exactly as ``Ed_HashEncoding`` does (R:lse_nerf/lse_field.py:43-91):

* ``tcnn``  -- tiny-cuda-nn 1.7 ``HashGrid`` (what ``implementation="tcnn"`` builds at
  R:lse_nerf/lse_field.py:72-86).  Restated from the published ``grid.h``.
  This is the GPU parity target of the HIP kernels.
* ``torch`` -- nerfstudio 0.3.2 ``HashEncoding.pytorch_fwd`` (the fallback when tcnn is missing,
  R:lse_nerf/lse_field.py:67-69).  This is the CPU baseline field.

Parity unpinned: neither upstream package is available here (see oracle/__init__.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np
import torch

PRIME_Y = 2654435761
PRIME_Z = 805459861


# --------------------------------------------------------------------------------------------
# tcnn layout
# --------------------------------------------------------------------------------------------
@dataclass
class TcnnGridMeta:
    """Host-side level table of a tcnn HashGrid (offsets in *entries* of ``n_features`` floats)."""

    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float
    scales: List[float]        # float32 values, grid_scale(level)
    resolutions: List[int]     # grid_resolution(scale)
    offsets: List[int]         # len n_levels+1, in entries
    @property
    def n_entries(self) -> int:
        return self.offsets[-1]

    @property
    def n_params(self) -> int:
        return self.offsets[-1] * self.n_features

    def level_size(self, l: int) -> int:
        return self.offsets[l + 1] - self.offsets[l]

    def is_dense(self, l: int) -> bool:
        # grid_index(): hash iff hashmap_size < stride after the 3-dim stride loop
        r = self.resolutions[l]
        size = self.level_size(l)
        stride = 1
        for _ in range(3):
            if stride > size:
                break
            stride *= r
        return not (size < stride)


def growth_factor(min_res: int, max_res: int, num_levels: int) -> float:
    """R:lse_nerf/lse_field.py:59 (np.exp of float64 logs)."""
    if num_levels <= 1:
        return 1.0
    return float(np.exp((np.log(max_res) - np.log(min_res)) / (num_levels - 1)))


def tcnn_grid_meta(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16,
                   per_level_scale=None, max_res=2048) -> TcnnGridMeta:
    """tcnn ``GridEncodingTemplated`` constructor: offset table + per-level scale/resolution.

    grid_scale(l)      = exp2f(l * log2f(per_level_scale)) * base_resolution - 1      (float32)
    grid_resolution(s) = (uint32)ceilf(s) + 1
    params_in_level    = min(next_multiple(res^3, 8), 2^log2_hashmap_size)            (Hash grids)
    """
    if per_level_scale is None:
        per_level_scale = growth_factor(base_resolution, max_res, n_levels)
    pls = np.float32(per_level_scale)
    log2_pls = np.log2(pls, dtype=np.float32)
    scales, ress, offsets = [], [], [0]
    for l in range(n_levels):
        s = np.float32(np.exp2(np.float32(l) * log2_pls, dtype=np.float32) * np.float32(base_resolution)
                       - np.float32(1.0))
        r = int(np.ceil(s)) + 1
        dense = r ** 3
        max_params = (2 ** 32 - 1) // 2
        n = max_params if float(r) ** 3 > float(max_params) else dense
        n = ((n + 7) // 8) * 8
        n = min(n, 1 << log2_hashmap_size)
        scales.append(float(s))
        ress.append(r)
        offsets.append(offsets[-1] + n)
    return TcnnGridMeta(n_levels, n_features, log2_hashmap_size, base_resolution, float(pls),
                        scales, ress, offsets)


def _tcnn_index(px, py, pz, res: int, size: int, dense: bool):
    """grid_index<3, CoherentPrime>() on int64 tensors holding uint32 values."""
    M = 0xFFFFFFFF
    if dense:
        # index accumulates in uint32; with 3 dims and stride<=size guard all three dims are added
        idx = (px + py * res + pz * (res * res)) & M
    else:
        idx = (px ^ ((py * PRIME_Y) & M) ^ ((pz * PRIME_Z) & M)) & M
    return idx % size


def hash_encode_tcnn(x: torch.Tensor, table: torch.Tensor, meta: TcnnGridMeta) -> torch.Tensor:
    """tcnn ``kernel_grid`` forward, Linear interpolation.  x [N,3] in [0,1]; table flat [n_params]
    (or [n_entries, F]); returns [N, L*F] (level-major columns, as the torch binding returns it).

    pos = fmaf(scale, x, 0.5); p0 = floor(pos); w = pos - p0;
    y_l = sum_{c=0..7} prod_d (c_d ? w_d : 1-w_d) * table[off_l + index(p0 + c)]   (c bit d <-> dim d)
    """
    F = meta.n_features
    tab = table.reshape(-1, F)
    outs = []
    for l in range(meta.n_levels):
        scale = meta.scales[l]
        res = meta.resolutions[l]
        off = meta.offsets[l]
        size = meta.level_size(l)
        dense = meta.is_dense(l)
        # fmaf emulation: the f32*f32 product is exact in f64; one extra f64 rounding on the add is
        # far below the stated tolerance.  Gradient d pos / d x = scale flows through autograd.
        pos = (x.double() * float(np.float32(scale)) + 0.5).to(x.dtype)
        p0 = torch.floor(pos)
        w = pos - p0
        p0i = p0.detach().to(torch.int64) & 0xFFFFFFFF
        acc = torch.zeros(x.shape[0], F, dtype=x.dtype, device=x.device)
        for c in range(8):
            wgt = torch.ones(x.shape[0], dtype=x.dtype, device=x.device)
            cs = []
            for d in range(3):
                if (c >> d) & 1:
                    wgt = wgt * w[:, d]
                    cs.append((p0i[:, d] + 1) & 0xFFFFFFFF)
                else:
                    wgt = wgt * (1 - w[:, d])
                    cs.append(p0i[:, d])
            idx = _tcnn_index(cs[0], cs[1], cs[2], res, size, dense) + off
            acc = acc + wgt[:, None] * tab[idx]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


def tcnn_corner_indices(x: torch.Tensor, meta: TcnnGridMeta, level: int) -> torch.Tensor:
    """[N,8] absolute entry indices touched at ``level`` (bit-exact integer side of the encode)."""
    scale = meta.scales[level]
    pos = (x.double() * float(np.float32(scale)) + 0.5).to(torch.float32)
    p0i = torch.floor(pos).to(torch.int64) & 0xFFFFFFFF
    out = []
    for c in range(8):
        cs = [((p0i[:, d] + ((c >> d) & 1)) & 0xFFFFFFFF) for d in range(3)]
        out.append(_tcnn_index(cs[0], cs[1], cs[2], meta.resolutions[level], meta.level_size(level),
                               meta.is_dense(level)) + meta.offsets[level])
    return torch.stack(out, dim=-1)


def init_tcnn_table(meta: TcnnGridMeta, generator: torch.Generator | None = None) -> torch.Tensor:
    """tcnn grid init: U(-1e-4, 1e-4), one flat float32 tensor."""
    return (torch.rand(meta.n_params, generator=generator) * 2 - 1) * 1e-4


# --------------------------------------------------------------------------------------------
# nerfstudio 0.3.2 torch layout  
# --------------------------------------------------------------------------------------------
@dataclass
class TorchGridMeta:
    num_levels: int
    features_per_level: int
    log2_hashmap_size: int
    scalings: torch.Tensor     # [L] float32, floor(min_res * g^l)   R:lse_nerf/lse_field.py:58-60
    hash_offset: torch.Tensor  # [L] int64, l * 2^T                  R:lse_nerf/lse_field.py:62

    @property
    def hash_table_size(self) -> int:
        return 1 << self.log2_hashmap_size


def torch_grid_meta(num_levels=16, min_res=16, max_res=2048, log2_hashmap_size=19,
                    features_per_level=2) -> TorchGridMeta:
    levels = torch.arange(num_levels)
    g = growth_factor(min_res, max_res, num_levels)
    scalings = torch.floor(min_res * g ** levels)
    return TorchGridMeta(num_levels, features_per_level, log2_hashmap_size, scalings.float(),
                         levels * (1 << log2_hashmap_size))


def init_torch_table(meta: TorchGridMeta, hash_init_scale=1e-3,
                     generator: torch.Generator | None = None) -> torch.Tensor:
    """R:lse_nerf/lse_field.py:63-65."""
    t = torch.rand((meta.hash_table_size * meta.num_levels, meta.features_per_level), generator=generator)
    return (t * 2 - 1) * hash_init_scale


def _ns_hash(v: torch.Tensor, meta: TorchGridMeta) -> torch.Tensor:
    """nerfstudio ``HashEncoding.hash_fn``: v [...,L,3] int32 -> [...,L] (int64 after promotion)."""
    v = v.to(torch.int64)
    x = v[..., 0] * 1
    y = v[..., 1] * PRIME_Y
    z = v[..., 2] * PRIME_Z
    h = torch.bitwise_xor(torch.bitwise_xor(x, y), z)
    h = h % meta.hash_table_size
    return h + meta.hash_offset.to(h.device)


def hash_encode_torch(x: torch.Tensor, table: torch.Tensor, meta: TorchGridMeta) -> torch.Tensor:
    """nerfstudio 0.3.2 ``HashEncoding.pytorch_fwd`` (every level hashed, no +0.5, no dense levels)."""
    xs = x[..., None, :]
    scaled = xs * meta.scalings.view(-1, 1).to(x.device)
    sc = torch.ceil(scaled).type(torch.int32)
    sf = torch.floor(scaled).type(torch.int32)
    offset = scaled - sf

    h0 = _ns_hash(sc, meta)
    h1 = _ns_hash(torch.cat([sc[..., 0:1], sf[..., 1:2], sc[..., 2:3]], dim=-1), meta)
    h2 = _ns_hash(torch.cat([sf[..., 0:1], sf[..., 1:2], sc[..., 2:3]], dim=-1), meta)
    h3 = _ns_hash(torch.cat([sf[..., 0:1], sc[..., 1:2], sc[..., 2:3]], dim=-1), meta)
    h4 = _ns_hash(torch.cat([sc[..., 0:1], sc[..., 1:2], sf[..., 2:3]], dim=-1), meta)
    h5 = _ns_hash(torch.cat([sc[..., 0:1], sf[..., 1:2], sf[..., 2:3]], dim=-1), meta)
    h6 = _ns_hash(sf, meta)
    h7 = _ns_hash(torch.cat([sf[..., 0:1], sc[..., 1:2], sf[..., 2:3]], dim=-1), meta)

    f0, f1, f2, f3 = table[h0], table[h1], table[h2], table[h3]
    f4, f5, f6, f7 = table[h4], table[h5], table[h6], table[h7]
    ox, oy, oz = offset[..., 0:1], offset[..., 1:2], offset[..., 2:3]
    f03 = f0 * ox + f3 * (1 - ox)
    f12 = f1 * ox + f2 * (1 - ox)
    f56 = f5 * ox + f6 * (1 - ox)
    f47 = f4 * ox + f7 * (1 - ox)
    f0312 = f03 * oy + f12 * (1 - oy)
    f4756 = f47 * oy + f56 * (1 - oy)
    enc = f0312 * oz + f4756 * (1 - oz)
    return torch.flatten(enc, start_dim=-2, end_dim=-1)
