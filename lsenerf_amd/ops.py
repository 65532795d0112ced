"""Torch-facing custom ops over the C-ABI (include/lse_hip.h): tensor checks, workspace allocation, autograd.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every op below launches the
hand-written gfx950 kernels on torch's current HIP stream.  No CPU implementation exists on this path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import ctypes
import time

import torch

from . import _lib
from ._lib import GridDesc, MlpDesc


# ----------------------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------------------
def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: Optional[torch.Tensor], dtype, name: str, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError(f"{name} is None")
    if not t.is_cuda:
        raise _lib.LseHipError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def _f32(t, name, allow_none=False):
    return _chk(t, torch.float32, name, allow_none)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


_N_DEV_AT = {"lse_positions_fwd": 6, "lse_positions_bwd": 6, "lse_hash_bwd_ex": 10}     # elsewhere: right before the stream


def _call_n(name: str, n_dev: Optional[torch.Tensor], *args):
    """``_lib.call`` of a per-sample entry point: inserts the ``n_dev`` argument (include/lse_hip.h: device-side sample count,
    right behind ``n``).  With a tensor (int64, on the device) the ``n`` among ``args`` is a CAPACITY and the kernels clamp it to
    ``n_dev[0]``; ``None``: plain call."""
    ptr = None
    if n_dev is not None:
        assert n_dev.dtype == torch.int64 and n_dev.is_cuda and n_dev.numel() >= 1
        ptr = ctypes.c_void_p(n_dev.data_ptr())
    at = _N_DEV_AT.get(name, len(args) - 1)
    return _lib.call(name, *args[:at], ptr, *args[at:])


# ----------------------------------------------------------------------------------------------------
# descriptors
# ----------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class GridMeta:
    """tcnn HashGrid level table (host side).  Same arithmetic as tcnn's GridEncoding constructor; the oracle's
    ``tcnn_grid_meta`` is an independent restatement used by the tests to cross-check these numbers."""
    n_levels: int
    n_features: int
    log2_hashmap_size: int
    base_resolution: int
    per_level_scale: float
    scales: Tuple[float, ...]
    resolutions: Tuple[int, ...]
    offsets: Tuple[int, ...]
    # overrides of lse_hash_bwd_opts for this grid's backward, as ((field, value), ...): the caller's knowledge of the sample
    # regime (HASH_BWD_DENSE_STEPS / HASH_BWD_DEFAULT below); call arguments of lse_hash_bwd_ex, nothing the library remembers
    bwd_tuning: Tuple[Tuple[str, int], ...] = ()

    @property
    def n_entries(self):
        return self.offsets[-1]

    @property
    def n_params(self):
        return self.offsets[-1] * self.n_features

    def desc(self) -> GridDesc:
        d = GridDesc()
        d.n_levels, d.n_features = self.n_levels, self.n_features
        for i, o in enumerate(self.offsets):
            d.offsets[i] = o
        for i in range(self.n_levels):
            d.scales[i] = self.scales[i]
            d.resolutions[i] = self.resolutions[i]
        return d


def make_grid_meta(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16, per_level_scale=None,
                   max_res=2048) -> GridMeta:
    import numpy as np
    if per_level_scale is None:   # R:lse_nerf/lse_field.py:59
        per_level_scale = float(np.exp((np.log(max_res) - np.log(base_resolution)) / (n_levels - 1))) if n_levels > 1 else 1.0
    pls = np.float32(per_level_scale)
    l2 = np.log2(pls, dtype=np.float32)
    scales, ress, offs = [], [], [0]
    for l in range(n_levels):
        s = np.float32(np.exp2(np.float32(l) * l2, dtype=np.float32) * np.float32(base_resolution) - np.float32(1))
        r = int(np.ceil(s)) + 1
        n = min(r ** 3, (2 ** 32 - 1) // 2)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << log2_hashmap_size)
        scales.append(float(s)); ress.append(r); offs.append(offs[-1] + n)
    return GridMeta(n_levels, n_features, log2_hashmap_size, base_resolution, float(pls), tuple(scales), tuple(ress),
                    tuple(offs))


@dataclass(frozen=True)
class MlpMeta:
    n_in: int                 # padded input width seen by the kernel (8/16/32/64)
    width: int
    n_hidden_layers: int
    out_activation: int = _lib.LSE_ACT_NONE
    in_layout: int = _lib.LSE_IN_ROWMAJOR
    w0_ld: int = 0            # first-layer view into params (include/lse_hip.h: lse_mlp_desc); 0 = plain layout
    w0_col: int = 0
    w0_mask_col0: int = 0
    arith: int = _lib.LSE_MLP_ARITH_AUTO      # forward arithmetic route where two are built (include/lse_hip.h: lse_mlp_desc.arith)

    @property
    def n_params(self):
        return self.width * (self.w0_ld or self.n_in) + (self.n_hidden_layers - 1) * self.width * self.width + 16 * self.width

    def desc(self) -> MlpDesc:
        return MlpDesc(self.n_in, self.width, self.n_hidden_layers, self.out_activation, self.in_layout, self.w0_ld,
                       self.w0_col, self.w0_mask_col0, self.arith)


# ----------------------------------------------------------------------------------------------------
# sampler
# ----------------------------------------------------------------------------------------------------
MAX_SLOT_ELEMS = 1 << 28   # single-pass marcher: upper limit for R * cap (2 x 1 GiB of scratch)
SYNC_STATS = {"seconds": 0.0, "count": 0}   # host time spent blocked in the sampler's count read-backs (bench.py reports it)


@torch.no_grad()
def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size: float, cone_angle: float,
                   max_span: float = None, fma_setup: bool = False):
    """nerfacc.grid.traverse_grids as consumed at R:lse_nerf/lse_grid_estimator.py:93-106.
    Returns (ray_indices int32 [N], t_starts [N], t_ends [N], packed_info int64 [R,2]).  One host sync (N).

    max_span: host-known upper bound of (t_exit - t_enter) over all rays (clipped by the planes and the outermost aabb).
    With it the march runs ONCE into fixed-capacity per-ray slots (every sample interval is >= step_size long, so
    cap = max_span / step_size + slack bounds the count) and a copy kernel packs them; without it (or when the slots
    would be too large) the published count pass + write pass run.
    fma_setup: LSE_TRAVERSE_FMA_SETUP -- nvcc's default contraction at the four a*b+c sites of the traversal set-up."""
    flags = _lib.LSE_TRAVERSE_FMA_SETUP if fma_setup else 0
    R = rays_o.shape[0]
    L, rx, ry, rz = binaries.shape
    dev = rays_o.device
    cnts = torch.empty(R, dtype=torch.int64, device=dev)
    packed = torch.empty((R, 2), dtype=torch.int64, device=dev)
    total = torch.zeros(2, dtype=torch.int64, device=dev)   # [N, overflow flag of the single-pass marcher]
    args = (_f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), R, _chk(binaries, torch.uint8, "binaries"),
            _f32(aabbs, "aabbs"), L, rx, ry, rz, _f32(near_planes, "near_planes"), _f32(far_planes, "far_planes"),
            float(step_size), float(cone_angle))
    if SINGLE_PASS_MARCH and max_span is not None and step_size > 0 and R > 0:
        cap = int(max_span / step_size) + 8
        if 0 < cap and R * cap <= MAX_SLOT_ELEMS:
            ts_slots = torch.empty(R * cap, dtype=torch.float32, device=dev)
            te_slots = torch.empty(R * cap, dtype=torch.float32, device=dev)
            flag = total[1:].view(torch.int32)    # low word of total[1]
            _lib.call("lse_traverse_grids_slots", *args, cap, ctypes.c_void_p(cnts.data_ptr()),
                      ctypes.c_void_p(ts_slots.data_ptr()), ctypes.c_void_p(te_slots.data_ptr()),
                      ctypes.c_void_p(flag.data_ptr()), flags, _stream())
            _lib.call("lse_pack_info_from_counts", ctypes.c_void_p(cnts.data_ptr()), R, ctypes.c_void_p(packed.data_ptr()),
                      ctypes.c_void_p(total.data_ptr()), _stream())
            _t0 = time.perf_counter()
            n, overflow = (int(v) for v in total.tolist())        # <- host sync #1: the marcher's sample count
            SYNC_STATS["seconds"] += time.perf_counter() - _t0
            SYNC_STATS["count"] += 1
            if not overflow:
                ri = torch.empty(n, dtype=torch.int32, device=dev)
                ts = torch.empty(n, dtype=torch.float32, device=dev)
                te = torch.empty(n, dtype=torch.float32, device=dev)
                if n > 0:
                    _lib.call("lse_compact_ray_slots", ctypes.c_void_p(ts_slots.data_ptr()), ctypes.c_void_p(te_slots.data_ptr()),
                              cap, ctypes.c_void_p(packed.data_ptr()), R, ctypes.c_void_p(ri.data_ptr()),
                              ctypes.c_void_p(ts.data_ptr()), ctypes.c_void_p(te.data_ptr()), _stream())
                return ri, ts, te, packed
            total.zero_()   # bound violated (not expected): redo with the two-pass scheme
    _lib.call("lse_traverse_grids", *args, 0, ctypes.c_void_p(cnts.data_ptr()), None, None, None, None, flags, _stream())
    _lib.call("lse_pack_info_from_counts", ctypes.c_void_p(cnts.data_ptr()), R, ctypes.c_void_p(packed.data_ptr()),
              ctypes.c_void_p(total.data_ptr()), _stream())
    n = int(total[0].item())
    ri = torch.empty(n, dtype=torch.int32, device=dev)
    ts = torch.empty(n, dtype=torch.float32, device=dev)
    te = torch.empty(n, dtype=torch.float32, device=dev)
    if n > 0:
        starts = packed[:, 0].contiguous()
        _lib.call("lse_traverse_grids", *args, 1, None, ctypes.c_void_p(starts.data_ptr()),
                  ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(ts.data_ptr()), ctypes.c_void_p(te.data_ptr()),
                  flags, _stream())
    return ri, ts, te, packed


@torch.no_grad()
def traverse_grids_deferred(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size: float, cone_angle: float,
                            cap: int, out=None, overflow=None, fma_setup: bool = False):
    """``traverse_grids`` without the host read-back of the sample count.  ``cap`` is a PROVEN upper bound of the samples of
    one ray (LSEOccGridEstimator._cap_per_ray); the packed outputs have room for ``R * cap`` samples and the actual count stays
    on the device.  Returns (ray_indices int32 [C], t_starts [C], t_ends [C], packed_info int64 [R,2], n_dev int64 [1],
    overflow int32 [1]) with C = R * cap; entries at and beyond n_dev[0] are never written nor read by the kernels that are
    handed ``n_dev``.
    ``overflow``: int32 [1] on the device that the marcher ORs into and NEVER clears -- a sticky accumulator the caller owns
    (LSEOccGridEstimator keeps one for all its calls, captured ones included, and reads it at the occupancy refresh); a fresh
    zero is allocated when absent.  Bit 0: a ray exceeded ``cap`` (its samples were truncated to ``cap``: nothing is read or
    written out of bounds); bit 1: such a ray's direction was shorter than 1.
    ``out``: a previous result of this function for the same R and cap, written into instead of allocating (a marcher that runs
    ahead of its step on a side stream fills buffers at addresses the consumer already knows: lsenerf_amd.graph)."""
    R = rays_o.shape[0]
    L, rx, ry, rz = binaries.shape
    dev = rays_o.device
    C = R * cap
    if not (0 < cap and C <= MAX_SLOT_ELEMS):
        raise _lib.LseHipError(f"deferred sampling: {R} rays x {cap} slots exceed the slot budget ({MAX_SLOT_ELEMS})")
    cnts = torch.empty(R, dtype=torch.int64, device=dev)
    if out is not None:
        ri, ts, te, packed, total, flag_o = out
        if not (ri.shape == ts.shape == te.shape == (C,) and packed.shape == (R, 2) and total.shape == (1,)):
            raise _lib.LseHipError("traverse_grids_deferred: `out` is not a result of this function for the same rays x capacity")
        if overflow is None:
            overflow = flag_o
    else:
        packed = torch.empty((R, 2), dtype=torch.int64, device=dev)
        total = torch.empty(1, dtype=torch.int64, device=dev)       # written (not accumulated) by lse_pack_info_from_counts
    if overflow is None:
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    flag = _chk(overflow, torch.int32, "overflow")
    ts_slots = torch.empty(C, dtype=torch.float32, device=dev)
    te_slots = torch.empty(C, dtype=torch.float32, device=dev)
    _lib.call("lse_traverse_grids_slots", _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), R, _chk(binaries, torch.uint8, "binaries"),
              _f32(aabbs, "aabbs"), L, rx, ry, rz, _f32(near_planes, "near_planes"), _f32(far_planes, "far_planes"),
              float(step_size), float(cone_angle), cap, ctypes.c_void_p(cnts.data_ptr()), ctypes.c_void_p(ts_slots.data_ptr()),
              ctypes.c_void_p(te_slots.data_ptr()), flag, _lib.LSE_TRAVERSE_FMA_SETUP if fma_setup else 0, _stream())
    _lib.call("lse_pack_info_from_counts", ctypes.c_void_p(cnts.data_ptr()), R, ctypes.c_void_p(packed.data_ptr()),
              ctypes.c_void_p(total.data_ptr()), _stream())
    if out is None:
        ri = torch.empty(C, dtype=torch.int32, device=dev)
        ts = torch.empty(C, dtype=torch.float32, device=dev)
        te = torch.empty(C, dtype=torch.float32, device=dev)
    _lib.call("lse_compact_ray_slots", ctypes.c_void_p(ts_slots.data_ptr()), ctypes.c_void_p(te_slots.data_ptr()), cap,
              ctypes.c_void_p(packed.data_ptr()), R, ctypes.c_void_p(ri.data_ptr()), ctypes.c_void_p(ts.data_ptr()),
              ctypes.c_void_p(te.data_ptr()), _stream())
    return ri, ts, te, packed, total, overflow


@torch.no_grad()
def fake_sample_if_empty(packed_info, n_dev, ray_indices, t_starts, t_ends, features=None):
    """nerfstudio's single fake sample (ray 0, t = 1) when the device-side count is 0 (lse_fake_sample_if_empty); in place.
    ``features = (x01 [C,3], selector [C], y [L,C,F])``: the parked pre-pass features of these very buffers -- slot 0 is zeroed
    together with the insertion (no survivor means nothing was compacted into it)."""
    fx = fs = fy = None
    stride, L, F = 0, 0, 0
    if features is not None:
        x01, sel, y = features
        fx, fs, fy = _f32(x01, "features.x01"), _chk(sel, torch.uint8, "features.selector"), _f32(y, "features.y")
        L, stride, F = y.shape[0], y.shape[1] * y.shape[2], y.shape[2]
    _lib.call("lse_fake_sample_if_empty", _chk(packed_info, torch.int64, "packed_info"), packed_info.shape[0],
              _chk(n_dev, torch.int64, "n_dev"), _chk(ray_indices, torch.int32, "ray_indices"), _f32(t_starts, "t_starts"),
              _f32(t_ends, "t_ends"), fx, fs, fy, stride, L, F, _stream())


@torch.no_grad()
def visibility_compact_deferred(ray_indices, t_starts, t_ends, sigmas, packed_info, early_stop_eps: float, alpha_thre: float,
                                from_alpha: bool = False, alpha_cap: Optional[torch.Tensor] = None):
    """``visibility_compact`` with the survivors' count left on the device: outputs keep the inputs' capacity.
    ``alpha_cap``: float32 [1] on the device; the threshold used is min(alpha_thre, alpha_cap[0]) (lse_visibility_mask_cap).
    Returns (ray_indices, t_starts, t_ends, packed_info, mask, n_dev)."""
    R = packed_info.shape[0]
    C = t_starts.shape[0]
    dev = t_starts.device
    mask = torch.empty(C, dtype=torch.uint8, device=dev)
    new_cnts = torch.empty(R, dtype=torch.int64, device=dev)
    if from_alpha:
        _lib.call("lse_visibility_mask_alpha", _f32(sigmas, "alphas"), _chk(packed_info, torch.int64, "packed_info"), R,
                  float(early_stop_eps), float(alpha_thre), ctypes.c_void_p(mask.data_ptr()),
                  ctypes.c_void_p(new_cnts.data_ptr()), _stream())
    elif alpha_cap is not None:
        _lib.call("lse_visibility_mask_cap", _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas"),
                  _chk(packed_info, torch.int64, "packed_info"), R, float(early_stop_eps), float(alpha_thre),
                  _f32(alpha_cap, "alpha_cap"), ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(new_cnts.data_ptr()), _stream())
    else:
        _lib.call("lse_visibility_mask", _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas"),
                  _chk(packed_info, torch.int64, "packed_info"), R, float(early_stop_eps), float(alpha_thre),
                  ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(new_cnts.data_ptr()), _stream())
    new_packed, total = pack_info_from_counts(new_cnts)
    o_ri = torch.empty(C, dtype=torch.int32, device=dev)
    o_ts = torch.empty(C, dtype=torch.float32, device=dev)
    o_te = torch.empty(C, dtype=torch.float32, device=dev)
    _lib.call("lse_compact_samples", ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(packed_info.data_ptr()),
              ctypes.c_void_p(new_packed.data_ptr()), R, _chk(ray_indices, torch.int32, "ray_indices"),
              _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), ctypes.c_void_p(o_ri.data_ptr()),
              ctypes.c_void_p(o_ts.data_ptr()), ctypes.c_void_p(o_te.data_ptr()), _stream())
    return o_ri, o_ts, o_te, new_packed, mask, total[0:1]


@torch.no_grad()
def ray_planes(n_rays: int, device, near_plane: float, far_plane: float, t_min=None, t_max=None, jitter=None,
               step_size: float = 0.0):
    """(near_planes[R], far_planes[R]) of R:lse_nerf/lse_grid_estimator.py:83-92 in one launch."""
    near = torch.empty(n_rays, dtype=torch.float32, device=device)
    far = torch.empty(n_rays, dtype=torch.float32, device=device)
    keep = [x for x in (t_min, t_max, jitter) if x is not None]      # (contiguous views stay alive through `keep`)
    keep = [_c(x.reshape(-1).float()) for x in keep]
    it = iter(keep)
    ptrs = [(_f32(next(it), n) if x is not None else None) for x, n in ((t_min, "t_min"), (t_max, "t_max"), (jitter, "jitter"))]
    _lib.call("lse_ray_planes", float(near_plane), float(far_plane), ptrs[0], ptrs[1], ptrs[2], float(step_size), n_rays,
              ctypes.c_void_p(near.data_ptr()), ctypes.c_void_p(far.data_ptr()), _stream())
    return near, far


@torch.no_grad()
def pack_info_from_counts(cnts: torch.Tensor):
    R = cnts.shape[0]
    packed = torch.empty((R, 2), dtype=torch.int64, device=cnts.device)
    total = torch.empty(1, dtype=torch.int64, device=cnts.device)       # written (not accumulated) by the kernel, also for R == 0
    _lib.call("lse_pack_info_from_counts", _chk(cnts, torch.int64, "cnts"), R, ctypes.c_void_p(packed.data_ptr()),
              ctypes.c_void_p(total.data_ptr()), _stream())
    return packed, total


@torch.no_grad()
def visibility_compact(ray_indices, t_starts, t_ends, sigmas, packed_info, early_stop_eps: float, alpha_thre: float,
                       from_alpha: bool = False):
    """render_visibility_from_density (or, ``from_alpha``, render_visibility_from_alpha with ``sigmas`` holding opacities)
    + mask compaction (R:lse_nerf/lse_grid_estimator.py:120-143)."""
    R = packed_info.shape[0]
    n = t_starts.shape[0]
    dev = t_starts.device
    if n == 0:   # nothing to cull (every ray missed every grid)
        return ray_indices, t_starts, t_ends, packed_info, torch.empty(0, dtype=torch.uint8, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    new_cnts = torch.empty(R, dtype=torch.int64, device=dev)
    if from_alpha:
        _lib.call("lse_visibility_mask_alpha", _f32(sigmas, "alphas"), _chk(packed_info, torch.int64, "packed_info"), R,
                  float(early_stop_eps), float(alpha_thre), ctypes.c_void_p(mask.data_ptr()),
                  ctypes.c_void_p(new_cnts.data_ptr()), _stream())
    else:
        _lib.call("lse_visibility_mask", _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas"),
                  _chk(packed_info, torch.int64, "packed_info"), R, float(early_stop_eps), float(alpha_thre),
                  ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(new_cnts.data_ptr()), _stream())
    new_packed, total = pack_info_from_counts(new_cnts)
    _t0 = time.perf_counter()
    m = int(total.item())                                         # <- host sync #2: the survivors of the visibility cull
    SYNC_STATS["seconds"] += time.perf_counter() - _t0
    SYNC_STATS["count"] += 1
    o_ri = torch.empty(m, dtype=torch.int32, device=dev)
    o_ts = torch.empty(m, dtype=torch.float32, device=dev)
    o_te = torch.empty(m, dtype=torch.float32, device=dev)
    if m > 0:
        _lib.call("lse_compact_samples", ctypes.c_void_p(mask.data_ptr()), ctypes.c_void_p(packed_info.data_ptr()),
                  ctypes.c_void_p(new_packed.data_ptr()), R, _chk(ray_indices, torch.int32, "ray_indices"),
                  _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), ctypes.c_void_p(o_ri.data_ptr()),
                  ctypes.c_void_p(o_ts.data_ptr()), ctypes.c_void_p(o_te.data_ptr()), _stream())
    return o_ri, o_ts, o_te, new_packed, mask


# ----------------------------------------------------------------------------------------------------
# positions (frustum mid-point -> contraction -> [0,1] -> selector)
# ----------------------------------------------------------------------------------------------------
class _PositionsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, contraction: bool, aabb6, pre=None, n_dev=None):
        direct = ray_idx is None
        n = rays_o.shape[0] if direct else ray_idx.shape[0]
        if pre is not None:     # (x01, selector) already computed for exactly these samples by the visibility pre-pass
            x01, sel = pre
            assert x01.shape == (n, 3) and sel.shape == (n,)
            x01 = x01.detach().view_as(x01)     # fresh tensor objects: outputs of this node
            sel = sel.view_as(sel)
        else:
            x01 = torch.empty((n, 3), dtype=torch.float32, device=rays_o.device)
            sel = torch.empty(n, dtype=torch.uint8, device=rays_o.device)
            aabb_arr = (ctypes.c_float * 6)(*aabb6) if aabb6 is not None else None
            _call_n("lse_positions_fwd", n_dev, _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d", True),
                      _chk(ray_idx, torch.int32, "ray_idx", True), _f32(t_starts, "t_starts", True),
                      _f32(t_ends, "t_ends", True), n, int(contraction), aabb_arr, ctypes.c_void_p(x01.data_ptr()),
                      ctypes.c_void_p(sel.data_ptr()), _stream())
        ctx.save_for_backward(rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info)
        ctx.contraction, ctx.aabb6, ctx.n, ctx.n_dev = contraction, aabb6, n, n_dev
        ctx.mark_non_differentiable(sel)
        return x01, sel

    @staticmethod
    def backward(ctx, d_x01, _d_sel):
        rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info = ctx.saved_tensors
        n = ctx.n
        d_x01 = _c(d_x01)
        d_pos = torch.empty((n, 3), dtype=torch.float32, device=d_x01.device)
        aabb_arr = (ctypes.c_float * 6)(*ctx.aabb6) if ctx.aabb6 is not None else None
        _call_n("lse_positions_bwd", ctx.n_dev, _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d", True),
                  _chk(ray_idx, torch.int32, "ray_idx", True), _f32(t_starts, "t_starts", True),
                  _f32(t_ends, "t_ends", True), n, int(ctx.contraction), aabb_arr, _f32(d_x01, "d_x01"),
                  ctypes.c_void_p(d_pos.data_ptr()), _stream())
        if ray_idx is None:
            return d_pos, None, None, None, None, None, None, None, None, None
        if packed_info is None:
            raise _lib.LseHipError("positions backward w.r.t. rays needs packed_info")
        R = rays_o.shape[0]
        d_o = torch.empty_like(rays_o) if ctx.needs_input_grad[0] else None
        d_d = torch.empty_like(rays_d) if ctx.needs_input_grad[1] else None
        if d_o is not None or d_d is not None:
            _lib.call("lse_ray_grad_reduce", ctypes.c_void_p(d_pos.data_ptr()), _f32(t_starts, "t_starts"),
                      _f32(t_ends, "t_ends"), _chk(packed_info, torch.int64, "packed_info"), R,
                      _f32(d_o, "d_o", True), _f32(d_d, "d_d", True), _stream())
        return d_o, d_d, None, None, None, None, None, None, None, None


def positions(rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, contraction=True, aabb6=None, precomputed=None,
              n_dev=None):
    """Sample positions in the field's unit cube + in-bounds selector (R:lse_nerf/lse_field.py:266-274).
    ``ray_idx is None``: ``rays_o`` holds positions directly (Field.density_fn).  ``precomputed = (x01, selector)``: values
    the visibility pre-pass already produced for exactly these samples (only the autograd node is created)."""
    return _PositionsFn.apply(rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, contraction, aabb6, precomputed, n_dev)


# ----------------------------------------------------------------------------------------------------
# hash grid
# ----------------------------------------------------------------------------------------------------
def _direct_grad(p: torch.Tensor):
    """Leaf parameter whose .grad is already allocated (optim.FlatParams keeps every .grad as a view of one flat, zeroed
    buffer): the backward kernels accumulate straight into it -- the entry points accumulate anyway -- and the Function
    returns None for that input, which saves a zero-fill plus an accumulate pass per parameter and step."""
    if DIRECT_PARAM_GRADS and p.is_leaf and p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32 \
            and p.grad.is_contiguous() and p.grad.shape == p.shape:
        return p.grad
    return None


class _DirectGradFn(torch.autograd.Function):
    """Identity on a leaf parameter whose gradient buffer exists (``_direct_grad``): the backward adds the incoming gradient into
    that buffer and hands NOTHING on, so the parameter's AccumulateGrad node never receives a gradient.  For parameters that plain
    torch ops consume (the torch route of the loss: MLP intensity mappers, ``ThreeToOne``, ``Powpow``) -- the kernels of the fast
    path write their parameter gradients in place already."""

    @staticmethod
    def forward(ctx, p):
        ctx.param = p
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        direct = _direct_grad(ctx.param)
        if direct is None:
            return g
        direct.add_(g)
        return None


def direct_grad_params(module: torch.nn.Module) -> Optional[dict]:
    """``{name: tensor}`` for ``torch.func.functional_call(module, ...)``: every parameter of ``module`` that has a preallocated
    gradient buffer behind ``_DirectGradFn``.  None when there is nothing to redirect (no such parameter, or no autograd)."""
    if not torch.is_grad_enabled():
        return None
    out = {n: _DirectGradFn.apply(p) for n, p in module.named_parameters() if _direct_grad(p) is not None}
    return out or None


# Path-selection thresholds of the hash backward by sample regime (profiles/r05_hash_bwd_thresholds.txt, MI355X): a wave that ends
# <= few_runs runs at a level adds them straight to memory; a level that ends > stage_max runs passes the sector cache unstaged.
#   constant step (cone_angle == 0: consecutive samples walk through cells at a fixed pace, the run-end count grows smoothly from
#   level to level and the queue fills well): (3, 56) -- re-tuned on the round-5 kernel (leaner cache pass, one wave per workgroup,
#   whole rays per XCD: a cache pass got cheaper than the direct adds it replaces), profiles/r05_hash_bwd_thresholds_final.txt:
#   against (6, 32) headline 2.53 -> 2.46 ms, inside-box 2.19 -> 2.14, M-packed 3.03 -> 2.96; (2, 64) is 0.6 % better at the
#   headline and 2 % worse on M-packed, and few_runs 1 falls off a cliff (2.69);
#   steps that grow with the distance (cone_angle > 0, the reference's default: sparser samples, more run ends per level):
#   (8, 48) = the library's defaults -- default configuration 1.51 -> 1.43 ms (still the best pair on the final kernel).
HASH_BWD_DENSE_STEPS = (("few_runs", 3), ("stage_max", 56))
HASH_BWD_DEFAULT = ()


_HASH_BWD_WS = {}      # (device index, stream) -> zeroed workspace of the hash backward's coarse-level replicas


def hash_bwd_opts_with_workspace(desc: GridDesc, device) -> "_lib.HashBwdOpts":
    """Default lse_hash_bwd_opts plus the zero-initialised replica workspace the call needs (lse_hash_bwd_workspace_bytes): one
    buffer per (device, stream), allocated once -- the call returns it zeroed, so it is never cleared from here."""
    o = _lib.hash_bwd_default_opts()
    nbytes = int(_lib.load().lse_hash_bwd_workspace_bytes(ctypes.byref(desc), ctypes.byref(o)))
    if nbytes > 0:
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        ws = _HASH_BWD_WS.get(key)
        if ws is None or ws.numel() * 4 < nbytes:
            ws = _HASH_BWD_WS[key] = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)
        o.workspace, o.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    return o


class _HashFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x01, table, meta: GridMeta, pre_y=None, n_dev=None):
        n = x01.shape[0]
        if pre_y is not None:      # features of exactly these positions, encoded by the visibility pre-pass with this table
            assert pre_y.shape == (meta.n_levels, n, meta.n_features)
            y = pre_y.detach().view_as(pre_y)
        else:
            y = torch.empty((meta.n_levels, n, meta.n_features), dtype=torch.float32, device=x01.device)
            desc = meta.desc()
            _call_n("lse_hash_fwd", n_dev, ctypes.byref(desc), _f32(x01, "x01"), _f32(table, "table"),
                    ctypes.c_void_p(y.data_ptr()), n, _stream())
        ctx.save_for_backward(x01, table)
        ctx.meta, ctx.n_dev = meta, n_dev
        return y

    @staticmethod
    def backward(ctx, dy):
        x01, table = ctx.saved_tensors
        meta = ctx.meta
        n = x01.shape[0]
        dy = _c(dy)
        direct = _direct_grad(table)
        dtable = direct if direct is not None else torch.zeros_like(table)
        dx = torch.empty_like(x01) if ctx.needs_input_grad[0] else None
        desc = meta.desc()
        opts = hash_bwd_opts_with_workspace(desc, x01.device)     # defaults + the coarse-level replica workspace
        for k, v in meta.bwd_tuning:                              # the caller's regime hint (call arguments, not library state)
            setattr(opts, k, v)
        hooks = _hash_bwd_hooks_of(table)      # per TABLE, not per process: other models in the process are untouched
        split = hooks.get("split") if hooks else None
        if hooks and hooks.get("before") is not None:     # e.g. fork a side stream here: the scatter below is the last big kernel of the backward pass
            hooks["before"]()
        if split is not None and direct is not None and 0 < split[0] < meta.n_levels:
            # fine levels first (most of the table bytes); their gradients are final when the callback runs, so the caller
            # can start exchanging them while the coarse levels are still being computed (dist.OverlappedGradExchange)
            for lo, hi, acc in ((split[0], meta.n_levels, 0), (0, split[0], 1)):
                _call_n("lse_hash_bwd_ex", ctx.n_dev, ctypes.byref(desc), _f32(x01, "x01"), _f32(dy, "dy"), _f32(table, "table"),
                        ctypes.c_void_p(dtable.data_ptr()), _f32(dx, "dx", True), acc, lo, hi, n, ctypes.byref(opts), _stream())
                if acc == 0:
                    split[1]()
        else:
            _call_n("lse_hash_bwd_ex", ctx.n_dev, ctypes.byref(desc), _f32(x01, "x01"), _f32(dy, "dy"), _f32(table, "table"),
                    ctypes.c_void_p(dtable.data_ptr()), _f32(dx, "dx", True), 0, 0, meta.n_levels, n, ctypes.byref(opts),
                    _stream())
        return dx, (None if direct is not None else dtable), None, None, None


def hash_encode(x01: torch.Tensor, table: torch.Tensor, meta: GridMeta, precomputed: Optional[torch.Tensor] = None,
                n_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tcnn HashGrid forward.  Returns level-major features y[L, N, F] (the fused MLP consumes this directly);
    ``y.permute(1, 0, 2).reshape(N, L*F)`` is the [N, L*F] tensor tcnn's torch binding returns.
    ``precomputed``: y already encoded for exactly these positions with this table (only the autograd node is created)."""
    return _HashFn.apply(x01, table, meta, precomputed, n_dev)


@torch.no_grad()
def compact_features(mask, packed_info, new_packed_info, n_new: int, x01, sel, y):
    """(x01, selector, y) of the ``n_new`` samples that survive ``mask`` (lse_compact_features); y is level-major [L, N, 2]."""
    R = packed_info.shape[0]
    n_old, L = x01.shape[0], y.shape[0]
    dev = x01.device
    o_x = torch.empty((n_new, 3), dtype=torch.float32, device=dev)
    o_s = torch.empty(n_new, dtype=torch.uint8, device=dev)
    o_y = torch.empty((L, n_new, 2), dtype=torch.float32, device=dev)
    _lib.call("lse_compact_features", _chk(mask, torch.uint8, "mask"), _chk(packed_info, torch.int64, "packed_info"),
              _chk(new_packed_info, torch.int64, "new_packed_info"), R, _f32(x01, "x01"), _chk(sel, torch.uint8, "selector"),
              _f32(y, "y"), L, n_old, n_new, ctypes.c_void_p(o_x.data_ptr()), ctypes.c_void_p(o_s.data_ptr()),
              ctypes.c_void_p(o_y.data_ptr()), _stream())
    return o_x, o_s, o_y


# ----------------------------------------------------------------------------------------------------
# fused MLP
# ----------------------------------------------------------------------------------------------------
class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, x, row_bias, row_bias_idx, bias_packed_info, selector, meta: MlpMeta, n: int, out_cols: int,
                density_scale, n_dev=None):
        dev = params.device
        out = torch.empty((n, out_cols), dtype=torch.float32, device=dev)
        sigma = torch.empty(n, dtype=torch.float32, device=dev) if density_scale is not None else None
        need_grad = any(t is not None and t.requires_grad for t in (params, x, row_bias))
        tiled = 1 if (FUSED_WGRAD and ACT_TILED) else 0
        # two hidden layers on a row-major input (the head): the first hidden layer is not saved, the backward recomputes it
        if tiled and RECOMPUTE_FIRST_LAYER and meta.n_hidden_layers == 2 and meta.in_layout == _lib.LSE_IN_ROWMAJOR \
                and meta.n_in % 16 == 0 and need_grad and params.requires_grad \
                and (row_bias is None or not row_bias.requires_grad
                     or (FUSED_BIAS_GRAD and row_bias_idx is not None and bias_packed_info is not None)):
            tiled = 2     # (only the second-generation backward recomputes: it needs the bias gradient reduced in-kernel)
        # third generation (mlp_x6.h): nothing is saved, the backward recomputes every hidden layer on the bf16 matrix cores
        if tiled and RECOMPUTE_ALL and meta.width == 64 and need_grad and params.requires_grad \
                and ((meta.n_in == 16 and meta.n_hidden_layers == 2 and meta.in_layout == _lib.LSE_IN_ROWMAJOR)
                     or (meta.n_in == 32 and meta.n_hidden_layers == 1)) \
                and (row_bias is None or not row_bias.requires_grad
                     or (FUSED_BIAS_GRAD and row_bias_idx is not None and bias_packed_info is not None)):
            tiled = 3
        n_act = (n + 15) // 16 * 16 if tiled else n
        n_saved = meta.n_hidden_layers - (1 if tiled == 2 else 0)
        act = torch.empty((n_saved, n_act, meta.width), dtype=torch.float32, device=dev) if (need_grad and tiled != 3) else None
        ctx.act_tiled = tiled
        desc = meta.desc()
        if n_dev is not None and need_grad and tiled != 3:
            raise _lib.LseHipError("a device-side sample count needs the recomputing MLP kernels (width 64, head / base shape)")
        _call_n("lse_mlp_fwd", n_dev, ctypes.byref(desc), _f32(params, "params"), _f32(x, "mlp input"),
                  _f32(row_bias, "row_bias", True), _chk(row_bias_idx, torch.int32, "row_bias_idx", True),
                  ctypes.c_void_p(out.data_ptr()), out_cols, _f32(act, "act", True), tiled, _f32(sigma, "sigma", True),
                  _chk(selector, torch.uint8, "selector", True), float(density_scale or 0.0), n, _stream())
        ctx.save_for_backward(params, x, act, out, row_bias, row_bias_idx, bias_packed_info, selector)
        ctx.meta, ctx.n, ctx.out_cols, ctx.density_scale, ctx.n_dev = meta, n, out_cols, density_scale, n_dev
        ctx.set_materialize_grads(False)
        return out if sigma is None else (out, sigma)

    @staticmethod
    def backward(ctx, d_out, d_sigma=None):
        params, x, act, out, row_bias, row_bias_idx, bias_packed_info, selector = ctx.saved_tensors
        meta, n, out_cols = ctx.meta, ctx.n, ctx.out_cols
        dev = params.device
        d_out = _c(d_out) if d_out is not None else torch.zeros((n, out_cols), dtype=torch.float32, device=dev)
        d_sigma = _c(d_sigma) if d_sigma is not None else None
        need_bias = row_bias is not None and ctx.needs_input_grad[2]
        # sorted row indices (packed samples): the kernel reduces the bias gradient per row itself
        fused_bias = FUSED_BIAS_GRAD and need_bias and row_bias_idx is not None and bias_packed_info is not None
        d_bias = torch.zeros_like(row_bias) if fused_bias else None
        d_act0 = torch.empty((n, meta.width), dtype=torch.float32, device=dev) if (need_bias and not fused_bias) else None
        d_in = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        direct = _direct_grad(params) if (ctx.needs_input_grad[0] and (ctx.act_tiled or FUSED_WGRAD)) else None
        d_params = direct if direct is not None else (torch.zeros_like(params) if ctx.needs_input_grad[0] else None)
        desc = meta.desc()
        scale = float(ctx.density_scale or 0.0)
        sel = _chk(selector, torch.uint8, "selector", True)
        if ctx.act_tiled or FUSED_WGRAD or d_params is None:
            _call_n("lse_mlp_bwd", ctx.n_dev, ctypes.byref(desc), _f32(params, "params"), _f32(x, "mlp input"),
                      _f32(act, "act", ctx.act_tiled == 3),
                      ctx.act_tiled, _f32(out, "out"), out_cols, _f32(d_out, "d_out"), _f32(d_sigma, "d_sigma", True), sel, scale,
                      None, None, _f32(d_act0, "d_act0", True), _f32(d_in, "d_in", True),
                      _f32(d_params, "d_params", True), _f32(row_bias, "row_bias", True),
                      _chk(row_bias_idx, torch.int32, "row_bias_idx", True) if (fused_bias or ctx.act_tiled >= 2) else None,
                      _f32(d_bias if fused_bias else None, "d_bias", True), n, _stream())
        else:   # reference structure: materialise d_act, then one G^T A reduction per layer (padded outputs only)
            assert out_cols == 16
            d_out_pre = torch.empty((n, 16), dtype=torch.float32, device=dev)
            d_act = torch.empty((meta.n_hidden_layers, n, meta.width), dtype=torch.float32, device=dev)
            _lib.call("lse_mlp_bwd", ctypes.byref(desc), _f32(params, "params"), _f32(x, "mlp input"), _f32(act, "act"),
                      0, _f32(out, "out"), out_cols, _f32(d_out, "d_out"), _f32(d_sigma, "d_sigma", True), sel, scale,
                      ctypes.c_void_p(d_out_pre.data_ptr()), ctypes.c_void_p(d_act.data_ptr()), None,
                      _f32(d_in, "d_in", True), None, None, None, None, n, None, _stream())
            _lib.call("lse_mlp_wgrad", ctypes.byref(desc), _f32(x, "mlp input"), _f32(act, "act"),
                      ctypes.c_void_p(d_act.data_ptr()), ctypes.c_void_p(d_out_pre.data_ptr()),
                      ctypes.c_void_p(d_params.data_ptr()), n, _stream())
            d_act0 = d_act[0] if need_bias else None
            fused_bias, d_bias = False, None
        if need_bias and not fused_bias:
            if row_bias_idx is None:
                d_bias = d_act0
            elif bias_packed_info is not None:
                d_bias = torch.zeros_like(row_bias)
                _lib.call("lse_segment_sum_rows", ctypes.c_void_p(d_act0.data_ptr()), meta.width,
                          _chk(bias_packed_info, torch.int64, "bias_packed_info"), row_bias.shape[0],
                          ctypes.c_void_p(d_bias.data_ptr()), _stream())
            else:   # unsorted row indices: generic scatter-add (not on the hot path)
                d_bias = torch.zeros_like(row_bias).index_add_(0, row_bias_idx.long(), d_act0)
        return (None if direct is not None else d_params), d_in, d_bias, None, None, None, None, None, None, None, None


# Hooks into the hash backward, keyed by the TABLE they belong to (its storage address): a second model in the same process -- an
# evaluation copy, the viewer thread nerfstudio runs beside training -- never sees the hooks of the model being trained.  (Until round
# 3 these were two process-global variables.)
#   "before": callable run (in the autograd thread, on the backward's stream) right before the hash backward is launched
#   "split":  (level, callback) of dist.OverlappedGradExchange: two launches, callback in between
_HASH_BWD_HOOKS: dict = {}


def _hash_bwd_hooks_of(table: torch.Tensor):
    """The hooks registered for the table stored at ``table``'s address, or None.  An entry whose owner (the tensor object the hooks
    were installed through) has been collected is dropped: its address may belong to somebody else's table by now."""
    h = _HASH_BWD_HOOKS.get(table.data_ptr())
    if h is not None and h["owner"]() is None:
        _HASH_BWD_HOOKS.pop(table.data_ptr(), None)
        return None
    return h


def set_hash_bwd_hook(table: torch.Tensor, kind: str, value) -> None:
    """Install (``value`` not None) or remove a hook of the hash backward of ``table`` (``kind``: "before" | "split").  Keyed by
    the table's storage address as it is NOW: install after optim.FlatParams has re-pointed the parameter into the flat buffer."""
    import weakref
    assert kind in ("before", "split")
    key = table.data_ptr()
    h = _hash_bwd_hooks_of(table)
    if value is None:
        if h is not None:
            h.pop(kind, None)
            if set(h) == {"owner"}:
                _HASH_BWD_HOOKS.pop(key, None)
        return
    if h is None:
        h = _HASH_BWD_HOOKS[key] = {"owner": weakref.ref(table)}
    h[kind] = value


def get_hash_bwd_hook(table: torch.Tensor, kind: str):
    h = _hash_bwd_hooks_of(table)
    return None if h is None else h.get(kind)


DIRECT_PARAM_GRADS = True   # backward kernels accumulate into a preallocated leaf .grad (see _direct_grad)
SINGLE_PASS_MARCH = True   # False: always the published count pass + write pass
FUSED_WGRAD = True    # False: materialised d_act + lse_mlp_wgrad (kept as an in-library cross-check)
FUSED_BIAS_GRAD = True   # per-row bias gradient reduced inside lse_mlp_bwd (False: d_act0 + lse_segment_sum_rows)
RECOMPUTE_ALL = True   # head + base MLPs: save no activations at all, third-generation backward (bf16 pieces) recomputes them
RECOMPUTE_FIRST_LAYER = True   # head MLP: the backward recomputes the first hidden layer instead of reading 1 KiB/sample back
ACT_TILED = True      # tile-major saved activations (1 KiB contiguous per store/load instruction); fused path only


def fused_mlp(params, x, meta: MlpMeta, n: int, row_bias=None, row_bias_idx=None, bias_packed_info=None,
              out_cols: int = 16, density=None, n_dev=None):
    """Bias-free fused MLP (tcnn layout).  Returns the padded output [n, 16] (or the compact [n, 4] = outputs 0..3).
    ``row_bias[rows, width]`` is added to the layer-0 pre-activation of sample i from row ``row_bias_idx[i]``;
    ``bias_packed_info[rows, 2]`` (start, count) must describe those rows' contiguous sample segments.
    ``density=(selector_or_None, scale)`` fuses the trunc_exp density head on output 0 and returns ``(out, sigma[n])``."""
    selector, scale = (None, None) if density is None else density
    return _MlpFn.apply(params, x, row_bias, row_bias_idx, bias_packed_info, selector, meta, n, out_cols, scale, n_dev)


# ----------------------------------------------------------------------------------------------------
# density activation
# ----------------------------------------------------------------------------------------------------
class _DensityFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, selector, scale: float):
        n = h.shape[0]
        sigma = torch.empty(n, dtype=torch.float32, device=h.device)
        _lib.call("lse_density_fwd", _f32(h, "h"), _chk(selector, torch.uint8, "selector", True), float(scale),
                  ctypes.c_void_p(sigma.data_ptr()), n, _stream())
        ctx.save_for_backward(h, selector)
        ctx.scale = scale
        return sigma

    @staticmethod
    def backward(ctx, d_sigma):
        h, selector = ctx.saved_tensors
        n = h.shape[0]
        d_h = torch.zeros_like(h)
        d_sigma = _c(d_sigma)     # bound to a name: stays alive until after the launch
        _lib.call("lse_density_bwd", _f32(h, "h"), _chk(selector, torch.uint8, "selector", True), float(ctx.scale),
                  _f32(d_sigma, "d_sigma"), ctypes.c_void_p(d_h.data_ptr()), n, _stream())
        return d_h, None, None


def density_from_mlp_out(h: torch.Tensor, selector: Optional[torch.Tensor], scale: float = 1.0) -> torch.Tensor:
    """sigma[N] = scale * trunc_exp(h[:,0]) * selector  (R:lse_nerf/lse_field.py:286-287); h is the [N,16] base output."""
    return _DensityFn.apply(h, selector, scale)


# ----------------------------------------------------------------------------------------------------
# per-ray head features and small dense layers
# ----------------------------------------------------------------------------------------------------
class _RayFeaturesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_d, emb_table, emb_idx):
        R = rays_d.shape[0]
        emb_dim = 0 if emb_table is None else emb_table.shape[1]
        feat = torch.empty((R, 64), dtype=torch.float32, device=rays_d.device)
        _lib.call("lse_ray_features_fwd", _f32(rays_d, "rays_d"), _f32(emb_table, "emb_table", True),
                  _chk(emb_idx, torch.int32, "emb_idx", True), R, emb_dim, ctypes.c_void_p(feat.data_ptr()), _stream())
        ctx.save_for_backward(rays_d, emb_table, emb_idx)
        return feat

    @staticmethod
    def backward(ctx, d_feat):
        rays_d, emb_table, emb_idx = ctx.saved_tensors
        R = rays_d.shape[0]
        emb_dim = 0 if emb_table is None else emb_table.shape[1]
        d_feat = _c(d_feat)
        d_dirs = torch.empty_like(rays_d) if ctx.needs_input_grad[0] else None
        d_emb = torch.zeros_like(emb_table) if (emb_table is not None and ctx.needs_input_grad[1]) else None
        _lib.call("lse_ray_features_bwd", _f32(rays_d, "rays_d"), _f32(d_feat, "d_feat"),
                  _chk(emb_idx, torch.int32, "emb_idx", True), R, emb_dim,
                  0 if emb_table is None else emb_table.shape[0], _f32(d_dirs, "d_dirs", True),
                  _f32(d_emb, "d_emb", True), _stream())
        return d_dirs, d_emb, None


def ray_features(rays_d, emb_table=None, emb_idx=None):
    """[R,64] = [SH16 | 15 zeros | emb32 | 1] per ray (tcnn SH-4 of the shifted direction)."""
    return _RayFeaturesFn.apply(rays_d, emb_table, emb_idx)


class _RayBiasFn(torch.autograd.Function):
    """Per-ray share of the head's first layer (lse_ray_bias_fwd / _bwd): directions + embedding rows -> row_bias[R, width].
    ``head_params`` is the head's tcnn parameter vector; its first width * in_pad floats are W_in[width][in_pad]."""

    @staticmethod
    def forward(ctx, rays_d, emb_table, emb_idx, head_params, width: int):
        R = rays_d.shape[0]
        emb_dim = 0 if emb_table is None else emb_table.shape[1]
        in_pad = (31 + emb_dim + 15) // 16 * 16
        dev = rays_d.device
        feat = torch.empty((R, in_pad), dtype=torch.float32, device=dev)
        row_bias = torch.empty((R, width), dtype=torch.float32, device=dev)
        _lib.call("lse_ray_bias_fwd", _f32(rays_d, "rays_d"), _f32(emb_table, "emb_table", True),
                  _chk(emb_idx, torch.int32, "emb_idx", True), R, emb_dim, _f32(head_params, "head params"), in_pad, width,
                  ctypes.c_void_p(feat.data_ptr()), ctypes.c_void_p(row_bias.data_ptr()), _stream())
        ctx.save_for_backward(rays_d, emb_table, emb_idx, head_params, feat)
        ctx.width, ctx.in_pad, ctx.emb_dim = width, in_pad, emb_dim
        return row_bias

    @staticmethod
    def backward(ctx, d_rb):
        rays_d, emb_table, emb_idx, head_params, feat = ctx.saved_tensors
        R, width, in_pad, emb_dim = rays_d.shape[0], ctx.width, ctx.in_pad, ctx.emb_dim
        d_rb = _c(d_rb)
        d_feat = torch.empty_like(feat)
        d_dirs = torch.empty_like(rays_d) if ctx.needs_input_grad[0] else None
        d_emb = direct_emb = None
        if emb_table is not None and ctx.needs_input_grad[1] and emb_idx is not None:
            direct_emb = _direct_grad(emb_table)
            d_emb = direct_emb if direct_emb is not None else torch.zeros_like(emb_table)
        _lib.call("lse_ray_bias_bwd", _f32(rays_d, "rays_d"), _chk(emb_idx, torch.int32, "emb_idx", True), R, emb_dim,
                  0 if emb_table is None else emb_table.shape[0], _f32(head_params, "head params"), in_pad, width,
                  _f32(d_rb, "d_row_bias"), ctypes.c_void_p(d_feat.data_ptr()), _f32(d_dirs, "d_dirs", True),
                  _f32(d_emb, "d_emb", True), _stream())
        d_params = None
        direct_w = None
        if ctx.needs_input_grad[3]:
            direct_w = _direct_grad(head_params)
            d_params = direct_w if direct_w is not None else torch.zeros_like(head_params)
            if in_pad in (16, 32, 64):
                _lib.call("lse_gemm_tn_acc", _f32(d_rb, "d_row_bias"), width, _f32(feat, "feat"), in_pad, _lib.LSE_IN_ROWMAJOR, R,
                          ctypes.c_void_p(d_params.data_ptr()), in_pad, _stream())
            else:       # other embedding widths (padded input 48, 80 .. 128): a plain [width x R] x [R x in_pad] library GEMM, per ray
                with torch.no_grad():
                    d_params[: width * in_pad].view(width, in_pad).addmm_(d_rb.t(), feat)
        return (d_dirs, None if direct_emb is not None else d_emb, None,
                None if direct_w is not None else d_params, None)


def ray_bias(rays_d, emb_table, emb_idx, head_params, width: int):
    """row_bias[R, width] = [SH16(dir) | 0 x 15 | emb[idx] | 1-padding] @ W_in^T, W_in = the head's tcnn input matrix."""
    return _RayBiasFn.apply(rays_d, emb_table, emb_idx, head_params, width)


class _LinearFn(torch.autograd.Function):
    """y[R,M] = x[R,K] @ w[M,K]^T on the HIP helpers (per-ray matrices only)."""

    @staticmethod
    def forward(ctx, x, w):
        R, K = x.shape
        M = w.shape[0]
        y = torch.empty((R, M), dtype=torch.float32, device=x.device)
        _lib.call("lse_linear_fwd", _f32(w, "w"), _f32(x, "x"), R, M, K, ctypes.c_void_p(y.data_ptr()), _stream())
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        R, K = x.shape
        M = w.shape[0]
        dy = _c(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("lse_linear_bwd_input", _f32(w, "w"), _f32(dy, "dy"), R, M, K, ctypes.c_void_p(dx.data_ptr()),
                      _stream())
        if ctx.needs_input_grad[1]:
            dw = torch.zeros_like(w)
            _lib.call("lse_gemm_tn_acc", _f32(dy, "dy"), M, _f32(x, "x"), K, _lib.LSE_IN_ROWMAJOR, R,
                      ctypes.c_void_p(dw.data_ptr()), K, _stream())
        return dx, dw


def linear(x, w):
    return _LinearFn.apply(x, w)


# ----------------------------------------------------------------------------------------------------
# volume rendering
# ----------------------------------------------------------------------------------------------------
class _VolRendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t_starts, t_ends, sigmas, rgb, packed_info, finish_depth: bool):
        R = packed_info.shape[0]
        n = t_starts.shape[0]
        dev = t_starts.device
        weights = torch.empty(n, dtype=torch.float32, device=dev)
        out_rgb = torch.empty((R, 3), dtype=torch.float32, device=dev) if rgb is not None else None
        out_acc = torch.empty(R, dtype=torch.float32, device=dev)
        out_dep = torch.empty(R, dtype=torch.float32, device=dev)      # depth numerator: sum w * (ts + te) / 2
        stride = rgb.stride(0) if rgb is not None else 0
        head = (_f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas"), _rgb_ptr(rgb), stride,
                _chk(packed_info, torch.int64, "packed_info"), R, ctypes.c_void_p(weights.data_ptr()),
                _f32(out_rgb, "out_rgb", True), ctypes.c_void_p(out_acc.data_ptr()), ctypes.c_void_p(out_dep.data_ptr()))
        depth = rng = None
        if finish_depth:     # DepthRenderer("expected") epilogue in the same call: one [R,2] workspace, no pass over N
            ws = torch.empty((R, 2), dtype=torch.float32, device=dev)
            depth = torch.empty(R, dtype=torch.float32, device=dev)
            rng = torch.empty(2, dtype=torch.float32, device=dev)
            _lib.call("lse_volrend_depth_fwd", *head, ctypes.c_void_p(ws.data_ptr()), ctypes.c_void_p(depth.data_ptr()),
                      ctypes.c_void_p(rng.data_ptr()), _stream())
        else:
            _lib.call("lse_volrend_fwd", *head, _stream())
        ctx.save_for_backward(t_starts, t_ends, sigmas, rgb, packed_info, weights, out_acc, out_dep, rng)
        ctx.finish_depth = finish_depth
        ctx.mark_non_differentiable(weights)
        ctx.set_materialize_grads(False)     # unused outputs (accumulation / depth) arrive as None, not as zero-filled tensors
        if out_rgb is None:
            out_rgb = torch.zeros((R, 3), dtype=torch.float32, device=dev)
        return out_rgb, out_acc, (depth if finish_depth else out_dep), weights

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_dep, _g_w):
        t_starts, t_ends, sigmas, rgb, packed_info, weights, out_acc, out_dep, rng = ctx.saved_tensors
        R = packed_info.shape[0]
        if ctx.finish_depth and g_dep is not None:
            # depth = clip(num / (acc + eps), lo, hi): O(R) chain rule back to (num, acc); zero where the clip is active
            den = out_acc + 1e-10
            raw = out_dep / den
            g = torch.where((raw >= rng[0]) & (raw <= rng[1]), g_dep.reshape(-1), torch.zeros_like(raw))
            extra_acc = -g * raw / den
            g_acc = extra_acc if g_acc is None else g_acc.reshape(-1) + extra_acc
            g_dep = g / den
        # every packed sample belongs to exactly one ray, so the kernel writes all of d_sigma; with the compact [N, 4] colour
        # layout it also writes the pad column (one 16-B store per sample), so neither buffer needs a zero-fill
        d_sigma = torch.empty_like(sigmas)
        stride = rgb.stride(0) if rgb is not None else 0
        d_rgb = None
        if rgb is not None and ctx.needs_input_grad[3]:
            d_rgb = torch.empty_like(rgb) if (stride == 4 and rgb.is_contiguous() and rgb.data_ptr() % 16 == 0) else torch.zeros_like(rgb)
        # contiguous copies are bound to names: a temporary freed before the launch could be re-used by the next allocation
        g_rgb = _c(g_rgb) if g_rgb is not None else None
        g_acc = _c(g_acc) if g_acc is not None else None
        g_dep = _c(g_dep) if g_dep is not None else None
        _lib.call("lse_volrend_bwd", _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas"),
                  _rgb_ptr(rgb), stride, _chk(packed_info, torch.int64, "packed_info"), R, _f32(weights, "weights"),
                  _f32(g_rgb, "g_rgb", True), _f32(g_acc, "g_acc", True), _f32(g_dep, "g_dep", True),
                  ctypes.c_void_p(d_sigma.data_ptr()), _rgb_ptr(d_rgb), _stream())
        return None, None, d_sigma, d_rgb, None, None


class _RenderWeightFn(torch.autograd.Function):
    """nerfacc.render_weight_from_density on packed samples (generic route: the caller composites with the weights)."""

    @staticmethod
    def forward(ctx, t_starts, t_ends, sigmas, packed_info):
        R = packed_info.shape[0]
        weights = torch.empty_like(sigmas)
        trans = torch.empty_like(sigmas)
        alphas = torch.empty_like(sigmas)
        _lib.call("lse_render_weight_fwd", _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas"),
                  _chk(packed_info, torch.int64, "packed_info"), R, ctypes.c_void_p(weights.data_ptr()),
                  ctypes.c_void_p(trans.data_ptr()), ctypes.c_void_p(alphas.data_ptr()), _stream())
        ctx.save_for_backward(t_starts, t_ends, sigmas, packed_info, weights)
        ctx.mark_non_differentiable(trans, alphas)
        return weights, trans, alphas

    @staticmethod
    def backward(ctx, g_w, _g_t, _g_a):
        t_starts, t_ends, sigmas, packed_info, weights = ctx.saved_tensors
        if g_w is None:
            return None, None, None, None
        g_w = _c(g_w)
        d_sigma = torch.empty_like(sigmas)
        _lib.call("lse_render_weight_bwd", _f32(t_starts, "t_starts"), _f32(t_ends, "t_ends"), _f32(sigmas, "sigmas"),
                  _chk(packed_info, torch.int64, "packed_info"), packed_info.shape[0], _f32(weights, "weights"),
                  _f32(g_w, "d_weights"), ctypes.c_void_p(d_sigma.data_ptr()), _stream())
        return None, None, d_sigma, None


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info):
    """(weights, transmittance, alphas), each [N]; differentiable w.r.t. sigmas through the weights."""
    if t_starts.shape[0] == 0:
        z = torch.zeros_like(sigmas)
        return z, z.clone(), z.clone()
    return _RenderWeightFn.apply(_c(t_starts), _c(t_ends), _c(sigmas), packed_info)


def _rgb_ptr(rgb):
    if rgb is None:
        return None
    if not rgb.is_cuda or rgb.dtype != torch.float32 or rgb.dim() != 2 or rgb.stride(1) != 1 or rgb.stride(0) < 3:
        raise ValueError("rgb must be a float32 GPU matrix [N, >=3] with unit column stride")
    return ctypes.c_void_p(rgb.data_ptr())


def volume_render(t_starts, t_ends, sigmas, rgb, packed_info):
    """render_weight_from_density + accumulate_along_rays (R:lse_nerf/lsenerf.py:301-318).
    ``rgb`` may be the padded [N,16] head output (only columns 0..2 are read).
    Returns (rgb[R,3], accumulation[R], depth_numerator[R] = sum w*(ts+te)/2, weights[N])."""
    return _VolRendFn.apply(t_starts, t_ends, sigmas, rgb, packed_info, False)


def volume_render_depth(t_starts, t_ends, sigmas, rgb, packed_info):
    """``volume_render`` + the DepthRenderer("expected") epilogue (R:lse_nerf/lsenerf.py:315-317) in one call:
    returns (rgb[R,3], accumulation[R], depth[R] = clip(num / (acc + 1e-10), min mid-point, max mid-point), weights[N])."""
    return _VolRendFn.apply(t_starts, t_ends, sigmas, rgb, packed_info, True)


# ----------------------------------------------------------------------------------------------------
# training epilogue: routing + mappers + losses
# ----------------------------------------------------------------------------------------------------
def _mapper_mlp_shapes(in_dim: int):
    return [(16, in_dim), (16,), (16, 16), (16,), (16, 16), (16,), (in_dim, 16), (in_dim,)]


def _mapper_mlp_struct(params, grads=None) -> Optional["_lib.MapperMlp"]:
    """lse_mapper_mlp of an MLP intensity mapper: ``params`` = its eight tensors in module order (layers.0.weight, layers.0.bias,
    ... layers.3.bias: nn.Linear layout), ``grads`` = the eight buffers the backward accumulates into."""
    if not params:
        return None
    in_dim = params[0].shape[1]
    if len(params) != 8 or in_dim not in (1, 3) or [tuple(p.shape) for p in params] != _mapper_mlp_shapes(in_dim):
        raise ValueError(f"MLP mapper parameters {[tuple(p.shape) for p in params]}: expected the four nn.Linear layers of "
                         f"MLP(in, num_layers=4, layer_width=16, out=in), in = 1 | 3")
    m = _lib.MapperMlp()
    for l in range(4):
        m.w[l] = _f32(params[2 * l], f"mlp.w[{l}]").value
        m.b[l] = _f32(params[2 * l + 1], f"mlp.b[{l}]").value
        if grads is not None:
            m.dw[l] = _f32(grads[2 * l], f"mlp.dw[{l}]").value
            m.db[l] = _f32(grads[2 * l + 1], f"mlp.db[{l}]").value
    return m


def _mapper_mlp_grads(params):
    """(buffers the kernel accumulates into, what the autograd Function returns for the eight inputs): a parameter whose .grad is
    preallocated (optim.FlatParams) takes its gradient there and the Function returns None for it (``_direct_grad``)."""
    bufs, rets = [], []
    for p in params:
        direct = _direct_grad(p)
        if direct is not None:
            bufs.append(direct)
            rets.append(None)
        else:
            g = torch.zeros_like(p, memory_format=torch.contiguous_format)
            bufs.append(g)
            rets.append(g)
    return bufs, tuple(rets)


def _byref(struct):
    return ctypes.byref(struct) if struct is not None else None


class _LossEpilogueFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, desc_fields, col_rgb, col_gt, prev_rgb, next_rgb, evs_gt, pow_rgb, pow_evs, w31, e_thresh=None, n_mlp_rgb=0,
                *mlp_params):
        dev = (col_rgb if col_rgb is not None else prev_rgb).device
        desc = _lib.EpilogueDesc(*desc_fields)
        group = desc.deblur_group
        n_col = col_rgb.shape[0] // group if col_rgb is not None else 0
        n_ev = prev_rgb.shape[0] if prev_rgb is not None else 0
        if col_rgb is not None and (col_rgb.shape[0] % group != 0 or col_gt.shape[0] != n_col):
            raise ValueError(f"colour bundle of {col_rgb.shape[0]} rays does not split into groups of {group} for {col_gt.shape[0]} targets")
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.call("lse_loss_epilogue_fwd", ctypes.byref(desc), _f32(col_rgb, "col_rgb", True), _f32(col_gt, "col_gt", True),
                  n_col, _f32(prev_rgb, "prev_rgb", True), _f32(next_rgb, "next_rgb", True), _f32(evs_gt, "evs_gt", True),
                  _f32(e_thresh, "e_thresh", True), n_ev, _f32(pow_rgb, "pow_rgb", True), _f32(pow_evs, "pow_evs", True),
                  _f32(w31, "w31", True),
                  _byref(_mapper_mlp_struct(mlp_params[:n_mlp_rgb])), _byref(_mapper_mlp_struct(mlp_params[n_mlp_rgb:])),
                  ctypes.c_void_p(losses.data_ptr()), _stream())
        ctx.save_for_backward(col_rgb, col_gt, prev_rgb, next_rgb, evs_gt, pow_rgb, pow_evs, w31, e_thresh, *mlp_params)
        ctx.desc_fields, ctx.n_col, ctx.n_ev, ctx.n_mlp_rgb = desc_fields, n_col, n_ev, n_mlp_rgb
        ctx.set_materialize_grads(False)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_rgb, g_evs):
        col_rgb, col_gt, prev_rgb, next_rgb, evs_gt, pow_rgb, pow_evs, w31, e_thresh, *mlp_params = ctx.saved_tensors
        dev = (col_rgb if col_rgb is not None else prev_rgb).device
        desc = _lib.EpilogueDesc(*ctx.desc_fields)
        mlp_bufs, mlp_rets = _mapper_mlp_grads(mlp_params)
        k = ctx.n_mlp_rgb
        g_rgb = g_rgb.reshape(1).contiguous().float() if g_rgb is not None else None
        g_evs = g_evs.reshape(1).contiguous().float() if g_evs is not None else None
        d_col = torch.empty_like(col_rgb) if col_rgb is not None else None
        d_prev = torch.empty_like(prev_rgb) if prev_rgb is not None else None
        d_next = torch.empty_like(next_rgb) if next_rgb is not None else None
        d_sc = torch.empty(5, dtype=torch.float32, device=dev)
        _lib.call("lse_loss_epilogue_bwd", ctypes.byref(desc), _f32(col_rgb, "col_rgb", True), _f32(col_gt, "col_gt", True),
                  ctx.n_col, _f32(prev_rgb, "prev_rgb", True), _f32(next_rgb, "next_rgb", True), _f32(evs_gt, "evs_gt", True),
                  _f32(e_thresh, "e_thresh", True), ctx.n_ev, _f32(pow_rgb, "pow_rgb", True), _f32(pow_evs, "pow_evs", True),
                  _f32(w31, "w31", True),
                  _byref(_mapper_mlp_struct(mlp_params[:k], mlp_bufs[:k])), _byref(_mapper_mlp_struct(mlp_params[k:], mlp_bufs[k:])),
                  _f32(g_rgb, "g_rgb_loss", True), _f32(g_evs, "g_event_loss", True), _f32(d_col, "d_col", True),
                  _f32(d_prev, "d_prev", True), _f32(d_next, "d_next", True), ctypes.c_void_p(d_sc.data_ptr()), _stream())
        return (None, d_col, None, d_prev, d_next, None) + _scalar_param_grads(d_sc, pow_rgb, pow_evs, w31) + (None, None) + mlp_rets


def _scalar_param_grads(d_sc, pow_rgb, pow_evs, w31):
    """Gradients of the epilogue's three scalar parameters (d_sc = [d pow_rgb, d pow_evs, d w31[3]]).  A parameter whose .grad is
    already allocated (optim.FlatParams: a view of the flat gradient buffer) is accumulated into HERE and the Function returns None
    for it, like the big parameters (_direct_grad): the step then runs no AccumulateGrad node of a model parameter at all.  That is
    what makes a captured step indifferent to EARLIER eager graphs of the same model that are still alive -- they keep the
    parameters' AccumulateGrad nodes alive, bound to the stream they were created on, and a capture whose backward executed such a
    node left the capturing stream (torch warns; the HIP runtime crashed in hipStreamEndCapture, round 4)."""
    out = []
    for p, lo, hi in ((pow_rgb, 0, 1), (pow_evs, 1, 2), (w31, 2, 5)):
        if p is None:
            out.append(None)
            continue
        g = d_sc[lo:hi].view_as(p)
        direct = _direct_grad(p)
        if direct is not None:
            direct.add_(g)
            out.append(None)
        else:
            out.append(g)
    return tuple(out)


class _LossEpiloguePackedFn(torch.autograd.Function):
    """``_LossEpilogueFn`` on the render of ONE packed pass: ``rgb_all`` [n_col_rays + 2 n_ev, 3] holds the colour, previous-event and
    next-event bundles' rows one after the other (LSENeRFModel.train_step_bundles).  The three parts are handed to the kernels as
    pointers into that one buffer, and the backward fills ONE gradient buffer -- the per-bundle slices of the generic route cost, in
    backward, a zero-fill + a copy per bundle and two adds (eight ~5 us launches per step in the rocprofv3 timeline of a replayed
    3-bundle step, tools/graph_timeline.py)."""

    @staticmethod
    def forward(ctx, desc_fields, rgb_all, n_col_rays, n_ev, col_gt, evs_gt, pow_rgb, pow_evs, w31, e_thresh=None, n_mlp_rgb=0,
                *mlp_params):
        dev = rgb_all.device
        desc = _lib.EpilogueDesc(*desc_fields)
        group = desc.deblur_group
        if n_col_rays % group != 0 or (n_col_rays and col_gt.shape[0] != n_col_rays // group):
            raise ValueError(f"colour bundle of {n_col_rays} rays does not split into groups of {group} for "
                             f"{0 if col_gt is None else col_gt.shape[0]} targets")
        if rgb_all.shape[0] != n_col_rays + 2 * n_ev:
            raise ValueError(f"packed render of {rgb_all.shape[0]} rays for {n_col_rays} colour + 2 x {n_ev} event rays")
        col = rgb_all[:n_col_rays] if n_col_rays else None
        prev = rgb_all[n_col_rays:n_col_rays + n_ev] if n_ev else None
        nxt = rgb_all[n_col_rays + n_ev:] if n_ev else None
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.call("lse_loss_epilogue_fwd", ctypes.byref(desc), _f32(col, "col_rgb", True), _f32(col_gt, "col_gt", True),
                  n_col_rays // group, _f32(prev, "prev_rgb", True), _f32(nxt, "next_rgb", True), _f32(evs_gt, "evs_gt", True),
                  _f32(e_thresh, "e_thresh", True), n_ev, _f32(pow_rgb, "pow_rgb", True), _f32(pow_evs, "pow_evs", True),
                  _f32(w31, "w31", True),
                  _byref(_mapper_mlp_struct(mlp_params[:n_mlp_rgb])), _byref(_mapper_mlp_struct(mlp_params[n_mlp_rgb:])),
                  ctypes.c_void_p(losses.data_ptr()), _stream())
        ctx.save_for_backward(rgb_all, col_gt, evs_gt, pow_rgb, pow_evs, w31, e_thresh, *mlp_params)
        ctx.desc_fields, ctx.n_col_rays, ctx.n_ev, ctx.n_mlp_rgb = desc_fields, n_col_rays, n_ev, n_mlp_rgb
        ctx.set_materialize_grads(False)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_rgb, g_evs):
        rgb_all, col_gt, evs_gt, pow_rgb, pow_evs, w31, e_thresh, *mlp_params = ctx.saved_tensors
        dev = rgb_all.device
        desc = _lib.EpilogueDesc(*ctx.desc_fields)
        mlp_bufs, mlp_rets = _mapper_mlp_grads(mlp_params)
        k = ctx.n_mlp_rgb
        n0, ne = ctx.n_col_rays, ctx.n_ev
        g_rgb = g_rgb.reshape(1).contiguous().float() if g_rgb is not None else None
        g_evs = g_evs.reshape(1).contiguous().float() if g_evs is not None else None
        d_all = torch.empty_like(rgb_all)          # every row belongs to exactly one part, and the kernel overwrites its parts
        part = lambda t, a, b: (t[a:b] if b > a else None)
        _lib.call("lse_loss_epilogue_bwd", ctypes.byref(desc), _f32(part(rgb_all, 0, n0), "col_rgb", True), _f32(col_gt, "col_gt", True),
                  n0 // desc.deblur_group, _f32(part(rgb_all, n0, n0 + ne), "prev_rgb", True),
                  _f32(part(rgb_all, n0 + ne, n0 + 2 * ne), "next_rgb", True), _f32(evs_gt, "evs_gt", True),
                  _f32(e_thresh, "e_thresh", True), ne,
                  _f32(pow_rgb, "pow_rgb", True), _f32(pow_evs, "pow_evs", True), _f32(w31, "w31", True),
                  _byref(_mapper_mlp_struct(mlp_params[:k], mlp_bufs[:k])), _byref(_mapper_mlp_struct(mlp_params[k:], mlp_bufs[k:])),
                  _f32(g_rgb, "g_rgb_loss", True), _f32(g_evs, "g_event_loss", True), _f32(part(d_all, 0, n0), "d_col", True),
                  _f32(part(d_all, n0, n0 + ne), "d_prev", True), _f32(part(d_all, n0 + ne, n0 + 2 * ne), "d_next", True),
                  ctypes.c_void_p((d_sc := torch.empty(5, dtype=torch.float32, device=dev)).data_ptr()), _stream())
        return (None, d_all, None, None, None, None) + _scalar_param_grads(d_sc, pow_rgb, pow_evs, w31) + (None, None) + mlp_rets


def _event_thresholds(e_thresh, n_ev: int, like):
    """evs_batch["e_thresh"] (a number, a 1-element tensor or one value per event ray, R:lse_nerf/lse_pixel_sampler.py:36-37) as the
    [n_ev] float vector of lse_loss_epilogue_*; None stays None (= 1)."""
    if e_thresh is None or like is None:
        return None
    if not torch.is_tensor(e_thresh):      # a number: a fill kernel (capturable), not a host-to-device copy
        return torch.full((n_ev,), float(e_thresh), dtype=torch.float32, device=like.device)
    t = e_thresh.to(device=like.device, dtype=torch.float32).reshape(-1)
    if t.numel() == 1:
        t = t.expand(n_ev)
    if t.numel() != n_ev:
        raise ValueError(f"e_thresh holds {t.numel()} values for {n_ev} event rays")
    return t.contiguous()


def loss_epilogue_packed(desc_fields: tuple, rgb_all, n_col_rays: int, n_ev: int, col_gt, evs_gt, pow_rgb=None, pow_evs=None, w31=None,
                         mlp_rgb=(), mlp_evs=(), e_thresh=None):
    """``loss_epilogue`` for the bundles of one packed pass, given as row blocks [colour | previous | next] of ONE render."""
    c = lambda t: _c(t.float()) if t is not None else None
    return _LossEpiloguePackedFn.apply(tuple(desc_fields), c(rgb_all), int(n_col_rays), int(n_ev), c(col_gt),
                                       c(evs_gt.reshape(-1)) if evs_gt is not None else None, pow_rgb, pow_evs, w31,
                                       _event_thresholds(e_thresh, int(n_ev), evs_gt), len(mlp_rgb), *mlp_rgb, *mlp_evs)


def loss_epilogue(desc_fields: tuple, col_rgb, col_gt, prev_rgb, next_rgb, evs_gt, pow_rgb=None, pow_evs=None, w31=None,
                  mlp_rgb=(), mlp_evs=(), e_thresh=None):
    """Routing + intensity mappers + rgb MSE + log-intensity event MSE in one launch (lse_loss_epilogue_fwd).
    ``desc_fields`` = (rgb_mapped, rgb_mapper, evs_mapper, ev_one_dim, deblur_group, evs_loss_weight[, event_loss_kind]).
    ``e_thresh``: evs_batch["e_thresh"] for event_loss_kind = LSE_EVLOSS_ENERF_NORM (R:lse_nerf/lsenerf.py:406-419).
    ``mlp_rgb`` / ``mlp_evs``: the eight parameters (module order) of an MLP mapper on the colour / event side (kinds LSE_MAP_MLP,
    LSE_MAP_RGB_MLP; R:lse_nerf/intensity_mappers.py:28-62).  Returns (rgb_loss, event_loss) as 0-dim tensors."""
    c = lambda t: _c(t.float()) if t is not None else None
    return _LossEpilogueFn.apply(tuple(desc_fields), c(col_rgb), c(col_gt), c(prev_rgb), c(next_rgb),
                                 c(evs_gt.reshape(-1)) if evs_gt is not None else None, pow_rgb, pow_evs, w31,
                                 _event_thresholds(e_thresh, prev_rgb.shape[0] if prev_rgb is not None else 0, evs_gt),
                                 len(mlp_rgb), *mlp_rgb, *mlp_evs)


# ----------------------------------------------------------------------------------------------------
# optimiser / occupancy grid
# ----------------------------------------------------------------------------------------------------
@torch.no_grad()
def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step: int, grad_scale: float = 1.0):
    _lib.call("lse_adam_step", _f32(params, "params"), _f32(grads, "grads"), _f32(exp_avg, "exp_avg"),
              _f32(exp_avg_sq, "exp_avg_sq"), params.numel(), float(lr), float(beta1), float(beta2), float(eps),
              int(step), float(grad_scale), _stream())


@torch.no_grad()
def adam_step_dev(params, grads, exp_avg, exp_avg_sq, hyper, grad_scale: float = 1.0):
    """Adam with every scalar of the step -- lr, 1 - beta1^t, 1 / sqrt(1 - beta2^t), beta1, beta2, eps -- in ``hyper`` (float32 [6]
    on the device, written by ``adam_schedule_dev``)."""
    _lib.call("lse_adam_step_dev", _f32(params, "params"), _f32(grads, "grads"), _f32(exp_avg, "exp_avg"),
              _f32(exp_avg_sq, "exp_avg_sq"), params.numel(), _f32(hyper, "hyper"), float(grad_scale), _stream())


@torch.no_grad()
def adam_schedule_dev(step_dev, hyper, sched):
    """Advance the device-side optimizer step counter (int64 [1]) and write the six scalars of the new step into ``hyper``
    (float32 [6]) from the schedule constants ``sched`` (float64 [6] on the device: lr_init, lr_final, max_steps, beta1, beta2,
    eps) -- lse_adam_schedule_dev; capturable, reads nothing from the host."""
    _lib.call("lse_adam_schedule_dev", _chk(step_dev, torch.int64, "step_dev"), _f32(hyper, "hyper"),
              _chk(sched, torch.float64, "sched"), _stream())


@torch.no_grad()
def occ_update_cells(occs, cell_ids, occ_new, ema_decay: float):
    n = cell_ids.shape[0]
    ws = torch.empty(n, dtype=torch.float32, device=occs.device)
    _lib.call("lse_occ_update_cells", _f32(occs, "occs"), _chk(cell_ids, torch.int64, "cell_ids"),
              _f32(occ_new, "occ_new"), n, float(ema_decay), ctypes.c_void_p(ws.data_ptr()), _stream())


@torch.no_grad()
def occ_binarize(occs, threshold: torch.Tensor, binaries_u8):
    _lib.call("lse_occ_binarize", _f32(occs, "occs"), occs.numel(), _f32(threshold, "threshold"),
              _chk(binaries_u8, torch.uint8, "binaries"), _stream())
