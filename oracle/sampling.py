"""Oracle: occupancy-grid ray marching (nerfacc 0.5.2 restated).  TEST INFRASTRUCTURE.  Parity unpinned.

* ``traverse_grids``      -> C restatement in oracle/c/lse_oracle.c (strict fp32), loaded via ctypes.
* ``traverse_grids_py``   -> the same algorithm as a numpy-float32 scalar loop (slow; cross-checks the C).
* ``OccGridOracle``       -> nerfacc ``OccGridEstimator`` state/update with the
                              reference's ``sampling`` override (R:lse_nerf/lse_grid_estimator.py:15-143).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Callable, Optional, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c_oracle(fma: bool = False) -> str:
    """Compile oracle/c/lse_oracle.c with gcc (no GPU involved).  Returns the .so path.  ``fma``: the build with explicit
    fused multiply-adds at grid.cu's a*b+c sites (what nvcc's default contraction makes of them)."""
    cdir = os.path.join(_HERE, "c")
    so = os.path.join(cdir, "liblse_oracle_fma.so" if fma else "liblse_oracle.so")
    src = os.path.join(cdir, "lse_oracle.c")
    if (not os.path.exists(so)) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", cdir, "-s"])
    return so


_LIB_FMA = None


def _lib(fma: bool = False):
    global _LIB, _LIB_FMA
    if fma:
        if _LIB_FMA is None:
            _LIB_FMA = ctypes.CDLL(build_c_oracle(True))
            assert _LIB_FMA.lse_oracle_fma_build() == 1
        return _LIB_FMA
    if _LIB is None:
        _LIB = ctypes.CDLL(build_c_oracle())
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def enlarge_aabb(aabb: torch.Tensor, factor: float) -> torch.Tensor:
    """nerfacc ``_enlarge_aabb``."""
    center = (aabb[:3] + aabb[3:]) / 2
    extent = (aabb[3:] - aabb[:3]) / 2
    return torch.cat([center - extent * factor, center + extent * factor])


def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane=-float("inf"), far_plane=float("inf"),
                       miss_value=float("inf")):
    o = np.ascontiguousarray(rays_o.detach().cpu().numpy(), dtype=np.float32)
    d = np.ascontiguousarray(rays_d.detach().cpu().numpy(), dtype=np.float32)
    a = np.ascontiguousarray(aabbs.detach().cpu().numpy(), dtype=np.float32)
    R, M = o.shape[0], a.shape[0]
    tmin = np.empty((R, M), np.float32)
    tmax = np.empty((R, M), np.float32)
    hits = np.empty((R, M), np.uint8)
    _lib().lse_oracle_ray_aabb_intersect(_p(o), _p(d), ctypes.c_int(R), _p(a), ctypes.c_int(M),
                                         ctypes.c_float(near_plane), ctypes.c_float(far_plane),
                                         ctypes.c_float(miss_value), _p(tmin), _p(tmax), _p(hits))
    return torch.from_numpy(tmin), torch.from_numpy(tmax), torch.from_numpy(hits.astype(bool))


def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes=None, far_planes=None, step_size=1e-3,
                   cone_angle=0.0, fma: bool = False):
    """nerfacc.grid.traverse_grids as consumed at R:lse_nerf/lse_grid_estimator.py:93-106.

    Returns (ray_indices int64 [N], t_starts f32 [N], t_ends f32 [N], packed_info int64 [R,2]).  ``fma``: evaluate the a*b+c
    sites of grid.cu as fused multiply-adds (second build of the C oracle)."""
    o = np.ascontiguousarray(rays_o.detach().cpu().numpy(), dtype=np.float32)
    d = np.ascontiguousarray(rays_d.detach().cpu().numpy(), dtype=np.float32)
    b = np.ascontiguousarray(binaries.detach().cpu().numpy().astype(np.uint8))
    a = np.ascontiguousarray(aabbs.detach().cpu().numpy(), dtype=np.float32)
    R = o.shape[0]
    L, rx, ry, rz = b.shape
    near = np.zeros(R, np.float32) if near_planes is None else np.ascontiguousarray(
        near_planes.detach().cpu().numpy(), dtype=np.float32)
    far = np.full(R, np.inf, np.float32) if far_planes is None else np.ascontiguousarray(
        far_planes.detach().cpu().numpy(), dtype=np.float32)
    cnts = np.zeros(R, np.int64)
    f = _lib(fma).lse_oracle_traverse_grids

    def call(mode, starts, ri, ts, te):
        f(_p(o), _p(d), ctypes.c_int(R), _p(b), _p(a), ctypes.c_int(L), ctypes.c_int(rx), ctypes.c_int(ry),
          ctypes.c_int(rz), _p(near), _p(far), ctypes.c_float(step_size), ctypes.c_float(cone_angle),
          ctypes.c_int(mode), _p(cnts), _p(starts), _p(ri), _p(ts), _p(te))

    dummy_i = np.zeros(1, np.int64)
    dummy_f = np.zeros(1, np.float32)
    call(0, dummy_i, dummy_i, dummy_f, dummy_f)
    starts = np.cumsum(cnts) - cnts
    N = int(cnts.sum())
    ri = np.zeros(max(N, 1), np.int64)
    ts = np.zeros(max(N, 1), np.float32)
    te = np.zeros(max(N, 1), np.float32)
    call(1, starts, ri, ts, te)
    packed = np.stack([starts, cnts], axis=-1)
    return (torch.from_numpy(ri[:N].copy()), torch.from_numpy(ts[:N].copy()), torch.from_numpy(te[:N].copy()),
            torch.from_numpy(packed))


# --------------------------------------------------------------------------------------------
# numpy-float32 scalar restatement (independent transcription; slow, small cases only)
# --------------------------------------------------------------------------------------------
def traverse_grids_py(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle):
    f32 = np.float32
    o_all = rays_o.numpy().astype(f32)
    d_all = rays_d.numpy().astype(f32)
    bins = binaries.numpy().astype(bool)
    ab = aabbs.numpy().astype(f32)
    L, rx, ry, rz = bins.shape
    resi = (rx, ry, rz)
    eps = f32(1e-6)
    step = f32(step_size)
    cone = f32(cone_angle)
    out_ri, out_ts, out_te, cnts = [], [], [], []

    def calc_dt(t):
        return f32(min(max(f32(t * cone), step), f32(1e10)))

    def slab(o, inv, box):
        def axis(a):
            if inv[a] >= 0:
                return f32(f32(box[a] - o[a]) * inv[a]), f32(f32(box[3 + a] - o[a]) * inv[a])
            return f32(f32(box[3 + a] - o[a]) * inv[a]), f32(f32(box[a] - o[a]) * inv[a])
        tmin, tmax = axis(0)
        for a in (1, 2):
            t0, t1 = axis(a)
            if tmin > t1 or t0 > tmax:
                return False, f32(0), f32(0)
            if t0 > tmin:
                tmin = t0
            if t1 < tmax:
                tmax = t1
        if tmax <= 0:
            return False, f32(0), f32(0)
        return True, tmin, tmax

    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for tid in range(o_all.shape[0]):
            o, d = o_all[tid], d_all[tid]
            inv = (f32(1.0) / d).astype(f32)
            near, far = f32(near_planes[tid]), f32(far_planes[tid])
            hits, vals = [], [f32(0)] * (2 * L)
            for l in range(L):
                h, t0, t1 = slab(o, inv, ab[l])
                hits.append(h)
                vals[l] = t0 if h else f32(np.inf)
                vals[L + l] = t1 if h else f32(np.inf)
            order = list(range(2 * L))
            if L > 1:
                order = sorted(order, key=lambda i: vals[i])     # python sort is stable
            ts = [vals[i] for i in order]
            n = 0
            t_last = near
            continuous = False
            for i in range(2 * L - 1):
                entering = order[i] < L
                level = order[i] % L
                if not hits[level]:
                    continue
                if not entering:
                    if order[i + 1] < L:
                        continue
                    level = order[i + 1] % L
                    if not hits[level]:
                        continue
                tmin = f32(max(ts[i], near))
                tmax = f32(min(ts[i + 1], far))
                if tmin >= tmax:
                    continue
                if not continuous:
                    if step <= 0:
                        t_last = tmin
                    else:
                        while True:
                            dt = calc_dt(t_last)
                            if f32(t_last + f32(dt * f32(0.5))) >= tmin:
                                break
                            t_last = f32(t_last + dt)
                box = ab[level]
                cur, fin, stp, ovf = [0] * 3, [0] * 3, [0] * 3, [0] * 3
                tdist, delta = [f32(0)] * 3, [f32(0)] * 3
                for a in range(3):
                    res = f32(resi[a])
                    voxel = f32(f32(box[3 + a] - box[a]) / res)
                    rs = f32(o[a] + f32(d[a] * f32(tmin + eps)))
                    re = f32(o[a] + f32(d[a] * f32(tmax - eps)))
                    ext = f32(box[3 + a] - box[a])
                    cur[a] = int(np.clip(int(f32(f32(f32(rs - box[a]) / ext) * res)), 0, resi[a] - 1))
                    fin[a] = int(np.clip(int(f32(f32(f32(re - box[a]) / ext) * res)), 0, resi[a] - 1))
                    start_index = cur[a] + (1 if d[a] > 0 else 0)
                    tmax_a = f32(f32(f32(box[a] + f32(f32(f32(start_index) * voxel) - rs)) * inv[a]) + tmin)
                    tdist[a] = tmax if d[a] == 0 else tmax_a
                    stepf = f32(0) if d[a] == 0 else (f32(1) if d[a] > 0 else f32(-1))
                    stp[a] = int(stepf)
                    delta[a] = tmax if d[a] == 0 else f32(f32(voxel * inv[a]) * stepf)
                    ovf[a] = fin[a] + stp[a]
                while True:
                    t_trav = f32(min(min(tdist[0], min(tdist[1], tdist[2])), tmax))
                    if not bins[level, cur[0], cur[1], cur[2]]:
                        if step <= 0:
                            t_last = t_trav
                        else:
                            while True:
                                dt = calc_dt(t_last)
                                if f32(t_last + f32(dt * f32(0.5))) >= t_trav:
                                    break
                                t_last = f32(t_last + dt)
                        continuous = False
                    else:
                        while True:
                            if step <= 0:
                                t_next = t_trav
                            else:
                                dt = calc_dt(t_last)
                                if f32(t_last + f32(dt * f32(0.5))) >= t_trav:
                                    break
                                t_next = f32(t_last + dt)
                            out_ri.append(tid)
                            out_ts.append(t_last)
                            out_te.append(t_next)
                            n += 1
                            continuous = True
                            t_last = t_next
                            if t_next >= t_trav:
                                break
                    if tdist[0] < tdist[1] and tdist[0] < tdist[2]:
                        a = 0
                    elif tdist[1] < tdist[2]:
                        a = 1
                    else:
                        a = 2
                    cur[a] += stp[a]
                    tdist[a] = f32(tdist[a] + delta[a])
                    if cur[a] == ovf[a]:
                        break
                    if not all(0 <= cur[k] < resi[k] for k in range(3)):
                        break
            cnts.append(n)
    cnts = np.asarray(cnts, np.int64)
    starts = np.cumsum(cnts) - cnts
    return (torch.tensor(out_ri, dtype=torch.int64), torch.tensor(np.asarray(out_ts, np.float32)),
            torch.tensor(np.asarray(out_te, np.float32)), torch.from_numpy(np.stack([starts, cnts], -1)))


# --------------------------------------------------------------------------------------------
# OccGridEstimator state + update (App. A.7) and the reference's sampling() override
# --------------------------------------------------------------------------------------------
class OccGridOracle:
    def __init__(self, roi_aabb: torch.Tensor, resolution: int = 128, levels: int = 4):
        self.resolution = torch.tensor([resolution] * 3, dtype=torch.int32)
        self.levels = levels
        self.cells_per_lvl = resolution ** 3
        self.aabbs = torch.stack([enlarge_aabb(roi_aabb.flatten().float(), 2 ** i) for i in range(levels)], 0)
        self.occs = torch.zeros(levels * self.cells_per_lvl)
        self.binaries = torch.zeros((levels, resolution, resolution, resolution), dtype=torch.bool)
        r = resolution
        self.grid_coords = torch.stack(torch.meshgrid(torch.arange(r), torch.arange(r), torch.arange(r),
                                                      indexing="ij"), -1).reshape(-1, 3)
        self.grid_indices = torch.arange(self.cells_per_lvl)

    # -- R:lse_nerf/lse_grid_estimator.py:15-143 --------------------------------------------
    def sampling(self, rays_o, rays_d, sigma_fn: Optional[Callable] = None, near_plane=0.0, far_plane=1e10,
                 t_min=None, t_max=None, render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0,
                 stratified=False, cone_angle=0.0, jitter: Optional[torch.Tensor] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        from .volrend import render_visibility_from_density
        near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if t_max is not None:
            far_planes = torch.clamp(far_planes, max=t_max)
        if stratified:
            # torch.rand_like in the reference (:92); the caller supplies the draw so that HIP/oracle agree
            u = jitter if jitter is not None else torch.rand_like(near_planes)
            near_planes = near_planes + u * render_step_size
        ray_indices, t_starts, t_ends, packed_info = traverse_grids(
            rays_o, rays_d, self.binaries, self.aabbs, near_planes, far_planes, render_step_size, cone_angle)
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigma_fn is not None:
            alpha_thre = min(alpha_thre, self.occs.mean().item())
            sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
            masks = render_visibility_from_density(t_starts, t_ends, sigmas.detach(), packed_info,
                                                   early_stop_eps, alpha_thre)
            ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
        return ray_indices, t_starts, t_ends

    # -- nerfacc OccGridEstimator._update / update_every_n_steps ----------------------------
    def all_cells(self):
        return [self.grid_indices[self.occs[l * self.cells_per_lvl:(l + 1) * self.cells_per_lvl] >= 0.0]
                for l in range(self.levels)]

    def sample_uniform_and_occupied_cells(self, n: int, gen: torch.Generator):
        out = []
        for l in range(self.levels):
            uniform = torch.randint(self.cells_per_lvl, (n,), generator=gen)
            lvl_occ = self.occs[l * self.cells_per_lvl + uniform]
            uniform = uniform[lvl_occ >= 0.0]
            occupied = torch.nonzero(self.binaries[l].flatten())[:, 0]
            if n < len(occupied):
                sel = torch.randint(len(occupied), (n,), generator=gen)
                occupied = occupied[sel]
            out.append(torch.cat([uniform, occupied], dim=0))
        return out

    def cell_points(self, lvl: int, indices: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
        """x = aabb_lvl(min) + (coord + u)/res * extent  (App. A.7)."""
        coords = self.grid_coords[indices]
        x = (coords + u) / self.resolution
        ab = self.aabbs[lvl]
        return ab[:3] + x * (ab[3:] - ab[:3])

    def apply_update(self, lvl: int, indices: torch.Tensor, occ: torch.Tensor, ema_decay: float):
        cell_ids = lvl * self.cells_per_lvl + indices
        self.occs[cell_ids] = torch.maximum(self.occs[cell_ids] * ema_decay, occ)

    def finish_update(self, occ_thre: float):
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
        self.binaries = (self.occs > thre).view(self.binaries.shape)

    def update(self, step: int, occ_eval_fn: Callable, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256,
               gen: Optional[torch.Generator] = None):
        gen = gen or torch.Generator().manual_seed(step)
        if step < warmup_steps:
            lvl_indices = self.all_cells()
        else:
            lvl_indices = self.sample_uniform_and_occupied_cells(self.cells_per_lvl // 4, gen)
        for lvl, indices in enumerate(lvl_indices):
            u = torch.rand(len(indices), 3, generator=gen)
            x = self.cell_points(lvl, indices, u)
            occ = occ_eval_fn(x).squeeze(-1)
            self.apply_update(lvl, indices, occ, ema_decay)
        self.finish_update(occ_thre)
