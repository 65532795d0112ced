"""Occupancy-grid estimator on the gfx950 kernels.

``LSEOccGridEstimator.sampling`` mirrors R:lse_nerf/lse_grid_estimator.py:15-143 (argument names, defaults,
assertion text, return order); the inherited nerfacc 0.5.2 ``OccGridEstimator`` surface (buffers ``resolution``,
``aabbs``, ``occs``, ``binaries``; ``update_every_n_steps`` / ``_update``) is restated per SURVEY.md App. A.7.

Differences from the reference, by design:
  * the visibility pre-pass runs under no_grad (the reference builds a dead autograd graph, :14);
  * ``ray_indices`` is int32 internally; ``sampling`` returns int64 like nerfacc unless ``return_packed`` is set;
  * stratified jitter can be supplied by the caller (``jitter``) so that parity tests share the draw;
  * ``_update`` draws its cells and in-cell jitter from the estimator's OWN generator, re-seeded from
    ``(update_seed, step)``: in data-parallel training every rank refreshes the same cells at the same positions, so the
    grids stay bit-identical across ranks without DDP's per-forward buffer broadcast (R:lse_nerf/lse_pipeline.py:97;
    the global RNG is seeded per rank, R:train.py:104).  Pass ``generator=`` to override.
"""
from __future__ import annotations

from typing import Callable, List, NamedTuple, Optional, Tuple, Union

import torch
from torch import Tensor, nn

from . import ops


def _enlarge_aabb(aabb: Tensor, factor: float) -> Tensor:
    center = (aabb[:3] + aabb[3:]) / 2
    extent = (aabb[3:] - aabb[:3]) / 2
    return torch.cat([center - extent * factor, center + extent * factor])


class Premarched(NamedTuple):
    """Result of ``LSEOccGridEstimator.march_deferred``: packed samples of capacity extent with their count on the device."""
    ray_indices: Tensor       # int32 [R * cap]
    t_starts: Tensor          # [R * cap]
    t_ends: Tensor            # [R * cap]
    packed_info: Tensor       # int64 [R, 2]
    n_dev: Tensor             # int64 [1]: valid leading entries
    overflow: Tensor          # int32 [1]: the estimator's sticky accumulator (LSEOccGridEstimator._overflow_flag)
    grid_version: int         # LSEOccGridEstimator.grid_version when the marcher was launched
    n_rays: int


class LSEOccGridEstimator(nn.Module):
    DIM: int = 3

    def __init__(self, roi_aabb: Union[List[float], Tensor], resolution: Union[int, List[int], Tensor] = 128,
                 levels: int = 1, **kwargs) -> None:
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * self.DIM
        if isinstance(resolution, (list, tuple)):
            resolution = torch.tensor(resolution, dtype=torch.int32)
        if isinstance(roi_aabb, (list, tuple)):
            roi_aabb = torch.tensor(roi_aabb, dtype=torch.float32)
        assert roi_aabb.numel() == 6
        roi_aabb = roi_aabb.detach().flatten().float()
        aabbs = torch.stack([_enlarge_aabb(roi_aabb, 2 ** i) for i in range(levels)], dim=0)
        self.cells_per_lvl = int(resolution.prod().item())
        self.levels = levels
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", aabbs)
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))
        r = resolution.tolist()
        grid_coords = torch.stack(torch.meshgrid([torch.arange(k) for k in r], indexing="ij"), dim=-1).reshape(-1, 3)
        self.register_buffer("grid_coords", grid_coords, persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.cells_per_lvl), persistent=False)
        self._occ_mean_host: Optional[float] = None   # cached occs.mean() (refreshed by _update), saves a sync per forward
        self.update_seed: int = 0x15E5EED             # base of the rank-independent update stream (same on every rank)
        self.after_march_hook: Optional[Callable[[], None]] = None   # dist.GradPipeline.flush: runs before sigma_fn
        # arithmetic convention of the traversal set-up (include/lse_hip.h: LSE_TRAVERSE_FMA_SETUP): False = every product and sum
        # rounded separately (bit-exact against oracle/c/liblse_oracle.so); True = nvcc's default contraction of nerfacc's grid.cu
        # (bit-exact against liblse_oracle_fma.so).  About one sample interval per million differs.
        self.traverse_fma: bool = False

    # ---------------------------------------------------------------------------------------------
    def _binaries_u8(self) -> Tensor:
        return self.binaries.view(torch.uint8)

    def _max_span(self, near_plane: float, far_plane: float) -> float:
        """Host-side upper bound of the t-range any ray can spend inside the grids: the diagonal of the outermost aabb,
        clipped by the planes (lets ops.traverse_grids march in a single pass).  Cached per version of `aabbs`."""
        ver = self.aabbs._version
        if getattr(self, "_diag_cache", (None, None))[0] != ver:
            box = self.aabbs[-1].detach().cpu()
            self._diag_cache = (ver, float((box[3:] - box[:3]).norm()))
        return max(0.0, min(self._diag_cache[1], float(far_plane) - float(near_plane))) + 1e-3 * self._diag_cache[1]

    def _cap_per_ray(self, near_plane: float, far_plane: float, step: float, cone: float) -> int:
        """Host-side PROVEN upper bound of the samples one ray can yield.  The marcher's cursor only moves forward, by
        dt(t) = clamp(t * cone, step, 1e10) per step -- through occupied cells (a sample each) and empty ones alike -- from the
        near plane to at most ``near + span`` (span: diagonal of the outermost box, clipped by the planes).  Below
        t_c = step / cone every step is ``step`` long; above, t grows by the factor 1 + cone per step.  A few steps of slack
        cover the re-alignments at interval boundaries."""
        assert step > 0, "deferred sampling needs a positive step size"
        t0 = max(float(near_plane), 0.0)
        t1 = t0 + self._max_span(near_plane, far_plane)
        if cone <= 0.0:
            n = (t1 - t0) / step
        else:
            import math
            tc = step / cone
            n = max(0.0, min(tc, t1) - t0) / step
            if t1 > tc:
                n += math.log(t1 / max(tc, t0, 1e-30)) / math.log1p(cone)
        return int(n) + 16 + 4 * self.levels

    @torch.no_grad()
    def _occ_mean_device(self) -> Tensor:
        """occs.mean() as a float32 [1] device tensor at a fixed address, recomputed (on the device) when `occs` has been
        written since the last call."""
        buf = self.__dict__.get("_occ_mean_dev")
        if buf is None or buf.device != self.occs.device:
            buf = self.__dict__["_occ_mean_dev"] = torch.zeros(1, dtype=torch.float32, device=self.occs.device)
            self.__dict__["_occ_mean_dev_version"] = None
        if self.__dict__.get("_occ_mean_dev_version") != self.occs._version:
            buf.copy_(self.occs.mean().reshape(1))
            self.__dict__["_occ_mean_dev_version"] = self.occs._version
        return buf

    def _overflow_flag(self) -> Tensor:
        """The ONE sticky overflow accumulator of this estimator (int32 [1] on the grid's device): every count-free marcher call --
        eager, captured into a HIP graph, or running ahead on a side stream -- ORs into it and nothing clears it but
        ``check_deferred_overflow``, so no call's flag can be dropped unread however long the host waits to look."""
        buf = self.__dict__.get("_overflow_acc")
        if buf is None or buf.device != self.occs.device:
            buf = self.__dict__["_overflow_acc"] = torch.zeros(1, dtype=torch.int32, device=self.occs.device)
        return buf

    def _invalidate_occ_mean(self) -> None:
        """Call after ANY write to ``occs`` (refresh, mark_all_occupied, dist.sync_grid, load_state_dict).  The HIP kernels write
        `occs` through its raw pointer (no tensor version bump), and a captured step reads the device-side mean at its fixed
        address: the host copy is dropped and the device copy recomputed now, in place."""
        self._occ_mean_host = None
        if "_occ_mean_dev" in self.__dict__:
            self.__dict__["_occ_mean_dev_version"] = None
            self._occ_mean_device()

    def _load_from_state_dict(self, *args, **kwargs):
        # a checkpoint's `occs` / `binaries` replace the grid: samples marched ahead are stale, so is min(alpha_thre, occs.mean())
        super()._load_from_state_dict(*args, **kwargs)
        self._bump_grid_version()
        self._invalidate_occ_mean()

    def check_deferred_overflow(self) -> None:
        """Deferred sampling never reads the marcher's slot-overflow flag on the critical path (the capacity is a proven bound for
        unit-length directions, and a violated bound truncates the ray instead of leaving its slots); this reads the sticky
        accumulator of all calls made so far -- one host synchronisation -- and raises if it is set.  The training path calls it
        at every occupancy refresh (``LSENeRFModel.update_occupancy_grid``: that step synchronises anyway)."""
        buf = self.__dict__.get("_overflow_acc")
        if buf is None:
            return
        bits = int(buf.item())
        if bits:
            buf.zero_()
            raise RuntimeError("deferred sampling: a ray produced more samples than LSEOccGridEstimator._cap_per_ray allows; its samples "
                               "were truncated to the capacity" + (" (that ray's direction is shorter than 1: the bound assumes "
                               "normalised directions -- normalise them or use the synchronising sampler)" if bits & 2 else ""))

    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None,
                 alpha_fn: Optional[Callable] = None, near_plane: float = 0.0, far_plane: float = 1e10,
                 t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0, jitter: Optional[Tensor] = None, return_packed: bool = False,
                 deferred: bool = False, premarched: Optional["Premarched"] = None):
        """Sampling with spatial skipping (not differentiable).  Returns (ray_indices, t_starts, t_ends); with
        ``return_packed`` also ``packed_info`` and ray_indices stays int32.

        ``deferred``: no host read-back of the sample counts (the reference's path synchronises four times here, the
        default path of this class once per stage).  Returns (ray_indices, t_starts, t_ends, packed_info, n_dev): the packed
        arrays have CAPACITY extent ``R * _cap_per_ray(...)``, ``n_dev`` (int64 [1], on the device) holds the number of valid
        leading entries, and every per-sample kernel downstream is handed ``n_dev`` (ops._call_n).  Values are those of the
        default path bit for bit; only the extents of the arrays differ.
        ``premarched`` (deferred only): the result of ``march_deferred`` for these rays and marching parameters -- the marcher
        is then not run again (it depends on the rays and the grid only, so it may run ahead of its step)."""
        if premarched is not None and not deferred:
            raise ValueError("premarched samples belong to the deferred (count-free) path")
        if deferred:
            return self._sampling_deferred(rays_o, rays_d, sigma_fn, alpha_fn, near_plane, far_plane, t_min, t_max,
                                           render_step_size, early_stop_eps, alpha_thre, stratified, cone_angle, jitter, premarched)
        # near / far planes clamped by t_min / t_max, stratified start offset u * step (one launch, bit-identical to the
        # reference's full_like / clamp / rand_like-multiply-add sequence)
        u = None
        if stratified:
            u = jitter if jitter is not None else torch.rand(rays_o.shape[0], dtype=torch.float32, device=rays_o.device)
        near_planes, far_planes = ops.ray_planes(rays_o.shape[0], rays_o.device, near_plane, far_plane, t_min, t_max, u,
                                                 render_step_size)
        ray_indices, t_starts, t_ends, packed_info = ops.traverse_grids(
            rays_o.contiguous(), rays_d.contiguous(), self._binaries_u8(), self.aabbs, near_planes, far_planes,
            render_step_size, cone_angle, max_span=self._max_span(near_plane, far_plane), fma_setup=self.traverse_fma)

        if self.after_march_hook is not None:   # e.g. finish the previous step's all-reduce + Adam (dist.GradPipeline)
            self.after_march_hook()

        # skip invisible space
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None) \
                and t_starts.shape[0] > 0:
            if self._occ_mean_host is None:
                self._occ_mean_host = self.occs.mean().item()
            alpha_thre = min(alpha_thre, self._occ_mean_host)
            if sigma_fn is not None:
                with torch.no_grad():
                    sigmas = sigma_fn(t_starts, t_ends, ray_indices)
                assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
                old_packed = packed_info
                ray_indices, t_starts, t_ends, packed_info, mask = ops.visibility_compact(
                    ray_indices, t_starts, t_ends, sigmas.contiguous(), packed_info, early_stop_eps, alpha_thre)
                on_cull = getattr(sigma_fn, "on_cull", None)      # the field keeps the survivors' pre-pass features
                if on_cull is not None and t_starts.shape[0] > 0:
                    on_cull(mask, old_packed, packed_info, ray_indices, t_starts, t_ends)
            else:
                with torch.no_grad():
                    alphas = alpha_fn(t_starts, t_ends, ray_indices)
                assert alphas.shape == t_starts.shape, "alphas must have shape of (N,)! Got {}".format(alphas.shape)
                ray_indices, t_starts, t_ends, packed_info, _ = ops.visibility_compact(
                    ray_indices, t_starts, t_ends, alphas.contiguous(), packed_info, early_stop_eps, alpha_thre,
                    from_alpha=True)
        if return_packed:
            return ray_indices, t_starts, t_ends, packed_info
        return ray_indices.long(), t_starts, t_ends

    @property
    def grid_version(self) -> int:
        """Counts the changes of ``binaries`` made through this class (refreshes, mark_all_occupied, dist.sync_grid): samples
        marched ahead of their step are valid while it has not moved."""
        return self.__dict__.get("_grid_version", 0)

    def _bump_grid_version(self) -> None:
        self.__dict__["_grid_version"] = self.grid_version + 1

    @torch.no_grad()
    def march_deferred(self, rays_o: Tensor, rays_d: Tensor, near_plane: float = 0.0, far_plane: float = 1e10,
                       t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                       stratified: bool = False, cone_angle: float = 0.0, jitter: Optional[Tensor] = None,
                       out: Optional["Premarched"] = None) -> "Premarched":
        """The marcher half of ``sampling(deferred=True)``: near / far planes, ray marching through the occupancy grid, packing.
        It reads the rays and the grid and nothing else -- no parameters -- so it can run ahead of the step that consumes its
        samples (on a side stream, behind the previous step's backward pass; lsenerf_amd.graph.GraphedTrainStep).  ``out``: a
        previous result for the same number of rays and the same marching parameters, overwritten in place."""
        u = None
        if stratified:
            u = jitter if jitter is not None else torch.rand(rays_o.shape[0], dtype=torch.float32, device=rays_o.device)
        near_planes, far_planes = ops.ray_planes(rays_o.shape[0], rays_o.device, near_plane, far_plane, t_min, t_max, u,
                                                 render_step_size)
        cap = self._cap_per_ray(near_plane, far_plane, render_step_size, cone_angle)
        res = ops.traverse_grids_deferred(
            rays_o.contiguous(), rays_d.contiguous(), self._binaries_u8(), self.aabbs, near_planes, far_planes,
            render_step_size, cone_angle, cap, out=None if out is None else tuple(out)[:6], overflow=self._overflow_flag(),
            fma_setup=self.traverse_fma)
        return Premarched(*res, self.grid_version, rays_o.shape[0])

    def _sampling_deferred(self, rays_o, rays_d, sigma_fn, alpha_fn, near_plane, far_plane, t_min, t_max, render_step_size,
                           early_stop_eps, alpha_thre, stratified, cone_angle, jitter, premarched=None):
        if premarched is None:
            premarched = self.march_deferred(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size, stratified,
                                             cone_angle, jitter)
        elif premarched.n_rays != rays_o.shape[0]:
            raise ValueError(f"premarched samples are for {premarched.n_rays} rays, {rays_o.shape[0]} given")
        # (a stale grid_version is the caller's to check before the launch: a captured step cannot)
        ray_indices, t_starts, t_ends, packed_info, n_dev = tuple(premarched)[:5]
        if self.after_march_hook is not None:
            self.after_march_hook()
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
            # `alpha_thre = min(alpha_thre, self.occs.mean().item())` of the reference, with the mean kept on the device
            # (refreshed in place whenever `occs` has been written): no read-back, and a captured launch follows grid refreshes
            alpha_cap = self._occ_mean_device()
            with torch.no_grad():
                if sigma_fn is not None:
                    try:
                        vals = sigma_fn(t_starts, t_ends, ray_indices, n_dev=n_dev)
                    except TypeError as e:
                        raise TypeError("deferred sampling needs a sigma_fn that accepts the device-side count (n_dev=...): "
                                        "LSEField.prepass_sigma_fn does") from e
                else:
                    vals = alpha_fn(t_starts, t_ends, ray_indices)
            assert vals.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(vals.shape)
            old_packed = packed_info
            ray_indices, t_starts, t_ends, packed_info, mask, n_dev = ops.visibility_compact_deferred(
                ray_indices, t_starts, t_ends, vals.contiguous(), packed_info, early_stop_eps, alpha_thre,
                from_alpha=sigma_fn is None, alpha_cap=alpha_cap)
            on_cull = getattr(sigma_fn, "on_cull", None) if sigma_fn is not None else None
            if on_cull is not None:
                on_cull(mask, old_packed, packed_info, ray_indices, t_starts, t_ends)
        return ray_indices, t_starts, t_ends, packed_info, n_dev

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16, generator: Optional[torch.Generator] = None) -> None:
        if not self.training:
            raise RuntimeError("You should only call this function only during training. "
                               "Please call _update() directly if you want to update the field during inference.")
        if step % n == 0 and self.training:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                         warmup_steps=warmup_steps, generator=generator)

    @torch.no_grad()
    def _get_all_cells(self) -> List[Tensor]:
        return [self.grid_indices[self.occs[l * self.cells_per_lvl:(l + 1) * self.cells_per_lvl] >= 0.0]
                for l in range(self.levels)]

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n: int, generator=None) -> List[Tensor]:
        out = []
        dev = self.occs.device
        for l in range(self.levels):
            uniform = torch.randint(self.cells_per_lvl, (n,), device=dev, generator=generator)
            lvl_occ = self.occs[l * self.cells_per_lvl + uniform]
            uniform = uniform[lvl_occ >= 0.0]
            occupied = torch.nonzero(self.binaries[l].flatten())[:, 0]
            if n < len(occupied):
                sel = torch.randint(len(occupied), (n,), device=dev, generator=generator)
                occupied = occupied[sel]
            out.append(torch.cat([uniform, occupied], dim=0))
        return out

    def _update_generator(self, step: int) -> torch.Generator:
        """The rank-independent stream of update ``step``: a fresh generator on the grid's device seeded from
        (update_seed, step) only -- never from the global RNG, which the trainer seeds per rank."""
        g = torch.Generator(device=self.occs.device)
        g.manual_seed((int(self.update_seed) * 1000003 + int(step)) & 0x7FFFFFFFFFFFFFFF)
        return g

    @torch.no_grad()
    def _update_samples(self, step: int, warmup_steps: int, generator: torch.Generator) -> List[Tuple[Tensor, Tensor]]:
        """Per level: (cell indices, one jittered position per cell in world coordinates).  Pure torch (runs on any
        device): this is the part of ``_update`` that consumes random numbers."""
        if step < warmup_steps:
            lvl_indices = self._get_all_cells()
        else:
            lvl_indices = self._sample_uniform_and_occupied_cells(self.cells_per_lvl // 4, generator)
        dev = self.occs.device
        out = []
        for lvl, indices in enumerate(lvl_indices):
            grid_coords = self.grid_coords[indices]
            u = torch.rand(grid_coords.shape, dtype=torch.float32, device=dev, generator=generator)
            x = (grid_coords + u) / self.resolution
            ab = self.aabbs[lvl]
            out.append((indices, ab[:3] + x * (ab[3:] - ab[:3])))
        return out

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01, ema_decay: float = 0.95,
                warmup_steps: int = 256, generator: Optional[torch.Generator] = None) -> None:
        """nerfacc ``OccGridEstimator._update``.  Index generation is torch plumbing; the EMA-max and the
        binarisation run in the HIP kernels (lse_occ_update_cells / lse_occ_binarize)."""
        if generator is None:
            generator = self._update_generator(step)
        for lvl, (indices, x) in enumerate(self._update_samples(step, warmup_steps, generator)):
            occ = occ_eval_fn(x).squeeze(-1)
            cell_ids = (lvl * self.cells_per_lvl + indices).contiguous()
            ops.occ_update_cells(self.occs, cell_ids, occ.contiguous().float(), ema_decay)
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre).reshape(1).contiguous()
        ops.occ_binarize(self.occs, thre, self._binaries_u8().view(-1))
        self._bump_grid_version()
        self._invalidate_occ_mean()
        hook = getattr(self, "after_update_hook", None)      # data parallel: dist.attach_grid_sync
        if hook is not None:
            hook()

    def mark_all_occupied(self, value: float = 1.0) -> None:
        """Benchmark helper: the 'grid fully occupied' regime of training steps < 256."""
        self.occs.fill_(value)
        self.binaries.fill_(True)
        self._bump_grid_version()
        self._invalidate_occ_mean()
