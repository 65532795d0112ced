"""Ray generation and camera-pose optimisers feeding the hot path (the first "next" row).

Mirrors, with the same names / argument meaning / shapes:
  * pose maths                  R:lse_nerf/interpolation_utils.py:14-246  (tangent <-> matrix, quaternions, slerp, pose spline)
  * ``EdCameras`` ray generation R:lse_nerf/lse_cameras.py:52-73, 257-586 (nerfstudio pinhole rays WITHOUT the half-pixel
                                 offset, pluggable ``get_c2w_fn``, time-interpolated poses, single intrinsics)
  * ``CameraOptimizer``          R:lse_nerf/ns_camera_optimizer.py:214-366 (per-camera SO3xR3 deltas)
  * ``SplineCameraOptimizer``    R:lse_nerf/ns_camera_optimizer.py:55-211  (control tangents -> quats -> slerp/lerp -> c2w;
                                 event camera = rgb camera @ dM with a learnable baseline scale; deblur timestamps)
  * ``CameraOptimizerConfig``    R:lse_nerf/ns_camera_optimizer.py:420-457
  * deblur ray tiling            R:lse_nerf/lse_ray_generator.py:103-147

These are O(rays) / O(control points) element-wise ops, so they stay in torch (device-agnostic, differentiable); what
they consume from the HIP path are d(loss)/d(origins) and d(loss)/d(directions), which ``lse_positions_bwd`` +
``lse_ray_grad_reduce`` + ``lse_ray_features_bwd`` produce.  Checked against scipy (Slerp / interp1d / Rotation) in
``tests/test_cameras_cpu.py`` -- the same oracle the reference's own self-tests use.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from .rays import RayBundle

EPS = 1e-6   # R:lse_nerf/utils.py:12


# ----------------------------------------------------------------------------------------------------
# pose maths
# ----------------------------------------------------------------------------------------------------
def _hat(v: Tensor) -> Tensor:
    """[N,3] -> [N,3,3] skew-symmetric matrices."""
    z = torch.zeros_like(v[:, 0])
    return torch.stack([torch.stack([z, -v[:, 2], v[:, 1]], -1),
                        torch.stack([v[:, 2], z, -v[:, 0]], -1),
                        torch.stack([-v[:, 1], v[:, 0], z], -1)], -2)


def hom_exp_map_SO3xR3(tangent_vector: Tensor) -> Tensor:
    """Tangent [N,6] = (translation, so(3) log-rotation) -> homogeneous [N,4,4] (Rodrigues with the same 1e-4 clamp on
    the squared angle as the reference).  R:lse_nerf/interpolation_utils.py:132-168."""
    log_rot = tangent_vector[:, 3:]
    theta = torch.clamp((log_rot * log_rot).sum(1), 1e-4).sqrt()
    K = _hat(log_rot)
    a = (theta.sin() / theta)[:, None, None]
    b = ((1.0 - theta.cos()) / (theta * theta))[:, None, None]
    R = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None] + a * K + b * (K @ K)
    out = torch.zeros(tangent_vector.shape[0], 4, 4, dtype=tangent_vector.dtype, device=tangent_vector.device)
    out[:, :3, :3] = R
    out[:, :3, 3] = tangent_vector[:, :3]
    out[:, 3, 3] = 1.0
    return out


def exp_map_SO3xR3(tangent_vector: Tensor) -> Tensor:
    """nerfstudio ``lie_groups.exp_map_SO3xR3`` -> [N,3,4]."""
    return hom_exp_map_SO3xR3(tangent_vector)[:, :3, :4]


def exp_map_SE3(tangent_vector: Tensor) -> Tensor:
    """SE(3) exponential: tangent [N,6] = (v, omega) -> [N,3,4] = [exp(omega^) | V v] with
    V = I + (1 - cos t)/t^2 omega^ + (t - sin t)/t^3 omega^2, t = |omega| (Taylor series below t = 1e-2).  What the reference's
    ``mode="SE3"`` evaluates through nerfstudio's lie_groups (R:lse_nerf/ns_camera_optimizer.py:276-277); checked against
    scipy.linalg.expm of the 4x4 twist in tests/test_cameras_cpu.py."""
    v, w = tangent_vector[:, :3], tangent_vector[:, 3:]
    t2 = (w * w).sum(-1)
    small = t2 < 1e-4
    # the sqrt only ever sees a safe argument: d sqrt / d t2 is infinite at 0, and autograd multiplies that infinity by the
    # zero the `where` hands back (-> NaN) -- exactly at the zeros CameraOptimizer(mode="SE3") starts from.  Every term
    # that reaches autograd on the small branch is a polynomial in t2 (nerfstudio's linalg.norm has subgradient 0 there).
    ts = torch.where(small, torch.ones_like(t2), t2).sqrt()   # safe denominators
    a = torch.where(small, 1 - t2 / 6, ts.sin() / ts)                          # sin t / t
    b = torch.where(small, 0.5 - t2 / 24, (1 - ts.cos()) / (ts * ts))          # (1 - cos t) / t^2
    c = torch.where(small, 1.0 / 6 - t2 / 120, (ts - ts.sin()) / (ts * ts * ts))   # (t - sin t) / t^3
    W = _hat(w)
    W2 = W @ W
    eye = torch.eye(3, dtype=w.dtype, device=w.device)[None]
    R = eye + a[:, None, None] * W + b[:, None, None] * W2
    V = eye + b[:, None, None] * W + c[:, None, None] * W2
    return torch.cat([R, V @ v[:, :, None]], dim=-1)


def matrix_to_tangent_vector(matrix: Tensor) -> Tensor:
    """4x4 (or 3x4) pose -> 6-vector (translation, axis*angle).  R:lse_nerf/interpolation_utils.py:14-53."""
    R = matrix[:3, :3]
    angle = torch.acos(torch.clamp((torch.trace(R) - 1) / 2, -1.0, 1.0))
    if angle.abs() < 1e-6:
        axis = torch.tensor([0.0, 0.0, 1.0], dtype=matrix.dtype)
    else:
        axis = torch.stack([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * torch.sin(angle))
    return torch.cat([matrix[:3, 3], axis * angle])


def exp_map_to_quat(v: Tensor) -> Tensor:
    """Rotation vectors [N,3] -> quaternions [N,4] (w,x,y,z).  R:lse_nerf/interpolation_utils.py:172-198."""
    theta = torch.norm(v, dim=1, keepdim=True)
    safe = torch.where(theta > 0, theta, torch.ones_like(theta))
    axis = torch.where(theta > 0, v / safe, torch.zeros_like(v))
    return torch.cat([torch.cos(theta / 2), axis * torch.sin(theta / 2)], dim=1)


def quat_to_rot_mat(q: Tensor) -> Tensor:
    """Quaternions [N,4] (w,x,y,z) -> [N,3,3].  R:lse_nerf/interpolation_utils.py:201-233."""
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def exp_map_to_quat_map(exp_map: Tensor) -> Tensor:
    """[N,6] (t, rotvec) -> [N,7] (t, quat)."""
    return torch.cat([exp_map[:, :3], exp_map_to_quat(exp_map[:, 3:])], dim=1)


def quat_map_to_mtx(quat_map: Tensor) -> Tensor:
    """[N,7] (t, quat) -> [N,3,4]."""
    return torch.cat([quat_to_rot_mat(quat_map[:, 3:]), quat_map[:, :3, None]], dim=2)


def slerp(v0: Tensor, v1: Tensor, t: Tensor) -> Tensor:
    """Batched quaternion slerp with the reference's edge-case handling (shortest arc; lerp when |dot| > 0.9995).
    v0, v1 [N,4]; t [N,1].  R:lse_nerf/interpolation_utils.py:56-99."""
    a = v0 / torch.norm(v0, dim=1, keepdim=True)
    b = v1 / torch.norm(v1, dim=1, keepdim=True)
    dot = (a * b).sum(1, keepdim=True).clamp(-1.0 + EPS, 1.0 - EPS)
    near = dot.abs().isnan() | (dot.abs() > 0.9995)
    flip = dot < 0
    b = torch.where(flip, -b, b)
    dot = torch.where(flip, -dot, dot)
    theta0 = torch.acos(dot)
    s0 = torch.sin(theta0)
    s0 = torch.where(s0 == 0, torch.ones_like(s0), s0)
    out = (torch.sin(theta0 - theta0 * t) / s0) * a + (torch.sin(theta0 * t) / s0) * b
    return torch.where(near.expand_as(out), (1 - t) * a + t * b, out)


def vectorized_generalized_interpolation(control_poses: Tensor, control_ts: Tensor, interp_ts: Tensor) -> Tensor:
    """Piecewise pose spline: lerp on translation, slerp on rotation between the two bracketing control points.
    control_poses [K,7] (t, quat); control_ts [K] ascending; interp_ts [N] -> [N,7].  R:...interpolation_utils.py:102-128."""
    control_poses, control_ts, interp_ts = control_poses.float(), control_ts.float(), interp_ts.float()
    idx = torch.searchsorted(control_ts, interp_ts, right=True)
    idx = torch.clamp(idx, 1, len(control_ts) - 1) - 1
    p0, p1 = control_poses[idx], control_poses[idx + 1]
    t0, t1 = control_ts[idx], control_ts[idx + 1]
    t = ((interp_ts - t0) / (t1 - t0)).unsqueeze(-1)
    return torch.cat([(1 - t) * p0[:, :3] + t * p1[:, :3], slerp(p0[:, 3:], p1[:, 3:], t)], dim=1)


# ----------------------------------------------------------------------------------------------------
# cameras
# ----------------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------------
# lens distortion (OpenCV radial-tangential model, as nerfstudio's camera_utils applies it at R:lse_nerf/lse_cameras.py:396-413)
# ----------------------------------------------------------------------------------------------------
def radial_and_tangential_distort(xy: Tensor, params: Tensor) -> Tensor:
    """Forward model: ideal normalised image coordinates [...,2] -> distorted ones; params [...,6] = (k1,k2,k3,k4,p1,p2)."""
    k1, k2, k3, k4, p1, p2 = [params[..., i] for i in range(6)]
    x, y = xy[..., 0], xy[..., 1]
    r = x * x + y * y
    d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
    return torch.stack([d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x), d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y)], -1)


def radial_and_tangential_undistort(coords: Tensor, distortion_params: Tensor, eps: float = 1e-3,
                                    max_iterations: int = 10) -> Tensor:
    """Inverse of the model above by Newton iterations on the 2x2 system (10 steps from the distorted point, a step is
    skipped where the Jacobian determinant is below ``eps``) -- the scheme the reference inherits from nerfstudio 0.3.2.
    coords [...,2]; distortion_params broadcastable [...,6]."""
    k1, k2, k3, k4, p1, p2 = [distortion_params[..., i] for i in range(6)]
    xd, yd = coords[..., 0], coords[..., 1]
    x, y = xd.clone(), yd.clone()
    for _ in range(max_iterations):
        r = x * x + y * y
        d = 1.0 + r * (k1 + r * (k2 + r * (k3 + r * k4)))
        fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
        fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
        d_r = k1 + r * (2.0 * k2 + r * (3.0 * k3 + r * 4.0 * k4))
        d_x, d_y = 2.0 * x * d_r, 2.0 * y * d_r
        fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
        fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
        fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
        fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
        det = fy_x * fx_y - fx_x * fy_y
        ok = det.abs() > eps
        safe = torch.where(ok, det, torch.ones_like(det))
        x = x + torch.where(ok, (fx * fy_y - fy * fx_y) / safe, torch.zeros_like(det))
        y = y + torch.where(ok, (fy * fx_x - fx * fy_x) / safe, torch.zeros_like(det))
    return torch.stack([x, y], -1)


class HardCamType:
    RGB = 0
    EVS = 1


class EdCameras:
    """Pinhole cameras with the reference's deviations from nerfstudio (R:lse_nerf/lse_cameras.py): no half-pixel
    offset in image coordinates (:69-73), pluggable ``get_c2w_fn`` incl. time-interpolated poses (:52-59, :349), one set
    of intrinsics for all cameras (:359-362)."""

    def __init__(self, camera_to_worlds: Tensor, fx: float, fy: float, cx: float, cy: float, width: int, height: int,
                 times: Optional[Tensor] = None, metadata: Optional[Dict[str, Tensor]] = None,
                 distortion_params: Optional[Tensor] = None):
        self.camera_to_worlds = camera_to_worlds.float()          # [C,3,4]
        # (k1,k2,k3,k4,p1,p2) for all cameras ([6]) or per camera ([C,6]); None / all zero = pinhole
        self.distortion_params = distortion_params.float() if distortion_params is not None else None
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.width, self.height = int(width), int(height)
        self.times = times.float().reshape(-1, 1) if times is not None else None
        self.metadata = metadata
        self.interpolator = None
        self.hard_cam_type = HardCamType.RGB
        self.get_c2w_fn: Callable = self.get_c2w

    def __len__(self):
        return self.camera_to_worlds.shape[0]

    def set_interpolator(self, interpolator) -> None:
        self.interpolator = interpolator

    def set_hard_cam_type(self, hard_cam_type: int) -> None:
        self.hard_cam_type = hard_cam_type

    def get_c2w(self, camera_indices: Tensor) -> Tensor:
        ci = camera_indices.reshape(-1).long()
        if self.interpolator is not None:
            return self.interpolator.get_fn_dic[self.hard_cam_type](self.times[ci])
        return self.camera_to_worlds[ci]

    def get_image_coords(self) -> Tensor:
        """[H,W,2] (y,x) integer pixel coordinates -- pixel_offset = 0 (R:lse_cameras.py:69-73)."""
        ys, xs = torch.meshgrid(torch.arange(self.height), torch.arange(self.width), indexing="ij")
        return torch.stack([ys, xs], -1).float()

    def generate_rays(self, camera_indices: Tensor, coords: Tensor, camera_opt_to_camera: Optional[Tensor] = None,
                      disable_distortion: bool = False) -> RayBundle:
        """camera_indices [R] (or [R,1]); coords [R,2] = (y, x) pixels.  Perspective model, OpenGL camera frame
        (x right, y up, looking down -z).  R:lse_nerf/lse_cameras.py:340-586."""
        c2w = self.get_c2w_fn(camera_indices)                      # [R,3,4]
        if camera_opt_to_camera is not None:                        # pose_utils.multiply(c2w, delta)
            R0, t0 = c2w[:, :3, :3], c2w[:, :3, 3:]
            c2w = torch.cat([R0 @ camera_opt_to_camera[:, :3, :3], R0 @ camera_opt_to_camera[:, :3, 3:] + t0], dim=-1)
        dev = c2w.device
        y, x = coords[..., 0].to(dev).float(), coords[..., 1].to(dev).float()

        def cam_xy(xx, yy):
            return torch.stack([(xx - self.cx) / self.fx, -(yy - self.cy) / self.fy], -1)
        xy = torch.stack([cam_xy(x, y), cam_xy(x + 1, y), cam_xy(x, y + 1)], 0)                    # [3,R,2]
        dist = self.distortion_params
        if dist is not None and not disable_distortion and bool((dist != 0).any()):                 # R:lse_cameras.py:396-413
            dist = dist.to(dev)
            if dist.dim() == 2:
                dist = dist[camera_indices.reshape(-1).long().to(dev)]
            xy = radial_and_tangential_undistort(xy, dist.expand(xy.shape[:-1] + (6,)))
        stack = torch.cat([xy, -torch.ones_like(xy[..., :1])], -1)                                  # [3,R,3]
        world = torch.sum(stack[..., None, :] * c2w[None, :, :3, :3], dim=-1)                       # rotate
        norm = torch.linalg.norm(world, dim=-1, keepdim=True)
        world = world / norm
        directions = world[0]
        dx = torch.sqrt(torch.sum((directions - world[1]) ** 2, dim=-1))
        dy = torch.sqrt(torch.sum((directions - world[2]) ** 2, dim=-1))
        ci = camera_indices.reshape(-1, 1).to(dev)
        meta = {"directions_norm": norm[0].detach()}
        if self.metadata is not None:
            for k, v in self.metadata.items():
                meta[k] = v.to(dev)[ci[:, 0]]
        return RayBundle(origins=c2w[:, :3, 3], directions=directions, pixel_area=(dx * dy)[..., None], camera_indices=ci,
                         times=self.times.to(dev)[ci[:, 0]] if self.times is not None else None, metadata=meta)


# ----------------------------------------------------------------------------------------------------
# optimisers
# ----------------------------------------------------------------------------------------------------
@dataclass
class CameraOptimizerConfig:
    """R:lse_nerf/ns_camera_optimizer.py:420-457."""
    mode: str = "off"                 # "off" | "SO3xR3" | "SE3"
    trans_l2_penalty: float = 1e-2
    rot_l2_penalty: float = 1e-3
    optim_type: str = "ns"            # "ns" | "spline" | "prevnext"
    control_pnt_factor: int = 1
    scheme: str = "active"            # "active" | "delayed"
    delay_cnt: int = 10000
    exp_t: float = 30000

    def __post_init__(self):
        if self.mode == "off":
            self.scheme = "active"
            self.delay_cnt = 1999999999

    def setup(self, **kwargs):
        return {"ns": CameraOptimizer, "spline": SplineCameraOptimizer,
                "prevnext": PrevNextCamOptimizer}[self.optim_type](self, **kwargs)       # R:ns_camera_optimizer.py:416-418


class _DelayedMode:
    def _init_scheme(self):
        self.is_on = True
        self.ori_mode = self.config.mode
        if self.config.scheme == "delayed":
            self.config.mode = "off"
            self.is_on = False

    def turn_on(self):
        self.config.mode = self.ori_mode
        self.is_on = True

    def update_mode(self, step):
        if not self.is_on and self.config.scheme == "delayed" and step > self.config.delay_cnt:
            self.turn_on()


class CameraOptimizer(nn.Module, _DelayedMode):
    """Per-camera learnable pose deltas.  R:lse_nerf/ns_camera_optimizer.py:214-366."""

    def __init__(self, config: CameraOptimizerConfig, num_cameras: int, device="cpu",
                 non_trainable_camera_indices: Optional[Tensor] = None, **kwargs) -> None:
        super().__init__()
        self.config, self.num_cameras, self.device = config, num_cameras, device
        self.non_trainable_camera_indices = non_trainable_camera_indices
        if config.mode in ("SO3xR3", "SE3"):
            self.pose_adjustment = nn.Parameter(torch.zeros((num_cameras, 6), device=device))
        else:
            assert config.mode == "off", config.mode
        self._init_scheme()

    def forward(self, indices: Tensor) -> Tensor:
        """-> [len(indices),3,4] transforms from optimised to given camera coordinates (identity when off)."""
        if self.config.mode == "off":
            return torch.eye(4, device=self.device)[None, :3, :4].tile(indices.shape[0], 1, 1)
        exp_map = exp_map_SE3 if self.config.mode == "SE3" else exp_map_SO3xR3
        out = exp_map(self.pose_adjustment[indices, :])
        if self.non_trainable_camera_indices is not None:
            frozen = torch.isin(indices.to(out.device), self.non_trainable_camera_indices.to(out.device))
            out = torch.where(frozen[:, None, None], torch.eye(4, device=out.device)[:3, :4].expand_as(out), out)
        return out

    def apply_to_raybundle(self, raybundle: RayBundle) -> None:
        """origins += t;  directions = R @ directions   (R:lse_nerf/ns_camera_optimizer.py:322-329)."""
        if self.config.mode != "off":
            shape = raybundle.origins.shape
            corr = self(raybundle.camera_indices.reshape(-1))
            o = raybundle.origins.reshape(-1, 3) + corr[:, :3, 3]
            d = torch.bmm(corr[:, :3, :3], raybundle.directions.reshape(-1, 3, 1)).squeeze(-1)
            raybundle.origins, raybundle.directions = o.reshape(shape), d.reshape(shape)

    def get_loss_dict(self, loss_dict: dict, prefix="") -> None:
        if self.config.mode != "off":
            loss_dict[f"{prefix}camera_opt_regularizer"] = (
                self.pose_adjustment[:, :3].norm(dim=-1).mean() * self.config.trans_l2_penalty
                + self.pose_adjustment[:, 3:].norm(dim=-1).mean() * self.config.rot_l2_penalty)

    def get_correction_matrices(self):
        return self(torch.arange(0, self.num_cameras).long())

    def get_metrics_dict(self, metrics_dict: dict, prefix: str = "") -> None:
        if self.config.mode != "off":
            metrics_dict[f"{prefix}camera_opt_translation"] = self.pose_adjustment[:, :3].norm()
            metrics_dict[f"{prefix}camera_opt_rotation"] = self.pose_adjustment[:, 3:].norm()

    def get_param_groups(self, param_groups: dict, prefix="") -> None:
        params = list(self.parameters())
        if self.config.mode != "off":
            assert len(params) > 0
            param_groups[f"{prefix}camera_opt"] = params
        else:
            assert len(params) == 0


class PrevNextCamOptimizer(nn.Module):
    """Two independent per-camera optimisers for the previous / next event-camera poses; ``apply_to_raybundle`` alternates
    between them call by call (the data manager corrects the prev bundle first, then the next one).
    R:lse_nerf/ns_camera_optimizer.py:368-414."""

    def __init__(self, config: CameraOptimizerConfig, num_cameras: int, device="cpu",
                 non_trainable_camera_indices: Optional[Tensor] = None, **kwargs) -> None:
        super().__init__()
        import copy
        # (the reference shares ONE config object between the two, so the delayed scheme's first constructor call already
        #  switches the shared mode to "off" and the second optimiser would be built without parameters; a copy per
        #  optimiser keeps both trainable, which is what the two param groups at :410-414 expect)
        self.prev_optim = CameraOptimizer(copy.copy(config), num_cameras, device, non_trainable_camera_indices, **kwargs)
        self.next_optim = CameraOptimizer(copy.copy(config), num_cameras, device, non_trainable_camera_indices, **kwargs)
        self.cnt_call = 0

    def turn_on(self):
        self.prev_optim.turn_on()
        self.next_optim.turn_on()

    def update_mode(self, step):
        self.prev_optim.update_mode(step)
        self.next_optim.update_mode(step)

    def forward(self, indices):
        assert 0, "not implemented"          # same behaviour as the reference (:389-390)

    def apply_to_raybundle(self, raybundle: RayBundle):
        (self.prev_optim if self.cnt_call % 2 == 0 else self.next_optim).apply_to_raybundle(raybundle)
        self.cnt_call = (self.cnt_call + 1) % 2

    def get_loss_dict(self, loss_dict):
        self.prev_optim.get_loss_dict(loss_dict, "prev_")
        self.next_optim.get_loss_dict(loss_dict, "next_")

    def get_metrics_dict(self, metrics_dict):
        self.prev_optim.get_metrics_dict(metrics_dict, "prev_")
        self.next_optim.get_metrics_dict(metrics_dict, "next_")

    def get_param_groups(self, param_groups):
        self.prev_optim.get_param_groups(param_groups, "prev_")
        self.next_optim.get_param_groups(param_groups, "next_")


class SplineCameraOptimizer(nn.Module, _DelayedMode):
    """Continuous-time camera trajectory: learnable control tangents on an SE(3)-like spline (lerp translation, slerp
    rotation).  R:lse_nerf/ns_camera_optimizer.py:55-211."""

    def __init__(self, config: CameraOptimizerConfig, num_cameras: int, device, cameras: EdCameras,
                 dM: Optional[Tensor] = None, **kwargs) -> None:
        super().__init__()
        self.config, self.device, self.cameras = config, device, cameras
        self.register_buffer("dM", dM)
        self.pnt_factor = config.control_pnt_factor
        self.build_control_pnts(cameras, n_factor=self.pnt_factor)
        self.scale = nn.Parameter(torch.ones(1))
        self.get_fn_dic = {HardCamType.RGB: self.get_rgb_cameras, HardCamType.EVS: self.get_evs_cameras}
        self.exp_t = config.exp_t
        self.n_deblur_rays = 4
        self._init_scheme()

    def build_control_pnts(self, cameras: EdCameras, n_factor=1):
        """Control points = the camera poses, plus (n_factor-1) scipy-interpolated poses inside every interval."""
        from scipy.interpolate import interp1d
        from scipy.spatial.transform import Rotation, Slerp
        c2w = cameras.camera_to_worlds.cpu().numpy()
        cam_ts = cameras.times.cpu().numpy().squeeze()
        rot_interp = Slerp(cam_ts, Rotation.from_matrix(c2w[:, :3, :3]))
        trans_interp = interp1d(cam_ts, c2w[:, :3, 3], axis=0, kind="linear")
        max_err = np.abs(rot_interp(cam_ts[0]).as_matrix() - c2w[0][:3, :3]).max()
        assert max_err < 1e-5, f"ERROR {max_err}, w2cs are mirror transforms"
        dts = (np.diff(cam_ts) / n_factor).reshape(-1, 1)
        steps = np.arange(0, n_factor, dtype=np.int32).reshape(1, -1)
        ctrl_ts = np.concatenate([(cam_ts.reshape(-1, 1)[:-1] + dts * steps).reshape(-1).astype(np.float32), cam_ts[-1:]])
        Rs, Ts = rot_interp(ctrl_ts).as_matrix(), trans_interp(ctrl_ts)
        poses = np.concatenate([Rs, Ts[..., None]], axis=-1)
        tang = torch.stack([matrix_to_tangent_vector(torch.from_numpy(M).float()) for M in poses])
        self.ctrl_tangents = nn.Parameter(tang.to(self.device), requires_grad=True)
        self.register_buffer("ctrl_ts", torch.tensor(ctrl_ts, dtype=torch.float32, device=self.device))
        self.register_buffer("orig_cam_ts", cameras.times.clone())

    def _maybe_no_grad(self, fn, *a):
        if self.config.mode == "off":
            with torch.no_grad():
                return fn(*a)
        return fn(*a)

    def get_rgb_cameras(self, times: Tensor) -> Tensor:
        """times [N] or [N,1] -> [N,3,4] camera-to-world."""
        def run(times):
            ts = torch.clip(times, self.ctrl_ts[0], self.ctrl_ts[-1]).reshape(-1)
            vec = vectorized_generalized_interpolation(exp_map_to_quat_map(self.ctrl_tangents), self.ctrl_ts, ts)
            return quat_map_to_mtx(vec)[:, :3, :4]
        return self._maybe_no_grad(run, times)

    def get_evs_cameras(self, times: Tensor) -> Tensor:
        """Event camera = rgb camera @ dM (4x4 relative pose); when optimising, its translation is scaled by ``scale``."""
        def run(times):
            dM = self.dM
            if self.config.mode != "off":
                dM = torch.cat((self.dM[:, :3], torch.cat((self.dM[:3, 3:4] * self.scale, self.dM[3:, 3:4]), dim=0)), dim=1)
            return self.get_rgb_cameras(times) @ dM
        return self._maybe_no_grad(run, times)

    def get_deblur_cameras(self, cam_ts: Tensor) -> Tensor:
        """cam_ts [n,1] exposure mid-times -> [4n,3,4]: poses at t - exp/2 + k*exp/3, k = 0..3 (BAD-NeRF style)."""
        def run(cam_ts):
            delta = self.exp_t / (self.n_deblur_rays - 1)
            steps = delta * torch.arange(self.n_deblur_rays, device=cam_ts.device)
            return self.get_rgb_cameras((cam_ts - self.exp_t / 2 + steps[None]).reshape(-1))
        return self._maybe_no_grad(run, cam_ts)

    def apply_to_raybundle(self, raybundle: RayBundle):
        return raybundle            # poses are already the optimised ones

    def get_param_groups(self, param_groups: dict) -> None:
        params = list(self.parameters())
        if self.config.mode != "off":
            assert len(params) > 0
            param_groups["camera_opt"] = params


def generate_deblur_rays(cameras: EdCameras, spline: SplineCameraOptimizer, camera_indices: Tensor, coords: Tensor
                         ) -> RayBundle:
    """``DeblurRayGenerator``: every sampled pixel becomes 4 rays from 4 virtual cameras across the exposure; the model
    averages their renders (R:lse_nerf/lse_ray_generator.py:103-147, R:lse_nerf/lsenerf.py:365-370)."""
    n = spline.n_deblur_rays
    ci = camera_indices.reshape(-1)
    old = cameras.get_c2w_fn
    cameras.get_c2w_fn = lambda idx: spline.get_deblur_cameras(cameras.times[idx.reshape(-1).long()][::n])
    try:
        rb = cameras.generate_rays(ci.repeat_interleave(n), coords.repeat_interleave(n, dim=0))
    finally:
        cameras.get_c2w_fn = old
    return rb


# ----------------------------------------------------------------------------------------------------
# ray generators (R:lse_nerf/lse_ray_generator.py; nerfstudio ``RayGenerator`` base restated)
# ----------------------------------------------------------------------------------------------------
class RayGenerator(nn.Module):
    """nerfstudio ``RayGenerator``: pixel indices ``[num_rays, 3] = (camera, row, col)`` -> ``RayBundle`` through the
    cameras and the pose optimiser (``pose_optimizer(camera_indices) -> [R,3,4]`` or ``None``)."""

    def __init__(self, cameras: EdCameras, pose_optimizer: Callable = lambda x: None) -> None:
        super().__init__()
        self.cameras = cameras
        self.pose_optimizer = pose_optimizer
        self.register_buffer("image_coords", cameras.get_image_coords(), persistent=False)

    def _split(self, ray_indices: Tensor):
        c, y, x = ray_indices[:, 0], ray_indices[:, 1], ray_indices[:, 2]
        return c, self.image_coords[y, x]

    def forward(self, ray_indices: Tensor) -> RayBundle:
        c, coords = self._split(ray_indices)
        return self.cameras.generate_rays(camera_indices=c.unsqueeze(-1), coords=coords,
                                          camera_opt_to_camera=self.pose_optimizer(c))


class ConsecRayGenerator(RayGenerator):
    """Event pixels seen from camera ``c`` and from the consecutive camera ``c + 1`` (R:lse_nerf/lse_ray_generator.py:36-68)."""

    def forward(self, ray_indices: Tensor) -> Tuple[RayBundle, RayBundle]:
        c, coords = self._split(ray_indices)
        ray_bundle_prev = self.cameras.generate_rays(camera_indices=c.unsqueeze(-1), coords=coords)
        ray_bundle_next = self.cameras.generate_rays(camera_indices=c.unsqueeze(-1) + 1, coords=coords)
        return ray_bundle_prev, ray_bundle_next


class PrevNextRayGenerator(RayGenerator):
    """The same pixels through two camera sets: poses at the start and at the end of every event window
    (R:lse_nerf/lse_ray_generator.py:71-100).  No pose optimiser here: PrevNextCamOptimizer corrects the bundles."""

    def __init__(self, prev_cameras: EdCameras, next_cameras: EdCameras, pose_optimizer: Callable = lambda x: None) -> None:
        super().__init__(prev_cameras, lambda x: None)
        self.prev_cameras, self.next_cameras = prev_cameras, next_cameras

    def forward(self, ray_indices: Tensor) -> Tuple[RayBundle, RayBundle]:
        c, coords = self._split(ray_indices)
        ray_bundle_prev = self.prev_cameras.generate_rays(camera_indices=c.unsqueeze(-1), coords=coords)
        ray_bundle_next = self.next_cameras.generate_rays(camera_indices=c.unsqueeze(-1), coords=coords)
        return ray_bundle_prev, ray_bundle_next


class DeblurRayGenerator(RayGenerator):
    """Every sampled pixel -> 4 rays from 4 virtual cameras across the exposure (R:lse_nerf/lse_ray_generator.py:103-147)."""

    def forward(self, ray_indices: Tensor) -> RayBundle:
        assert self.cameras.interpolator is not None, "requires interpolator!!"
        c, coords = self._split(ray_indices)
        return generate_deblur_rays(self.cameras, self.cameras.interpolator, c, coords)
