"""CPU tests of the f-1 row (ray generation + camera optimisers, lsenerf_amd/cameras.py) against scipy -- the same
oracle and the same < 1e-5 bars as the reference's own self-tests (R:lse_nerf/interpolation_utils.py:392-457,
R:lse_nerf/ns_camera_optimizer.py:505-537)."""
import numpy as np
import pytest
import torch
from scipy.interpolate import interp1d
from scipy.spatial.transform import Rotation, Slerp

from lsenerf_amd import cameras as cam


def gen_data(n_rnd=10, max_t=10.0, seed=0):
    rng = np.random.default_rng(seed)
    ts = np.sort(rng.uniform(0, max_t, n_rnd)).astype(np.float32)
    Rs = Rotation.random(n_rnd, random_state=seed).as_matrix()
    Ts = rng.uniform(-1, 1, (n_rnd, 3))
    c2w = np.concatenate([Rs, Ts[..., None]], -1).astype(np.float32)
    return c2w, ts


def scipy_spline(c2w, ts, q):
    R = Slerp(ts, Rotation.from_matrix(c2w[:, :3, :3]))(q).as_matrix()
    T = interp1d(ts, c2w[:, :3, 3], axis=0)(q)
    return np.concatenate([R, T[..., None]], -1)


def test_slerp_and_lerp_match_scipy():
    c2w, ts = gen_data()
    q = np.random.default_rng(1).uniform(ts[0], ts[-1], 200).astype(np.float32)
    tang = torch.stack([cam.matrix_to_tangent_vector(torch.from_numpy(M)) for M in c2w])
    out = cam.quat_map_to_mtx(cam.vectorized_generalized_interpolation(cam.exp_map_to_quat_map(tang), torch.from_numpy(ts),
                                                                       torch.from_numpy(q)))
    assert np.abs(out.numpy() - scipy_spline(c2w, ts, q)).max() < 1e-5


def test_tangent_and_quaternion_round_trips():
    c2w, _ = gen_data(seed=3)
    for M in c2w:
        M4 = torch.eye(4)
        M4[:3] = torch.from_numpy(M)
        t = cam.matrix_to_tangent_vector(M4)
        assert (cam.hom_exp_map_SO3xR3(t[None])[0] - M4).abs().max() < 1e-5
        assert (cam.quat_map_to_mtx(cam.exp_map_to_quat_map(t[None]))[0] - M4[:3]).abs().max() < 1e-5
    rv = torch.from_numpy(Rotation.random(20, random_state=1).as_rotvec()).float()
    assert np.abs(cam.quat_to_rot_mat(cam.exp_map_to_quat(rv)).numpy() - Rotation.from_rotvec(rv.numpy()).as_matrix()).max() < 1e-5
    z = cam.exp_map_to_quat(torch.zeros(2, 3))
    assert torch.equal(z, torch.tensor([[1.0, 0, 0, 0]] * 2))


def _cameras(n=10, seed=0):
    c2w, ts = gen_data(n, seed=seed)
    return cam.EdCameras(torch.from_numpy(c2w), 500.0, 510.0, 320.0, 240.0, 640, 480, times=torch.from_numpy(ts)), c2w, ts


@pytest.mark.parametrize("factor", [1, 3])
def test_spline_optimizer_matches_scipy_and_receives_gradients(factor):
    cams, c2w, ts = _cameras()
    opt = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="spline", control_pnt_factor=factor).setup(
        num_cameras=len(cams), device="cpu", cameras=cams, dM=torch.eye(4))
    assert len(opt.ctrl_ts) == (len(ts) - 1) * factor + 1
    q = torch.from_numpy(np.random.default_rng(2).uniform(ts[0], ts[-1], 64).astype(np.float32))
    M = opt.get_rgb_cameras(q)
    assert np.abs(M.detach().numpy() - scipy_spline(c2w, ts, q.numpy())).mean() < 1e-5
    M.sum().backward()
    assert opt.ctrl_tangents.grad is not None and float(opt.ctrl_tangents.grad.abs().sum()) > 0
    # times outside the trajectory clamp to its ends
    ends = opt.get_rgb_cameras(torch.tensor([ts[0] - 5.0, ts[-1] + 5.0]))
    assert np.abs(ends.detach().numpy() - c2w[[0, -1]]).max() < 1e-5


def test_event_camera_relative_pose_and_deblur_timestamps():
    cams, c2w, ts = _cameras()
    dM = torch.eye(4)
    dM[:3, 3] = torch.tensor([0.1, -0.02, 0.03])
    cfg = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="spline", exp_t=0.3)
    opt = cfg.setup(num_cameras=len(cams), device="cpu", cameras=cams, dM=dM)
    with torch.no_grad():
        opt.scale.fill_(2.0)
    t = torch.tensor([ts[3]])
    rgb, evs = opt.get_rgb_cameras(t)[0], opt.get_evs_cameras(t)[0]
    expect = rgb[:, :3] @ (2.0 * dM[:3, 3]) + rgb[:, 3]
    assert torch.allclose(evs[:, 3], expect, atol=1e-6) and torch.allclose(evs[:, :3], rgb[:, :3], atol=1e-6)
    evs.sum().backward()
    assert opt.scale.grad is not None                                   # the baseline scale is learnable
    mid = torch.tensor([[ts[4]], [ts[5]]])
    D = opt.get_deblur_cameras(mid)
    assert D.shape == (8, 3, 4)
    want = torch.cat([opt.get_rgb_cameras(m - 0.15 + 0.1 * torch.arange(4)) for m in mid.reshape(-1)])
    assert torch.allclose(D, want, atol=1e-6)
    # mode "off": same numbers, no graph
    opt.config.mode = "off"
    assert not opt.get_rgb_cameras(t).requires_grad and torch.equal(opt.get_evs_cameras(t)[0, :, 3], (rgb @ dM)[:, 3].detach())


def test_camera_optimizer_deltas_regulariser_and_schemes():
    from lsenerf_amd import RayBundle
    opt = cam.CameraOptimizerConfig(mode="SO3xR3").setup(num_cameras=5, device="cpu")
    o = torch.rand(7, 3)
    d = torch.nn.functional.normalize(torch.randn(7, 3), dim=-1)
    idx = torch.tensor([0, 1, 2, 3, 4, 0, 1])[:, None]
    rb = RayBundle(o.clone(), d.clone(), camera_indices=idx)
    opt.apply_to_raybundle(rb)
    assert torch.allclose(rb.origins, o) and torch.allclose(rb.directions, d, atol=1e-6)       # zero deltas = identity
    with torch.no_grad():
        opt.pose_adjustment[1] = torch.tensor([0.1, 0.2, -0.3, 0.0, 0.0, np.pi / 2])
    rb = RayBundle(o.clone(), d.clone(), camera_indices=idx)
    opt.apply_to_raybundle(rb)
    Rz = torch.tensor(Rotation.from_rotvec([0, 0, np.pi / 2]).as_matrix(), dtype=torch.float32)
    assert torch.allclose(rb.origins[1], o[1] + torch.tensor([0.1, 0.2, -0.3]), atol=1e-6)
    assert torch.allclose(rb.directions[6], Rz @ d[6], atol=1e-5) and torch.allclose(rb.directions[0], d[0], atol=1e-6)
    (rb.origins.sum() + rb.directions.sum()).backward()
    assert float(opt.pose_adjustment.grad[1].abs().sum()) > 0
    ld, md, pg = {}, {}, {}
    opt.get_loss_dict(ld); opt.get_metrics_dict(md); opt.get_param_groups(pg)
    assert abs(float(ld["camera_opt_regularizer"]) - (np.sqrt(0.14) / 5 * 1e-2 + (np.pi / 2) / 5 * 1e-3)) < 1e-6
    assert set(pg) == {"camera_opt"} and "camera_opt_rotation" in md
    off = cam.CameraOptimizerConfig(mode="off").setup(num_cameras=5, device="cpu")
    assert off(torch.tensor([0, 3])).shape == (2, 3, 4) and len(list(off.parameters())) == 0
    delayed = cam.CameraOptimizerConfig(mode="SO3xR3", scheme="delayed", delay_cnt=10).setup(num_cameras=2, device="cpu")
    assert delayed.config.mode == "off" and not delayed.is_on
    delayed.update_mode(5); assert delayed.config.mode == "off"
    delayed.update_mode(11); assert delayed.config.mode == "SO3xR3" and delayed.is_on


def test_generate_rays_pinhole_without_half_pixel_offset():
    cams, c2w, ts = _cameras()
    ci = torch.tensor([2, 2, 5])
    coords = torch.tensor([[240.0, 320.0], [0.0, 0.0], [479.0, 639.0]])          # (y, x); first = principal point
    rb = cams.generate_rays(ci, coords)
    R2 = torch.from_numpy(c2w[2][:3, :3])
    assert torch.allclose(rb.directions[0], -R2[:, 2], atol=1e-6)                 # looks down -z, no 0.5 offset
    v = torch.tensor([(0 - 320.0) / 500.0, -(0 - 240.0) / 510.0, -1.0])
    assert torch.allclose(rb.directions[1], (R2 @ v) / v.norm(), atol=1e-6)
    assert torch.allclose(rb.origins[2], torch.from_numpy(c2w[5][:3, 3])) and rb.pixel_area.shape == (3, 1)
    assert torch.allclose(rb.metadata["directions_norm"][1, 0], v.norm(), atol=1e-6) and float(rb.pixel_area.min()) > 0
    assert torch.allclose(rb.times[:, 0], torch.from_numpy(ts)[ci])
    assert cams.get_image_coords()[0, 0].tolist() == [0.0, 0.0]
    # pose deltas compose on the right (pose_utils.multiply)
    delta = cam.exp_map_SO3xR3(torch.tensor([[0.0, 0, 0.5, 0, 0, 0]]).repeat(3, 1))
    rb2 = cams.generate_rays(ci, coords, camera_opt_to_camera=delta)
    assert torch.allclose(rb2.origins[0], rb.origins[0] + 0.5 * R2[:, 2], atol=1e-6)


def test_deblur_ray_tiling_and_time_interpolated_poses():
    cams, c2w, ts = _cameras()
    spl = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="spline", exp_t=0.2).setup(
        num_cameras=len(cams), device="cpu", cameras=cams, dM=torch.eye(4))
    ci = torch.tensor([3, 6])
    coords = torch.tensor([[100.0, 200.0], [50.0, 60.0]])
    rb = cam.generate_deblur_rays(cams, spl, ci, coords)
    assert len(rb) == 8 and rb.camera_indices.reshape(-1).tolist() == [3] * 4 + [6] * 4
    want = spl.get_deblur_cameras(torch.from_numpy(ts)[ci][:, None])
    assert torch.allclose(rb.origins, want[:, :3, 3], atol=1e-6)
    rb.origins.sum().backward()
    assert float(spl.ctrl_tangents.grad.abs().sum()) > 0
    # interpolator plugged into the cameras: poses come from the spline at the cameras' own times
    cams.set_interpolator(spl)
    r = cams.generate_rays(torch.tensor([4]), torch.tensor([[240.0, 320.0]]))
    assert torch.allclose(r.origins[0], torch.from_numpy(c2w[4][:3, 3]), atol=1e-5)


def test_se3_exponential_matches_scipy_expm_and_drives_the_optimizer():
    """mode="SE3" (R:lse_nerf/ns_camera_optimizer.py:276-277): [R | V v] of the twist (v, omega), against scipy's matrix
    exponential of the 4x4 twist -- also in the small-angle branch -- and through CameraOptimizer."""
    from scipy.linalg import expm
    from lsenerf_amd import RayBundle
    rng = np.random.default_rng(3)
    tang = rng.normal(size=(40, 6)) * np.array([1, 1, 1, 1.5, 1.5, 1.5])
    tang[:8, 3:] *= 1e-3                                                     # Taylor branch (|omega| < 1e-2)
    tang[8, 3:] = 0.0
    got = cam.exp_map_SE3(torch.tensor(tang, dtype=torch.float64)).numpy()
    for k in range(40):
        v, w = tang[k, :3], tang[k, 3:]
        twist = np.zeros((4, 4))
        twist[:3, :3] = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        twist[:3, 3] = v
        assert np.abs(got[k] - expm(twist)[:3, :4]).max() < 1e-9, k
    got32 = cam.exp_map_SE3(torch.tensor(tang, dtype=torch.float32)).numpy()
    assert np.abs(got32 - got).max() < 1e-5                                  # the reference's own bar
    opt = cam.CameraOptimizerConfig(mode="SE3").setup(num_cameras=3, device="cpu")
    with torch.no_grad():
        opt.pose_adjustment[2] = torch.tensor([0.3, -0.1, 0.2, 0.0, 0.0, np.pi / 2])
    o, d = torch.zeros(2, 3), torch.tensor([[1.0, 0, 0], [1.0, 0, 0]])
    rb = RayBundle(o.clone(), d.clone(), camera_indices=torch.tensor([[0], [2]]))
    opt.apply_to_raybundle(rb)
    assert torch.allclose(rb.directions[0], d[0]) and torch.allclose(rb.directions[1], torch.tensor([0.0, 1, 0]), atol=1e-6)
    twist = np.zeros((4, 4)); twist[0, 1], twist[1, 0] = -np.pi / 2, np.pi / 2; twist[:3, 3] = [0.3, -0.1, 0.2]
    assert np.abs(rb.origins[1].detach().numpy() - expm(twist)[:3, 3]).max() < 1e-6    # translation couples with rotation
    rb.origins.sum().backward()
    assert float(opt.pose_adjustment.grad[2, 3:].abs().sum()) > 0


def test_se3_exponential_has_finite_exact_gradients_at_the_zero_tangent():
    """CameraOptimizer(mode="SE3") starts from pose_adjustment = 0: the gradient there must be finite and equal the analytic
    Jacobian of exp at the identity -- d[R | t]/d(v, omega) = [omega^ generators | I for v, 0.5 * omega^ v terms] -- and the
    value/gradient must be continuous across the Taylor switch at |omega| = 1e-2."""
    p = torch.zeros(2, 6, dtype=torch.float64, requires_grad=True)
    m = cam.exp_map_SE3(p)
    assert torch.equal(m[0].detach(), torch.eye(3, 4, dtype=torch.float64))
    # d(sum of c_ij * M_ij) for a fixed random c: analytic at identity: dM/dv_k = e_k in the last column; dM/dw_k = hat(e_k) in R
    c = torch.tensor(np.random.default_rng(5).normal(size=(3, 4)))
    (m[0] * c).sum().backward()
    g = p.grad[0]
    assert torch.isfinite(p.grad).all()
    hat = lambda k: np.array([[0, 0, 0], [0, 0, -1], [0, 1, 0]]) if k == 0 else (
        np.array([[0, 0, 1], [0, 0, 0], [-1, 0, 0]]) if k == 1 else np.array([[0, -1, 0], [1, 0, 0], [0, 0, 0]]))
    want = [float(c[k, 3]) for k in range(3)] + [float((c[:, :3].numpy() * hat(k)).sum()) for k in range(3)]
    assert np.abs(g.numpy() - np.array(want)).max() < 1e-12
    # float32 as the optimiser holds it, through CameraOptimizer: one Adam step from zeros stays finite
    opt = cam.CameraOptimizerConfig(mode="SE3").setup(num_cameras=2, device="cpu")
    from lsenerf_amd import RayBundle
    rb = RayBundle(torch.zeros(2, 3), torch.tensor([[1.0, 0, 0], [0, 1.0, 0]]), camera_indices=torch.tensor([[0], [1]]))
    opt.apply_to_raybundle(rb)
    (rb.origins.sum() + (rb.directions * torch.tensor([0.3, -0.2, 0.5])).sum()).backward()
    assert torch.isfinite(opt.pose_adjustment.grad).all() and float(opt.pose_adjustment.grad[:, 3:].abs().sum()) > 0
    # continuity across the branch switch
    for scale in (0.999e-2, 1.001e-2):
        q = (torch.tensor([[0.2, -0.1, 0.3, 0.6, -0.64, 0.48]], dtype=torch.float64) * torch.tensor([1, 1, 1, scale, scale, scale])
             ).requires_grad_(True)
        mm = cam.exp_map_SE3(q)
        (mm[0] * c).sum().backward()
        if scale < 1e-2:
            lo_v, lo_g = mm.detach().clone(), q.grad.clone()
        else:
            assert (mm.detach() - lo_v).abs().max() < 1e-4 and (q.grad - lo_g).abs().max() < 1e-4


def test_prev_next_optimizer_alternates_and_prefixes():
    """R:lse_nerf/ns_camera_optimizer.py:368-414."""
    from lsenerf_amd import RayBundle
    pn = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="prevnext").setup(num_cameras=4, device="cpu")
    assert isinstance(pn, cam.PrevNextCamOptimizer)
    with torch.no_grad():
        pn.prev_optim.pose_adjustment[:, 0] = 1.0
        pn.next_optim.pose_adjustment[:, 1] = 2.0
    mk = lambda: RayBundle(torch.zeros(3, 3), torch.tensor([[0.0, 0, 1]] * 3), camera_indices=torch.tensor([[0], [1], [3]]))
    a, b, c = mk(), mk(), mk()
    pn.apply_to_raybundle(a); pn.apply_to_raybundle(b); pn.apply_to_raybundle(c)
    assert a.origins[:, 0].tolist() == [1.0] * 3 and a.origins[:, 1].tolist() == [0.0] * 3          # prev
    assert b.origins[:, 1].tolist() == [2.0] * 3 and b.origins[:, 0].tolist() == [0.0] * 3          # next
    assert torch.equal(c.origins, a.origins)                                                         # prev again
    ld, md, pg = {}, {}, {}
    pn.get_loss_dict(ld); pn.get_metrics_dict(md); pn.get_param_groups(pg)
    assert set(ld) == {"prev_camera_opt_regularizer", "next_camera_opt_regularizer"}
    assert set(pg) == {"prev_camera_opt", "next_camera_opt"} and "next_camera_opt_translation" in md
    with pytest.raises(AssertionError):
        pn(torch.tensor([0]))
    delayed = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="prevnext", scheme="delayed", delay_cnt=3).setup(
        num_cameras=2, device="cpu")
    assert len(list(delayed.parameters())) == 2 and not delayed.prev_optim.is_on
    delayed.update_mode(4)
    assert delayed.prev_optim.is_on and delayed.next_optim.is_on


def test_event_ray_generators():
    """R:lse_nerf/lse_ray_generator.py:36-100: the same pixels seen from (c, c+1) or from a prev / next camera set."""
    cams, c2w, ts = _cameras()
    idx = torch.tensor([[0, 10, 20], [4, 479, 639], [2, 240, 320]])
    gen = cam.RayGenerator(cams, cam.CameraOptimizerConfig(mode="SO3xR3").setup(num_cameras=len(cams), device="cpu"))
    rb = gen(idx)
    want = cams.generate_rays(idx[:, 0], torch.tensor([[10.0, 20.0], [479.0, 639.0], [240.0, 320.0]]))
    assert torch.allclose(rb.origins, want.origins) and torch.allclose(rb.directions, want.directions, atol=1e-6)
    prev, nxt = cam.ConsecRayGenerator(cams)(idx)
    assert torch.allclose(prev.origins, torch.from_numpy(c2w[[0, 4, 2], :3, 3]))
    assert torch.allclose(nxt.origins, torch.from_numpy(c2w[[1, 5, 3], :3, 3]))
    assert prev.camera_indices.reshape(-1).tolist() == [0, 4, 2] and nxt.camera_indices.reshape(-1).tolist() == [1, 5, 3]
    # same pixel, same intrinsics: in camera coordinates the directions coincide
    R0, R1 = torch.from_numpy(c2w[0][:3, :3]), torch.from_numpy(c2w[1][:3, :3])
    assert torch.allclose(R0.T @ prev.directions[0], R1.T @ nxt.directions[0], atol=1e-5)
    c2w_b, _ = gen_data(8, seed=5)
    cams_b = cam.EdCameras(torch.from_numpy(c2w_b), cams.fx, cams.fy, cams.cx, cams.cy, cams.width, cams.height,
                           times=torch.from_numpy(ts))
    p2, n2 = cam.PrevNextRayGenerator(cams, cams_b)(idx)
    assert torch.allclose(p2.origins, prev.origins) and torch.allclose(n2.origins, torch.from_numpy(c2w_b[[0, 4, 2], :3, 3]))
    assert n2.camera_indices.reshape(-1).tolist() == [0, 4, 2]
    # deblur generator = the tiling function behind the nn.Module interface
    spl = cam.CameraOptimizerConfig(mode="SO3xR3", optim_type="spline", exp_t=0.2).setup(
        num_cameras=len(cams), device="cpu", cameras=cams, dM=torch.eye(4))
    cams.set_interpolator(spl)
    rb4 = cam.DeblurRayGenerator(cams)(idx)
    assert len(rb4) == 12 and rb4.camera_indices.reshape(-1).tolist() == [0] * 4 + [4] * 4 + [2] * 4


def test_lens_distortion_is_applied_and_inverts_the_forward_model():
    """R:lse_nerf/lse_cameras.py:396-413: rays of a camera with (k1,k2,k3,k4,p1,p2) != 0 go through the iterative
    radial-tangential undistortion.  Independent check: distort ideal points with the closed-form forward model, undistort
    them, recover the originals; and a distorted camera's ray through a distorted pixel equals the pinhole ray through the
    ideal pixel."""
    g = torch.Generator().manual_seed(0)
    ideal = (torch.rand(500, 2, generator=g, dtype=torch.float64) - 0.5) * 1.2
    params = torch.tensor([-0.12, 0.05, -0.01, 0.0, 0.002, -0.003], dtype=torch.float64)
    distorted = cam.radial_and_tangential_distort(ideal, params.expand(500, 6))
    assert float((distorted - ideal).abs().max()) > 1e-3
    back = cam.radial_and_tangential_undistort(distorted, params.expand(500, 6))
    assert float((back - ideal).abs().max()) < 1e-9
    # through the cameras
    cams, c2w, ts = _cameras()
    dcams = cam.EdCameras(torch.from_numpy(c2w), cams.fx, cams.fy, cams.cx, cams.cy, cams.width, cams.height,
                          times=torch.from_numpy(ts), distortion_params=params.float())
    px_ideal = torch.tensor([[100.0, 500.0], [400.0, 50.0], [240.0, 320.0]])           # (y, x)
    xy_i = torch.stack([(px_ideal[:, 1] - cams.cx) / cams.fx, -(px_ideal[:, 0] - cams.cy) / cams.fy], -1)
    xy_d = cam.radial_and_tangential_distort(xy_i.double(), params.expand(3, 6)).float()
    px_dist = torch.stack([cams.cy - xy_d[:, 1] * cams.fy, xy_d[:, 0] * cams.fx + cams.cx], -1)
    ci = torch.tensor([1, 4, 7])
    want = cams.generate_rays(ci, px_ideal)
    got = dcams.generate_rays(ci, px_dist)
    assert torch.allclose(got.directions, want.directions, atol=2e-6) and torch.allclose(got.origins, want.origins)
    raw = dcams.generate_rays(ci, px_dist, disable_distortion=True)
    assert float((raw.directions - want.directions).abs().max()) > 1e-4
    assert torch.allclose(cams.generate_rays(ci, px_ideal).directions,
                          cam.EdCameras(torch.from_numpy(c2w), cams.fx, cams.fy, cams.cx, cams.cy, cams.width, cams.height,
                                        distortion_params=torch.zeros(6)).generate_rays(ci, px_ideal).directions)
