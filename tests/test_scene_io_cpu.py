"""LSENeRF-data-formatter scene layout reader on a synthetic scene written in that layout."""
import json
import os

import numpy as np
import pytest
import torch


def _rot(seed):
    from scipy.spatial.transform import Rotation
    return Rotation.random(random_state=seed).as_matrix()


def _write_cam(path, R, pos, t=None, f=300.0, size=(8, 6), dist=0.0):
    d = {"orientation": R.tolist(), "position": pos.tolist(), "focal_length": f, "principal_point": [size[0] / 2, size[1] / 2],
         "image_size": list(size), "radial_distortion": [dist, 0.0, 0.0], "tangential_distortion": [0.0, 0.0]}
    if t is not None:
        d["t"] = t
    with open(path, "w") as fh:
        json.dump(d, fh)


@pytest.fixture()
def scene(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    root = tmp_path / "scene"
    n = 9
    rel_R, rel_T = _rot(100), np.array([0.1, -0.02, 0.03])
    (root).mkdir()
    with open(root / "rel_cam.json", "w") as fh:
        json.dump({"R": rel_R.tolist(), "T": rel_T.tolist()}, fh)
    col = root / "colcam_set"
    (col / "camera").mkdir(parents=True)
    (col / "rgb" / "1x").mkdir(parents=True)
    Rs, Ps = [], []
    for i in range(n):
        R, p = _rot(i), rng.normal(size=3)
        Rs.append(R); Ps.append(p)
        _write_cam(col / "camera" / f"{i:05d}.json", R, p, t=float(i) * 0.5)
        Image.fromarray(rng.integers(0, 255, (6, 8, 3), dtype=np.uint8)).save(col / "rgb" / "1x" / f"{i:05d}.png")
    with open(col / "dataset.json", "w") as fh:
        json.dump({"train_ids": ["0", "2", "4", "6", "8"], "val_ids": ["1", "3", "5", "7"]}, fh)
    meta = {f"{i:05d}": {"appearance_id": i % 3} for i in range(n)}
    meta["colmap_scale"] = 2.0
    with open(col / "metadata.json", "w") as fh:
        json.dump(meta, fh)
    ec = root / "ecam_set"
    (ec / "prev_camera").mkdir(parents=True)
    (ec / "next_camera").mkdir(parents=True)
    (ec / "camera").mkdir(parents=True)
    (ec / "eimgs").mkdir(parents=True)
    m = 6
    for i in range(m):
        _write_cam(ec / "camera" / f"{i:05d}.json", _rot(50 + i), rng.normal(size=3), t=float(i))
        _write_cam(ec / "prev_camera" / f"{i:05d}.json", _rot(50 + i), rng.normal(size=3), t=float(i))
        _write_cam(ec / "next_camera" / f"{i:05d}.json", _rot(60 + i), rng.normal(size=3), t=float(i) + 0.5)
    np.save(ec / "eimgs" / "eimgs_1x.npy", rng.integers(-3, 4, (m, 6, 8)).astype(np.int8))
    with open(ec / "dataset.json", "w") as fh:
        json.dump({"train_ids": [str(i) for i in range(m)]}, fh)
    with open(ec / "metadata.json", "w") as fh:
        json.dump({f"{i:05d}": {"appearance_id": 7 + i} for i in range(m)}, fh)
    with open(ec / "scene.json", "w") as fh:
        json.dump({"e_thresh": 0.35}, fh)
    return {"root": str(root), "Rs": Rs, "Ps": Ps, "rel": (rel_R, rel_T)}


def test_cv_to_gl_conversion_is_inverse_with_flipped_axes():
    from lsenerf_amd.scene_io import cv_w2c_to_gl_c2w
    R, p = _rot(3), np.array([0.3, -1.2, 2.0])
    w2c = np.eye(4); w2c[:3, :3] = R; w2c[:3, 3] = -R @ p
    c2w = cv_w2c_to_gl_c2w(w2c)
    ref = np.linalg.inv(w2c) @ np.diag([1.0, -1.0, -1.0, 1.0])       # independent formulation
    assert np.allclose(c2w, ref, atol=1e-12) and np.allclose(c2w[:3, 3], p)


def test_color_reader_splits_cameras_and_relative_pose(scene):
    from lsenerf_amd.scene_io import ColorDataset, ColorSceneReader, cv_w2c_to_gl_c2w
    rd = ColorSceneReader(os.path.join(scene["root"], "colcam_set"), scale_factor=0.5, scene_scale=2.0)
    out = rd.outputs("train")
    assert out.data_idxs == [0, 2, 4, 6]                     # id 8 = last frame is dropped (idx < n_images - 1)
    assert out.appearance_ids == [0, 2, 1, 0] and len(out.image_filenames) == 4
    assert torch.equal(out.scene_aabb, torch.tensor([[-2.0, -2, -2], [2, 2, 2]]))
    cams = out.cameras
    assert len(cams) == 4 and (cams.width, cams.height) == (8, 6) and cams.fx == 300.0 and cams.distortion_params is None
    assert torch.allclose(cams.times.flatten(), torch.tensor([0.0, 1.0, 2.0, 3.0]))
    for k, i in enumerate(out.data_idxs):
        R, p = scene["Rs"][i], scene["Ps"][i]
        assert np.allclose(cams.camera_to_worlds[k, :, 3].numpy(), p * 0.5, atol=1e-6)          # scaled position
        assert np.allclose(cams.camera_to_worlds[k, :, :3].numpy(), (R.T @ np.diag([1.0, -1, -1])), atol=1e-6)
    # dM: c2w_evs(gl) = c2w_rgb(gl) @ dM for every camera
    rel_R, rel_T = scene["rel"]
    d_cv = np.eye(4); d_cv[:3, :3] = rel_R; d_cv[:3, 3] = rel_T * 2.0
    for i in (0, 4):
        w2c = np.eye(4); w2c[:3, :3] = scene["Rs"][i]; w2c[:3, 3] = -scene["Rs"][i] @ scene["Ps"][i]
        rgb, evs = cv_w2c_to_gl_c2w(w2c), cv_w2c_to_gl_c2w(d_cv @ w2c)
        rgb[:3, 3] *= 0.5; evs[:3, 3] *= 0.5
        assert np.allclose(rgb @ out.dM.numpy().astype(np.float64), evs, atol=1e-5)
    val = rd.outputs("val")
    assert val.data_idxs == [1, 3, 5, 7]
    ev = ColorSceneReader(os.path.join(scene["root"], "colcam_set"), is_eval=True).outputs("train")
    assert ev.data_idxs == [1, 3, 5, 7]                      # frozen-NeRF evaluation optimises on the val cameras
    ds = ColorDataset(out)
    d = ds.get_data(1)
    assert d["image"].shape == (6, 8, 3) and d["image"].dtype == torch.float32 and float(d["image"].max()) <= 1.0
    assert d["appearance_id"] == 2 and rd.train_ids() == [0, 2, 4, 6] and rd.max_appearance_id() == 3
    assert len(rd.all_cameras()) == 8 and torch.allclose(rd.train_times(), torch.tensor([0.0, 1.0, 2.0, 3.0]))


def test_event_reader_and_dataset(scene):
    from lsenerf_amd.cameras import HardCamType
    from lsenerf_amd.scene_io import EventFrameDataset, EventSceneReader
    rd = EventSceneReader(os.path.join(scene["root"], "ecam_set"))
    out = rd.outputs()
    assert out.events.shape == (6, 6, 8, 1) and out.e_thresh == 0.35 and out.appearance_ids == [7, 8, 9, 10, 11, 12]
    assert out.prev_cameras is not None and len(out.next_cameras) == 6 and out.cameras.hard_cam_type == HardCamType.EVS
    assert torch.allclose(out.next_cameras.times.flatten(), torch.arange(6).float() + 0.5)
    ds = EventFrameDataset(out)
    raw = np.load(os.path.join(scene["root"], "ecam_set", "eimgs", "eimgs_1x.npy"))
    assert torch.allclose(ds.get_image(2), torch.from_numpy(raw[2][..., None].astype(np.float32)) * 0.35)
    assert float(ds.get_data(0)["e_thresh"]) == pytest.approx(0.35)
    with pytest.raises(AssertionError):
        ds.get_numpy_image(0)
    # overrides: explicit threshold wins over scene.json; "none" strings are None; decam_set forces 1
    assert EventSceneReader(os.path.join(scene["root"], "ecam_set"), e_thresh="0.5").outputs().e_thresh == 0.5
    assert EventSceneReader(os.path.join(scene["root"], "ecam_set"), e_thresh="None", event_type="none").outputs().e_thresh == 0.35
    os.rename(os.path.join(scene["root"], "ecam_set"), os.path.join(scene["root"], "decam_set"))
    assert EventSceneReader(os.path.join(scene["root"], "ecam_set"), event_type="decam_set").outputs().e_thresh == 1
