"""Reader for the LSENeRF-data-formatter scene layout: the data that feeds the hot path's callers.

Layout (R:lse_nerf/lse_parser.py, R:lse_nerf/lse_dataset.py)::

    <scene>/rel_cam.json                     {"R": 3x3, "T": 3}  colour -> event camera, OpenCV frame, colmap units
    <scene>/colcam_set/                      (optionally <quality>_<image_type>_colcam_set/)
        camera/*.json                        orientation 3x3 (world->cam, OpenCV), position 3, focal_length, principal_point,
                                             image_size, radial_distortion, tangential_distortion, optional t
        rgb/1x/*.png|jpg
        dataset.json                         {"train_ids": [...], "val_ids": [...], optional "half_train_ids"}
        metadata.json                        {"<img id>": {"appearance_id": k, ...}, optional "colmap_scale": s}
        optional: msk.npy, camera_transform.json {"translation"}, full_camera/*.json, scene.json
    <scene>/ecam_set/                        same camera files, or prev_camera/ + next_camera/ for event frames
        eimgs/eimgs_1x.npy                   [n_frames, H, W] accumulated events per frame
        scene.json                           optional {"e_thresh": ...}

Pure numpy / json host code; nothing here touches the GPU.  ``numpy.load`` is used with its default
``allow_pickle=False`` (and memory-mapped for the event stack), images are opened with PIL on demand.
"""
from __future__ import annotations

import glob
import json
import os.path as osp
import warnings
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .cameras import EdCameras, HardCamType


# ---------------------------------------------------------------------------------------------------- helpers
def load_json(path: str):
    """R:lse_nerf/lse_parser.py:32-45: a missing file is not an error (returns None); a non-.json name is."""
    if not osp.exists(path):
        warnings.warn(f"{path} does not exist")
        return None
    assert osp.splitext(path)[-1] == ".json", f"{osp.basename(path)} is not json"
    with open(path, encoding="UTF-8") as f:
        return json.load(f)


def cv_w2c_to_gl_c2w(w2c: np.ndarray) -> np.ndarray:
    """OpenCV world->camera [4,4] -> OpenGL camera->world [4,4] (R:lse_nerf/lse_parser.py:48-63): invert the rigid
    transform and flip the camera's y and z axes."""
    src = np.asarray(w2c, dtype=np.float64)
    rot, t = src[:3, :3], src[:3, 3]
    right, up, fwd = rot                               # rows of the world->camera rotation = camera axes in world space
    out = src.copy()
    out[:3, :3] = np.stack([right, -up, -fwd]).T
    out[:3, 3] = -rot.T @ t
    return out


def _image_files(img_dir: str) -> List[str]:
    return sorted(glob.glob(osp.join(img_dir, "rgb", "1x", "*.[pj][np]g")))


# ---------------------------------------------------------------------------------------------------- outputs
@dataclass
class SceneOutputs:
    """What the reference's ``CameraDataparserOutputs`` / ``EventsparserOutputs`` carry (R:lse_nerf/lse_parser.py:66-71,
    248-253)."""
    cameras: EdCameras
    scene_aabb: torch.Tensor                          # [2,3]
    dataparser_scale: float
    appearance_ids: List[int]
    image_filenames: Optional[List[str]] = None
    msk: Optional[np.ndarray] = None
    dM: Optional[torch.Tensor] = None                 # colour -> event relative pose in the OpenGL frame [4,4]
    distortion_params: Optional[torch.Tensor] = None  # (k1,k2,k3,0,p1,p2), None when all zero
    events: Optional[np.ndarray] = None               # [n, H, W, 1]
    e_thresh: Optional[float] = None
    prev_cameras: Optional[EdCameras] = None
    next_cameras: Optional[EdCameras] = None
    data_idxs: List[int] = field(default_factory=list)


# ---------------------------------------------------------------------------------------------------- base reader
class CameraSetReader:
    """R:lse_nerf/lse_parser.py:79-241 (``CameraParser``)."""

    def __init__(self, data_dir: str, scale_factor: float = 1.0, scene_scale: float = 1.0):
        self.data = str(data_dir)
        self.scale_factor = float(scale_factor)
        self.scene_scale = float(scene_scale)
        tf = osp.join(self.data, "camera_transform.json")
        self.cam_translation = None
        if osp.exists(tf):
            with open(tf) as f:
                self.cam_translation = np.array(json.load(f)["translation"])
        self.cam_data_json = self.load_camera_jsons()
        self.metadata = self._load_metadata()
        keys = sorted(self.metadata.keys())
        self.appearance_ids = [self.metadata[k]["appearance_id"] for k in keys]

    # -- files
    def load_camera_jsons(self, cam_dir: Optional[str] = None, idxs: Optional[Sequence[int]] = None):
        cam_dir = cam_dir or osp.join(self.data, "camera")
        if not osp.exists(cam_dir):
            return None
        fs = sorted(glob.glob(osp.join(cam_dir, "*.json")))
        if idxs is not None:
            kept = []
            for i in idxs:
                if i < len(fs):
                    kept.append(fs[i])
                else:                                      # the reference only warns (:121-127)
                    warnings.warn(f"camera index {i} out of range of {len(fs)} files")
            fs = kept
        return [load_json(f) for f in fs]

    def _load_metadata(self) -> Dict[int, dict]:
        meta = load_json(osp.join(self.data, "metadata.json")) or {}
        out = {}
        for k, v in meta.items():
            try:
                img_id = int(k)
            except (TypeError, ValueError):               # e.g. the "colmap_scale" entry
                continue
            v = dict(v)
            v["img_id"] = img_id
            out[img_id] = v
        return out

    def load_msk(self, data_idxs: Optional[Sequence[int]] = None) -> Optional[np.ndarray]:
        f = osp.join(self.data, "msk.npy")
        if not osp.exists(f):
            return None
        msk = np.load(f)
        if data_idxs is not None:
            msk = np.stack([msk[i] for i in data_idxs])
        return msk

    def scene_aabb(self) -> torch.Tensor:
        s = self.scene_scale
        return torch.tensor([[-s, -s, -s], [s, s, s]], dtype=torch.float32)

    def max_appearance_id(self) -> int:
        return max(v["appearance_id"] for v in self.metadata.values()) + 1

    # -- cameras
    def format_cameras(self, data: List[dict], cam_type: int, calc_dm: bool = False):
        """R:lse_nerf/lse_parser.py:148-199.  Returns EdCameras (+ dM, distortion)."""
        n = len(data)
        c2ws = np.tile(np.eye(4, dtype=np.float32)[None], (n, 1, 1))
        w2cs = np.zeros((n, 4, 4), dtype=np.float32)
        times = None
        for i, d in enumerate(data):
            rot = np.array(d["orientation"], dtype=np.float64)
            pos = np.array(d["position"], dtype=np.float64).reshape(3, 1)
            if self.cam_translation is not None:
                pos = pos + self.cam_translation
            w2c = np.concatenate([np.concatenate([rot, -rot @ pos], axis=1), np.array([[0.0, 0, 0, 1]])], 0)
            w2cs[i] = w2c
            c2ws[i, :3, :4] = cv_w2c_to_gl_c2w(w2c)[:3, :4]
            if d.get("t") is not None:
                times = [float(d["t"])] if times is None else times + [float(d["t"])]
        dM = None
        meta_raw = load_json(osp.join(self.data, "metadata.json")) or {}
        if meta_raw.get("colmap_scale") is not None:
            dM = self.relative_event_camera(w2cs, meta_raw["colmap_scale"])
        c2ws[:, :3, 3] *= self.scale_factor
        d0 = data[0]
        cx, cy = d0["principal_point"]
        w, h = d0["image_size"]
        k1, k2, k3 = d0["radial_distortion"]
        p1, p2 = d0["tangential_distortion"]
        dist = torch.tensor((k1, k2, k3, 0, p1, p2), dtype=torch.float32)
        cams = EdCameras(camera_to_worlds=torch.from_numpy(c2ws)[:, :3, :4], fx=d0["focal_length"], fy=d0["focal_length"],
                         cx=cx, cy=cy, width=w, height=h,
                         times=torch.tensor(times, dtype=torch.float32) if times is not None else None)
        cams.set_hard_cam_type(cam_type)
        cams.distortion_params = None if float(dist.sum()) == 0 else dist     # applied by EdCameras.generate_rays
        return (cams, dM) if calc_dm else cams

    def relative_event_camera(self, w2cs: np.ndarray, colmap_scale: float) -> torch.Tensor:
        """R:lse_nerf/lse_parser.py:201-232: dM with  c2w_evs(gl) = c2w_rgb(gl) @ dM, from rel_cam.json next to the set."""
        with open(osp.join(osp.dirname(self.data), "rel_cam.json")) as f:
            rel = json.load(f)
        R, T = np.array(rel["R"], dtype=np.float64), np.array(rel["T"], dtype=np.float64) * colmap_scale
        d_cv = np.concatenate([np.concatenate([R, T.reshape(3, 1)], axis=1), np.array([[0.0, 0, 0, 1]])], 0)
        rgb_gl = np.stack([cv_w2c_to_gl_c2w(m) for m in w2cs])
        evs_gl = np.stack([cv_w2c_to_gl_c2w(d_cv @ m) for m in w2cs])
        rgb_gl[:, :3, 3] *= self.scale_factor
        evs_gl[:, :3, 3] *= self.scale_factor
        d0 = np.linalg.inv(rgb_gl[0]) @ evs_gl[0]
        if len(rgb_gl) > 5:                               # the reference cross-checks against camera 5 (:227-229)
            d5 = np.linalg.inv(rgb_gl[5]) @ evs_gl[5]
            assert (np.abs(d0 - d5) < 1e-6).all(), "gl relative extrinsics calculated wrong!"
        return torch.tensor(d0).float()


# ---------------------------------------------------------------------------------------------------- colour set
class ColorSceneReader(CameraSetReader):
    """R:lse_nerf/lse_parser.py:362-484 (``Color``)."""
    SPLIT_KEYS = {"train": "train_ids", "test": "val_ids", "val": "val_ids"}

    def __init__(self, data_dir: str, scale_factor: float = 1.0, scene_scale: float = 1.0, quality: str = "clear",
                 image_type: str = "gamma", is_eval: bool = False, do_pretrain: bool = False):
        super().__init__(data_dir, scale_factor, scene_scale)
        self.quality, self.image_type = quality, image_type
        self.is_eval, self.do_pretrain = is_eval, do_pretrain
        self.dataset_meta = None

    def image_dir(self, *prefixes) -> str:
        """<quality>_<image_type>_colcam_set if it exists, else colcam_set (:395-409)."""
        prefix = "".join(f"{e}_" for e in prefixes if e not in (None, ""))
        base = osp.dirname(self.data)
        cand = osp.join(base, prefix + "colcam_set")
        return cand if osp.exists(cand) else osp.join(base, "colcam_set")

    def outputs(self, split: str = "train", spec_data_idxs: Optional[Sequence[int]] = None) -> SceneOutputs:
        quality = self.quality if split == "train" else "clear"          # evaluation always on the clear images
        img_dir = self.image_dir(quality, self.image_type)
        if osp.abspath(img_dir) != osp.abspath(self.data):
            CameraSetReader.__init__(self, img_dir, self.scale_factor, self.scene_scale)
        self.dataset_meta = load_json(osp.join(self.data, "dataset.json"))
        if split == "train" and self.is_eval and self.dataset_meta.get("half_train_ids") is not None:
            id_key = "half_train_ids"
        else:
            if self.is_eval and not self.do_pretrain:
                split = "val"
            id_key = self.SPLIT_KEYS[split]
        img_fs = _image_files(img_dir)
        idxs = sorted(int(e) for e in self.dataset_meta[id_key]) if spec_data_idxs is None else list(spec_data_idxs)
        idxs = [i for i in idxs if i < len(img_fs) - 1]                  # the last frame has no successor (:425)
        cams, dM = self.format_cameras([self.cam_data_json[i] for i in idxs], HardCamType.RGB, calc_dm=True)
        return SceneOutputs(cameras=cams, scene_aabb=self.scene_aabb(), dataparser_scale=self.scale_factor,
                            appearance_ids=[self.appearance_ids[i] for i in idxs],
                            image_filenames=[img_fs[i] for i in idxs], msk=self.load_msk(idxs), dM=dM,
                            distortion_params=cams.distortion_params, data_idxs=idxs)

    def train_ids(self) -> List[int]:
        n = len(_image_files(self.data))
        meta = self.dataset_meta or load_json(osp.join(self.data, "dataset.json"))
        return sorted(int(e) for e in meta["train_ids"] if int(e) < n - 1)

    def all_cameras(self) -> EdCameras:
        full = osp.join(self.data, "full_camera")
        data = self.load_camera_jsons(full) if osp.exists(full) else self.cam_data_json[:-1]
        return self.format_cameras(data, HardCamType.RGB)

    def train_times(self) -> Optional[torch.Tensor]:
        data = [self.cam_data_json[i] for i in self.train_ids()]
        if not data or data[0].get("t") is None:
            return None
        return torch.tensor([d["t"] for d in data], dtype=torch.float32)


# ---------------------------------------------------------------------------------------------------- event set
class EventSceneReader(CameraSetReader):
    """R:lse_nerf/lse_parser.py:255-360 (``Events``)."""

    def __init__(self, data_dir: str, scale_factor: float = 1.0, scene_scale: float = 1.0,
                 e_thresh: Optional[float] = None, event_type: Optional[str] = None):
        if isinstance(e_thresh, str):
            e_thresh = None if e_thresh.lower() == "none" else float(e_thresh)
        if isinstance(event_type, str) and event_type.lower() == "none":
            event_type = None
        if event_type is not None:                        # sibling directory named after the event type (:268-270)
            data_dir = osp.join(osp.dirname(str(data_dir)), event_type)
        super().__init__(data_dir, scale_factor, scene_scale)
        self.e_thresh_override, self.event_type = e_thresh, event_type

    def load_events(self, idxs: Sequence[int]) -> np.ndarray:
        src = np.load(osp.join(self.data, "eimgs", "eimgs_1x.npy"), mmap_mode="r")
        ev = np.zeros((len(idxs), *src.shape[1:]), dtype=src.dtype)
        for i, idx in enumerate(idxs):
            ev[i] = src[idx]
        return ev[..., None]

    def outputs(self, split: str = "train") -> SceneOutputs:
        if split != "train":
            warnings.warn(f"event camera data supports the train split only, got {split}")
        meta = load_json(osp.join(self.data, "dataset.json"))
        idxs = sorted(int(e) for e in meta["train_ids"])
        prev_dir, next_dir = osp.join(self.data, "prev_camera"), osp.join(self.data, "next_camera")
        prev_c = next_c = None
        if osp.exists(prev_dir):
            pj, nj = self.load_camera_jsons(prev_dir, idxs), self.load_camera_jsons(next_dir, idxs)
            cams = self.format_cameras(pj, HardCamType.EVS)
            prev_c, next_c = self.format_cameras(pj, HardCamType.EVS), self.format_cameras(nj, HardCamType.EVS)
        else:
            cams = self.format_cameras(self.cam_data_json, HardCamType.EVS)
        scene = load_json(osp.join(self.data, "scene.json")) if osp.exists(osp.join(self.data, "scene.json")) else None
        e_thresh = 0.2
        if scene is not None and scene.get("e_thresh") is not None:
            e_thresh = scene["e_thresh"]
        if self.e_thresh_override is not None:
            e_thresh = self.e_thresh_override
        if self.event_type == "decam_set":
            e_thresh = 1
        return SceneOutputs(cameras=cams, scene_aabb=self.scene_aabb(), dataparser_scale=self.scale_factor,
                            appearance_ids=[self.appearance_ids[i] for i in idxs], msk=self.load_msk(),
                            events=self.load_events(idxs), e_thresh=e_thresh, prev_cameras=prev_c, next_cameras=next_c,
                            distortion_params=cams.distortion_params, data_idxs=idxs)


# ---------------------------------------------------------------------------------------------------- datasets
class ColorDataset:
    """R:lse_nerf/lse_dataset.py:18-58: uint8 images -> float [H,W,3] in [0,1], per-image appearance id and mask."""

    def __init__(self, outputs: SceneOutputs, scale_factor: float = 1.0, use_gray: bool = False):
        self.out, self.scale_factor, self.use_gray = outputs, scale_factor, use_gray
        self.appearance_ids = outputs.appearance_ids
        self.msk = torch.from_numpy(outputs.msk) if outputs.msk is not None else None
        self.cameras = outputs.cameras

    def __len__(self):
        return len(self.out.image_filenames)

    def get_numpy_image(self, image_idx: int) -> np.ndarray:
        from PIL import Image
        img = Image.open(self.out.image_filenames[image_idx])
        if self.scale_factor != 1.0:
            w, h = img.size
            img = img.resize((int(w * self.scale_factor), int(h * self.scale_factor)), resample=Image.BILINEAR)
        if self.use_gray:
            img = img.convert("L")
        a = np.array(img, dtype="uint8")
        if a.ndim == 2:
            a = a[:, :, None].repeat(3, axis=2)
        assert a.ndim == 3 and a.dtype == np.uint8 and a.shape[2] in (3, 4), f"Image shape of {a.shape} is in correct."
        return a

    def get_image(self, image_idx: int) -> torch.Tensor:
        a = torch.from_numpy(self.get_numpy_image(image_idx).astype("float32") / 255.0)
        return a[:, :, :3]

    def get_data(self, image_idx: int) -> Dict:
        d = {"image_idx": image_idx, "image": self.get_image(image_idx), "appearance_id": self.appearance_ids[image_idx]}
        if self.msk is not None:
            d["msk"] = self.msk[image_idx]
        return d


class EventFrameDataset(ColorDataset):
    """R:lse_nerf/lse_dataset.py:60-90: event frames scaled by the contrast threshold."""

    def __init__(self, outputs: SceneOutputs, scale_factor: float = 1.0):
        super().__init__(outputs, scale_factor)
        self.e_thresh = torch.tensor([outputs.e_thresh], dtype=torch.float32)
        evs = outputs.events
        if len(evs) > 1000:                               # the reference drops the last 8 frames of long sequences (:66-67)
            evs = evs[:-8]
        if outputs.e_thresh == 1 and self.msk is not None:
            evs = np.clip(evs / 255, 0, 1)
        self.evs = torch.from_numpy(np.ascontiguousarray(evs))

    def __len__(self):
        return len(self.evs)

    def get_numpy_image(self, image_idx: int):
        raise AssertionError("no images in event frames dataset")

    def get_image(self, image_idx: int) -> torch.Tensor:
        return (self.evs[image_idx] * self.e_thresh).float()

    def get_data(self, image_idx: int) -> Dict:
        d = {"image_idx": image_idx, "image": self.get_image(image_idx), "appearance_id": self.appearance_ids[image_idx],
             "e_thresh": self.e_thresh}
        if self.msk is not None:
            d["msk"] = self.msk[image_idx]
        return d
