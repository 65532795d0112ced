"""nerfstudio checkpoint import / export for the hot-path modules (SURVEY.md section 8f-4, section 5 "checkpoint / resume").

The reference trainer writes ``<base>/nerfstudio_models/step-%09d.ckpt`` = ``torch.save({"step", "pipeline", "optimizers",
"scalers"})`` ([UP] ``Trainer.save_checkpoint``) and reads it back at R:lse_nerf/lse_trainer.py:85-122 (latest step chosen
by parsing the file names, ``:94``) through ``load_pipeline`` (R:lse_nerf/lse_pipeline.py:236-247: strips a DDP
``module.`` prefix, ``load_state_dict(strict=False)``).  The pipeline state dict names the model ``_model`` and keeps the
tcnn parameters under ``.tcnn_encoding.params``; this module maps those names onto ``lsenerf_amd.LSENeRFModel`` and back,
so a checkpoint trained by the reference loads here and a checkpoint written here loads in the reference.

Only loaders that execute nothing from the file are used (``torch.load(weights_only=True)``).
"""
from __future__ import annotations

import os
import re
from typing import Dict, Iterable, Optional, Tuple

import torch

CKPT_RE = re.compile(r"^step-(\d+)\.ckpt$")

# reference suffix -> ours (applied after the `_model.` / `module.` prefixes are stripped)
_TCNN_SUFFIX = ".tcnn_encoding.params"
_DEAD_KEYS = ("field.mlp_base_grid.hash_table",)       # torch-layout table the reference always allocates and, in tcnn
                                                       # mode, never reads (R:lse_nerf/lse_field.py:63-65)


def reference_to_local_key(key: str) -> Optional[str]:
    """Pipeline state-dict key of the reference -> state-dict key of ``lsenerf_amd.LSENeRFModel``.
    Returns None for entries that have no counterpart (dead table, data-manager state)."""
    if key.startswith("module."):
        key = key[len("module."):]
    if not key.startswith("_model."):
        return None                                        # datamanager.* (camera optimiser of the data manager, ...)
    key = key[len("_model."):]
    if key in _DEAD_KEYS:
        return None
    if key.endswith(_TCNN_SUFFIX):
        key = key[: -len(_TCNN_SUFFIX)] + ".params"
    return key


def local_to_reference_key(key: str) -> str:
    tcnn_modules = ("field.mlp_base_grid", "field.mlp_base_mlp", "field.mlp_head")
    for m in tcnn_modules:
        if key == m + ".params":
            return "_model." + m + _TCNN_SUFFIX
    return "_model." + key


def checkpoint_path(directory: str, step: int) -> str:
    return os.path.join(directory, f"step-{step:09d}.ckpt")


def latest_step(directory: str) -> int:
    """The reference's rule (R:lse_nerf/lse_trainer.py:94): the largest step number among the file names."""
    steps = [int(m.group(1)) for m in (CKPT_RE.match(f) for f in os.listdir(directory)) if m]
    if not steps:
        raise FileNotFoundError(f"no step-*.ckpt in {directory}")
    return max(steps)


def convert_pipeline_state(pipeline_state: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Iterable[str]]:
    """Reference ``loaded_state["pipeline"]`` -> (state dict for LSENeRFModel, keys that were dropped)."""
    out, dropped = {}, []
    for k, v in pipeline_state.items():
        lk = reference_to_local_key(k)
        if lk is None:
            dropped.append(k)
            continue
        if torch.is_tensor(v) and v.dtype in (torch.float16, torch.bfloat16):
            v = v.float()                                  # a half-precision tcnn build stores fp16 parameters
        out[lk] = v
    return out, dropped


def load_nerfstudio_checkpoint(path: str, model: torch.nn.Module, load_step: Optional[int] = None,
                               drop_camera_optimizer: bool = False,
                               optimizers: Optional[Dict[str, object]] = None) -> Dict[str, object]:
    """Load a reference checkpoint into ``model`` (and, when given, the optimizer states by param-group name into
    objects with ``load_state_dict`` -- e.g. ``{"fields": FlatAdam}`` -- so that training resumes with its Adam moments
    and learning-rate schedule position).  ``path`` is a ``.ckpt`` file or a ``nerfstudio_models`` directory
    (then ``load_step`` or the latest step is taken).  ``drop_camera_optimizer`` mirrors the reference's eval mode
    (R:lse_nerf/lse_trainer.py:68-82).  Returns {"step", "missing", "unexpected", "dropped"}; like the reference's
    ``strict=False`` load nothing is raised for missing / unexpected names, but shape mismatches are errors."""
    if os.path.isdir(path):
        step = latest_step(path) if load_step is None else int(load_step)
        path = checkpoint_path(path, step)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Checkpoint {path} does not exist")
    loaded = torch.load(path, map_location="cpu", weights_only=True)
    if "pipeline" not in loaded or "step" not in loaded:
        raise ValueError(f"{path} is not a nerfstudio checkpoint (keys: {sorted(loaded)})")
    state, dropped = convert_pipeline_state(loaded["pipeline"])
    if drop_camera_optimizer:
        for k in [k for k in state if "camera_optimizer" in k]:
            dropped.append(k)
            del state[k]
    own = model.state_dict()
    for k, v in state.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            raise ValueError(f"shape mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
    res = model.load_state_dict(state, strict=False)
    # the estimator caches a host copy of occs.mean(); a loaded grid invalidates it
    for mod in model.modules():
        if hasattr(mod, "_occ_mean_host"):
            mod._occ_mean_host = None
    if optimizers:   # resume: Adam moments + step count (the lr schedule continues from there)
        for name, o in optimizers.items():
            if name in loaded.get("optimizers", {}):
                o.load_state_dict(loaded["optimizers"][name])
    return {"step": int(loaded["step"]), "missing": list(res.missing_keys), "unexpected": list(res.unexpected_keys),
            "dropped": list(dropped)}


def save_nerfstudio_checkpoint(directory: str, model: torch.nn.Module, step: int,
                               optimizers: Optional[Dict[str, object]] = None, keep_only_latest: bool = False) -> str:
    """Write ``step-%09d.ckpt`` with the reference's layout and key names, so that R:lse_nerf/lse_trainer.py:85-122 can
    read it.  ``optimizers`` maps a param-group name ("fields", "camera_opt") to an object with ``state_dict()`` or to a
    plain dict of tensors."""
    os.makedirs(directory, exist_ok=True)
    pipeline = {local_to_reference_key(k): v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt_state = {}
    for name, o in (optimizers or {}).items():
        if hasattr(o, "state_dict"):
            opt_state[name] = o.state_dict()
        elif isinstance(o, dict):
            opt_state[name] = o
        else:   # a pickled object would make the file unreadable for the weights_only loader above
            raise TypeError(f"optimizers[{name!r}] must have state_dict() or be a dict of tensors, got {type(o).__name__}")
    path = checkpoint_path(directory, step)
    torch.save({"step": int(step), "pipeline": pipeline, "optimizers": opt_state, "scalers": {}}, path)
    if keep_only_latest:                                   # [UP] Trainer.save_checkpoint(save_only_latest_checkpoint=True)
        for f in os.listdir(directory):
            if CKPT_RE.match(f) and os.path.join(directory, f) != path:
                os.remove(os.path.join(directory, f))
    return path
