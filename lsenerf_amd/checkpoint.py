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


# buffers of the reference's model (they sit between the parameters in the pipeline state dict but are no optimizer entries)
_REF_BUFFER_LEAVES = ("aabb", "max_res", "num_levels", "log2_hashmap_size", "c2g_vec")
# get_param_groups order of the reference (R:lse_nerf/lsenerf.py:231-249): field parameters, then the mapper modules
_REF_GROUP_PREFIXES = ("field.", "rgb_mapper.", "rgb_to_one.", "evs_mapper.")
# nn.Sequential alias of (mlp_base_grid, mlp_base_mlp) (R:lse_nerf/lse_field.py:208): same tensors again, no new parameters
_REF_ALIAS_PREFIXES = ("field.mlp_base.",)


def reference_param_order(pipeline_state: Dict[str, torch.Tensor]) -> list:
    """The reference's ``get_param_groups()["fields"]`` as a list of pipeline keys (``_model.`` / ``module.`` prefixes
    stripped), reconstructed from the checkpoint itself: ``list(self.field.parameters())`` follows the module registration
    order, which is the key order of the saved state dict, followed by ``rgb_mapper`` / ``rgb_to_one`` / ``evs_mapper``
    parameters.  Buffers and the ``mlp_base`` Sequential aliases are skipped; the dead hash table and zero-sized tcnn
    parameters (tcnn's SH encoding) stay in the list -- they consume an index in torch's optimizer state."""
    keys = []
    for k in pipeline_state:
        if k.startswith("module."):
            k = k[len("module."):]
        if k.startswith("_model."):
            keys.append(k[len("_model."):])
    order = []
    for prefix in _REF_GROUP_PREFIXES:
        for k in keys:
            if not k.startswith(prefix) or k.startswith(_REF_ALIAS_PREFIXES) or k.rsplit(".", 1)[-1] in _REF_BUFFER_LEAVES:
                continue
            order.append(k)
    return order


def reference_optimizer_index_map(pipeline_state: Dict[str, torch.Tensor], model: torch.nn.Module, params: list):
    """({reference optimizer index: position in ``params`` or None}, {reference index: key}) for the "fields" group.
    ``params`` is the local optimizer's parameter list (``FlatParams.params``); a reference parameter maps to the local
    parameter of the same (translated) name, to None when this model has no such tensor (dead table, zero-sized tcnn
    parameters).  Raises when a non-empty, non-dead reference parameter has no local counterpart."""
    local_by_name = dict(model.named_parameters())
    pos = {id(p): i for i, p in enumerate(params)}
    index_map, names = {}, {}
    stripped = {}
    for k, v in pipeline_state.items():
        kk = k[len("module."):] if k.startswith("module.") else k
        if kk.startswith("_model."):
            stripped[kk[len("_model."):]] = v
    for i, k in enumerate(reference_param_order(pipeline_state)):
        names[i] = k
        lk = reference_to_local_key("_model." + k)
        p = local_by_name.get(lk) if lk is not None else None
        if p is not None and id(p) in pos:
            index_map[i] = pos[id(p)]
        elif lk is None or stripped[k].numel() == 0:
            index_map[i] = None
        else:
            raise ValueError(f"reference parameter {k!r} (optimizer index {i}) has no counterpart in the local parameter list")
    return index_map, names


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
                sd = loaded["optimizers"][name]
                flat = getattr(o, "flat", None)
                if name == "fields" and flat is not None and not sd.get("lsenerf_amd_layout", False):
                    # reference-written state: indices follow the reference's parameter list, translate them by name
                    imap, names = reference_optimizer_index_map(loaded["pipeline"], model, flat.params)
                    o.load_state_dict(sd, index_map=imap, names=names)
                else:
                    o.load_state_dict(sd)
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
