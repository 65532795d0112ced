"""Optional nerfstudio entry point (R:pyproject.toml:15-16, R:lse_nerf/lse_config.py:14-42).

nerfstudio discovers methods through the ``nerfstudio.method_configs`` entry-point group; the reference registers
``lsenerf = lse_nerf.lse_config:lsenerf_method``.  This module provides the same ``MethodSpecification`` with the model
swapped for the gfx950 hot path -- but only where nerfstudio itself is importable (it is not in the build container:
``import nerfstudio`` raises ModuleNotFoundError there, so the shim is exercised only as "fails with a clear message").
Data managers, trainer, viewer and writers stay the reference's (out of scope): the maintainer's
binding is the three-line import swap shown in INTEGRATION.md section A.

    [project.entry-points.'nerfstudio.method_configs']
    lsenerf-amd = 'lsenerf_amd.ns_plugin:lsenerf_method'
"""
from __future__ import annotations

OPTIMIZERS = {   # R:lse_nerf/lse_config.py:29-38 (consumed by lsenerf_amd.optim.FlatAdam outside nerfstudio)
    "fields": {"lr": 1e-2, "eps": 1e-15, "lr_final": 1e-4, "max_steps": 200000},
    "camera_opt": {"lr": 1e-3, "eps": 1e-15, "lr_final": 1e-4, "max_steps": 5000},
}
TRAIN_NUM_RAYS_PER_BATCH = 3512      # R:lse_nerf/lse_config.py:24
EVAL_NUM_RAYS_PER_CHUNK = 3512       # R:lse_nerf/lse_config.py:27


def convert_embed_config(ns_embed):
    """The reference's ``LSEEmbeddingConfig`` (R:lse_nerf/lse_embeddings.py:94-107) -> ``lsenerf_amd.LSEEmbeddingConfig``."""
    from dataclasses import fields
    from .field import LSEEmbeddingConfig
    if ns_embed is None or isinstance(ns_embed, LSEEmbeddingConfig):
        return ns_embed or LSEEmbeddingConfig()
    return LSEEmbeddingConfig(**{f.name: getattr(ns_embed, f.name) for f in fields(LSEEmbeddingConfig) if hasattr(ns_embed, f.name)})


def convert_model_config(ns_cfg):
    """nerfstudio's config object for this model (the reference's ``LSENeRFModelConfig(InstantNGPModelConfig)``,
    R:lse_nerf/lsenerf.py:47-99, as the pipeline hands it to ``Model.__init__``) -> ``lsenerf_amd.LSENeRFModelConfig``, field by
    field.  Fields this path does not consume (``_target``, collider / loss-coefficient tables, ``use_mapper_loss`` ...) are
    dropped; every field the hot path reads keeps the value the user configured, including the reference's post-init
    coercions (they are idempotent, so running them again on already-coerced values changes nothing)."""
    from dataclasses import fields
    from .model import LSENeRFModelConfig
    if isinstance(ns_cfg, LSENeRFModelConfig):
        return ns_cfg
    kw = {}
    for f in fields(LSENeRFModelConfig):
        if hasattr(ns_cfg, f.name):
            kw[f.name] = getattr(ns_cfg, f.name)
    if "embed_config" in kw:
        kw["embed_config"] = convert_embed_config(kw["embed_config"])
    if kw.get("ev_one_dim") is True:            # (never produced by the reference's post-init; tolerated)
        kw["ev_one_dim"] = "learned"
    return LSENeRFModelConfig(**kw)


def build_method_specification():
    """The reference's ``lsenerf_method`` with ``lsenerf_amd.LSENeRFModel`` behind the model config's ``_target``.
    Works on a deep copy: resolving this entry point must not rename or re-target the reference's own ``lsenerf`` method
    object, which lives in the same process."""
    try:
        from nerfstudio.plugins.types import MethodSpecification
    except ModuleNotFoundError as e:   # pragma: no cover - nerfstudio is absent from the build image
        raise ModuleNotFoundError(
            "lsenerf_amd.ns_plugin needs nerfstudio==0.3.2 (the reference's pin, R:pyproject.toml:6) to register the "
            "method; without it use lsenerf_amd.LSENeRFModel directly (INTEGRATION.md)") from e
    import copy
    # the reference's own config objects carry the data managers / trainer / optimisers; only the model target moves
    from lse_nerf.lse_config import lsenerf_method as ref   # noqa: E402  (the reference package must be installed too)
    from . import model as _model
    cfg = copy.deepcopy(ref.config)
    cfg.method_name = "lsenerf-amd"
    cfg.pipeline.model._target = _model.LSENeRFModel      # its __init__ converts the nerfstudio config (convert_model_config)
    return MethodSpecification(cfg, description="lsenerf on the MI355X hot path (lsenerf_amd)")


def __getattr__(name):   # lazy: importing this module never requires nerfstudio
    if name == "lsenerf_method":
        return build_method_specification()
    raise AttributeError(name)
