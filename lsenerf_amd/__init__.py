"""lsenerf_amd -- MI355X (gfx950) native hot path of LSENeRF: hash-grid field, occupancy-grid ray marching and
alpha compositing as hand-written HIP kernels behind a C-ABI (include/lse_hip.h), with the reference's
Field / OccGridEstimator / Renderer / Model interface on top.  See DESIGN.md."""
from . import _lib  # noqa: F401
from .field import Ed_HashEncoding, FieldHeadNames, LSEEmbeddingConfig, LSEField, MLP  # noqa: F401
from .grid_estimator import LSEOccGridEstimator  # noqa: F401
from .model import LSENeRFModel, LSENeRFModelConfig, VolumetricSampler  # noqa: F401
from .rays import Frustums, RayBundle, RaySamples, SceneBox, SceneContraction  # noqa: F401
from .renderer import AccumulationRenderer, DepthRenderer, LinearRenderer, RGBRenderer  # noqa: F401

__all__ = ["Ed_HashEncoding", "LSEField", "LSEOccGridEstimator", "LinearRenderer", "LSENeRFModel",
           "LSENeRFModelConfig", "RayBundle", "RaySamples", "Frustums", "SceneBox", "SceneContraction"]
