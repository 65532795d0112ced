"""Host-side mirror of the reference's field interface, running on the gfx950 kernels.

Mirrors (same names, argument meaning, shapes, error behaviour):
  * ``Ed_HashEncoding``                      R:lse_nerf/lse_field.py:43-91
  * ``LSEField`` (get_density/get_outputs/forward/density_fn)   R:lse_nerf/lse_field.py:94-360
  * ``EvsFrameEmbedding`` / ``GlobalEmbedding`` / ``LSEEmbeddingConfig``   R:lse_nerf/lse_embeddings.py:19-107
nerfstudio itself is not importable here, so ``Field``/``MLP``/``SHEncoding`` are duck-typed.  Parameters use
the tcnn layouts the reference trains with (``implementation="tcnn"``, R:lse_nerf/lsenerf.py:175); unlike the
reference, the dead torch ``hash_table`` (R:lse_nerf/lse_field.py:63-65, 67 MB never used in tcnn mode) is not
allocated.
"""
from __future__ import annotations

import math
import weakref
from dataclasses import dataclass
from enum import Enum
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _lib, ops
from .rays import RaySamples


class FieldHeadNames(Enum):
    """nerfstudio ``FieldHeadNames`` (the members a nerfacto-style field can return)."""
    RGB = "rgb"
    DENSITY = "density"
    PRED_NORMALS = "pred_normals"
    UNCERTAINTY = "uncertainty"
    TRANSIENT_RGB = "transient_rgb"
    TRANSIENT_DENSITY = "transient_density"
    SEMANTICS = "semantics"


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


# ----------------------------------------------------------------------------------------------------
# encodings / MLP modules (parameter containers + single-op forwards)
# ----------------------------------------------------------------------------------------------------
class Ed_HashEncoding(nn.Module):
    """tcnn HashGrid with the constructor signature of R:lse_nerf/lse_field.py:44-51.

    ``forward(x[N,3] in [0,1]) -> [N, L*F]`` like tcnn's binding; ``forward_levelmajor`` returns the [L,N,F]
    layout the fused MLP consumes without a transpose."""

    def __init__(self, num_levels: int = 16, min_res: int = 16, max_res: int = 1024, log2_hashmap_size: int = 19,
                 features_per_level: int = 2, hash_init_scale: float = 0.001, implementation: str = "hip",
                 interpolation: Optional[str] = None) -> None:
        super().__init__()
        assert interpolation is None or interpolation == "Linear", \
            f"interpolation '{interpolation}' is not supported for the hip encoding backend"
        self.num_levels = num_levels
        self.features_per_level = features_per_level
        self.log2_hashmap_size = log2_hashmap_size
        self.hash_table_size = 2 ** log2_hashmap_size
        self._meta = ops.make_grid_meta(num_levels, features_per_level, log2_hashmap_size, min_res, max_res=max_res)
        # sample regime of the rays this grid is trained on, set by the model from its sampler configuration (constant step vs
        # steps that grow with the distance): selects the hash backward's path thresholds (ops.HASH_BWD_DENSE_STEPS)
        self.dense_steps: bool = False
        # tcnn grid init U(-1e-4, 1e-4) (hash_init_scale only applies to the torch table the reference leaves dead)
        self.params = nn.Parameter((torch.rand(self.meta.n_params) * 2 - 1) * 1e-4)

    @property
    def meta(self) -> "ops.GridMeta":
        import dataclasses
        tuning = ops.HASH_BWD_DENSE_STEPS if self.dense_steps else ops.HASH_BWD_DEFAULT
        if self._meta.bwd_tuning != tuning:
            self._meta = dataclasses.replace(self._meta, bwd_tuning=tuning)
        return self._meta

    def get_out_dim(self) -> int:
        return self.num_levels * self.features_per_level

    def forward_levelmajor(self, in_tensor: Tensor, n_dev: Optional[Tensor] = None) -> Tensor:
        return ops.hash_encode(in_tensor, self.params, self.meta, n_dev=n_dev)

    def forward(self, in_tensor: Tensor) -> Tensor:
        y = self.forward_levelmajor(in_tensor.reshape(-1, 3).contiguous())
        n = y.shape[1]
        return y.permute(1, 0, 2).reshape(n, -1).view(*in_tensor.shape[:-1], -1)


class MLP(nn.Module):
    """nerfstudio ``MLP(implementation="tcnn")``: bias-free, ReLU, tcnn flat ``params``."""

    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: int, activation=None,
                 out_activation=None, implementation: str = "hip", in_layout: int = _lib.LSE_IN_ROWMAJOR) -> None:
        super().__init__()
        self.in_dim, self.out_dim, self.layer_width, self.num_layers = in_dim, out_dim, layer_width, num_layers
        self.in_pad = _pad16(in_dim)     # tcnn pads network inputs to a multiple of 16 (with ones)
        self.n_hidden_layers = num_layers - 1
        assert out_dim <= 16, "the fused kernel pads outputs to 16"
        self.out_act = _lib.LSE_ACT_SIGMOID if isinstance(out_activation, nn.Sigmoid) or out_activation == "Sigmoid" \
            else _lib.LSE_ACT_NONE
        self.in_layout = in_layout
        shapes = [(layer_width, self.in_pad)] + [(layer_width, layer_width)] * (self.n_hidden_layers - 1) + [(16, layer_width)]
        self.shapes = shapes
        chunks = []
        for (o, i) in shapes:   # tcnn xavier_uniform
            s = math.sqrt(6.0 / (i + o))
            chunks.append((torch.rand(o * i) * 2 - 1) * s)
        self.params = nn.Parameter(torch.cat(chunks))

    def get_out_dim(self) -> int:
        return self.out_dim

    def meta(self, n_in: Optional[int] = None) -> ops.MlpMeta:
        return ops.MlpMeta(self.in_pad if n_in is None else n_in, self.layer_width, self.n_hidden_layers, self.out_act,
                           self.in_layout)

    def split_padding(self):
        """For inputs narrower than the tcnn-padded width (the 8 features of an L=4 grid): kernel parameters over the
        real columns, plus the ones-padding columns folded into one bias row [1, width]."""
        w0 = self.first_layer()
        kparams = torch.cat([w0[:, : self.in_dim].reshape(-1), self.rest()])
        bias = w0[:, self.in_dim:].sum(dim=1)[None, :].contiguous()
        return kparams, bias

    def first_layer(self) -> Tensor:
        return self.params[: self.layer_width * self.in_pad].view(self.layer_width, self.in_pad)

    def rest(self) -> Tensor:
        return self.params[self.layer_width * self.in_pad:]

    def forward(self, in_tensor: Tensor) -> Tensor:
        """Generic path (any caller): ``in_tensor[..., in_dim]`` row-major like nerfstudio's MLP; narrower inputs are
        padded with ones like tcnn's Identity encoding.  (The field's fast path feeds level-major features directly.)"""
        x = in_tensor.reshape(-1, in_tensor.shape[-1])
        n = x.shape[0]
        if self.in_layout == _lib.LSE_IN_LEVELMAJOR:
            xl = x.reshape(n, self.in_dim // 2, 2).permute(1, 0, 2).contiguous()
            if self.in_pad == self.in_dim:
                out = ops.fused_mlp(self.params, xl, self.meta(), n)
            else:
                kparams, bias = self.split_padding()
                idx = torch.zeros(n, dtype=torch.int32, device=x.device)
                seg = torch.tensor([[0, n]], dtype=torch.int64, device=x.device)
                out = ops.fused_mlp(kparams, xl, self.meta(self.in_dim), n, bias, idx, seg)
        else:
            if x.shape[-1] < self.in_pad:
                x = torch.cat([x, torch.ones(n, self.in_pad - x.shape[-1], device=x.device, dtype=x.dtype)], dim=-1)
            out = ops.fused_mlp(self.params, x.contiguous(), self.meta(), n)
        return out[:, : self.out_dim].view(*in_tensor.shape[:-1], self.out_dim)


class DenseMLP(nn.Module):
    """nerfstudio ``MLP(implementation="tcnn")`` for the shapes the fused kernels do not cover (outputs wider than 16): the side
    heads the reference's field can switch on (R:lse_nerf/lse_field.py:210-252).  Same flat ``params`` as tcnn (bias-free,
    row-major ``[out, in]`` per layer, input padded to a multiple of 16 with ones, output to a multiple of 16), evaluated with
    the device's library GEMMs (rocBLAS behind ``torch.mm``) in fp32 -- plain GEMMs off the hot path, differentiated by autograd."""

    def __init__(self, in_dim: int, num_layers: int, layer_width: int, out_dim: int, activation=None, out_activation=None,
                 implementation: str = "hip") -> None:
        super().__init__()
        assert out_activation is None, "the side heads end in a linear layer"
        self.in_dim, self.out_dim, self.layer_width, self.num_layers = in_dim, out_dim, layer_width, num_layers
        self.in_pad, self.out_pad = _pad16(in_dim), _pad16(out_dim)
        self.shapes = [(layer_width, self.in_pad)] + [(layer_width, layer_width)] * (num_layers - 2) + [(self.out_pad, layer_width)]
        chunks = []
        for (o, i) in self.shapes:   # tcnn xavier_uniform
            b = math.sqrt(6.0 / (i + o))
            chunks.append((torch.rand(o * i) * 2 - 1) * b)
        self.params = nn.Parameter(torch.cat(chunks))

    def get_out_dim(self) -> int:
        return self.out_dim

    def forward(self, in_tensor: Tensor) -> Tensor:
        x = in_tensor.reshape(-1, in_tensor.shape[-1])
        if x.shape[-1] < self.in_pad:
            x = torch.cat([x, torch.ones(x.shape[0], self.in_pad - x.shape[-1], device=x.device, dtype=x.dtype)], dim=-1)
        off = 0
        for k, (o, i) in enumerate(self.shapes):
            w = self.params[off:off + o * i].view(o, i)
            off += o * i
            x = x @ w.t()
            if k < len(self.shapes) - 1:
                x = torch.relu(x)
        return x[:, : self.out_dim].reshape(*in_tensor.shape[:-1], self.out_dim)


class FieldHead(nn.Module):
    """nerfstudio ``FieldHead``: ``net = nn.Linear(in_dim, out_dim)`` + activation (state-dict key ``<name>.net.*``)."""

    def __init__(self, out_dim: int, field_head_name: FieldHeadNames, in_dim: int, activation: Optional[nn.Module] = None) -> None:
        super().__init__()
        self.out_dim, self.field_head_name, self.in_dim, self.activation = out_dim, field_head_name, in_dim, activation
        self.net = nn.Linear(in_dim, out_dim)

    def forward(self, in_tensor: Tensor) -> Tensor:
        out = self.net(in_tensor)
        return out if self.activation is None else self.activation(out)


class PredNormalsFieldHead(FieldHead):
    """nerfstudio ``PredNormalsFieldHead``: tanh, then unit length."""

    def __init__(self, in_dim: int) -> None:
        super().__init__(3, FieldHeadNames.PRED_NORMALS, in_dim, nn.Tanh())

    def forward(self, in_tensor: Tensor) -> Tensor:
        return torch.nn.functional.normalize(super().forward(in_tensor), dim=-1)


class FrequencyEncoding(nn.Module):
    """nerfstudio ``NeRFEncoding(in_dim, num_frequencies, min_freq_exp=0, max_freq_exp=num_frequencies-1, implementation="tcnn")``
    = tcnn's ``Frequency`` encoding (R:lse_nerf/lse_field.py:190-192): output ``j`` of input ``i`` is
    ``sin(x_i * 2^((j / 2) % F) * pi + (j % 2) * pi / 2)`` -- per input dimension (sin, cos) of frequency 0, (sin, cos) of
    frequency 1, ...; parameter-free, element-wise torch ops on the device."""

    def __init__(self, in_dim: int = 3, num_frequencies: int = 2) -> None:
        super().__init__()
        self.in_dim, self.num_frequencies = in_dim, num_frequencies

    def get_out_dim(self) -> int:
        return self.in_dim * self.num_frequencies * 2

    def forward(self, in_tensor: Tensor) -> Tensor:
        f = torch.exp2(torch.arange(self.num_frequencies, device=in_tensor.device, dtype=in_tensor.dtype)) * math.pi
        arg = in_tensor[..., :, None, None] * f[:, None]                                             # [.., in, F, 1]
        arg = arg + torch.tensor([0.0, math.pi / 2], device=in_tensor.device, dtype=in_tensor.dtype)  # [.., in, F, 2]
        return torch.sin(arg).reshape(*in_tensor.shape[:-1], self.get_out_dim())


# ----------------------------------------------------------------------------------------------------
# embeddings (R:lse_nerf/lse_embeddings.py)
# ----------------------------------------------------------------------------------------------------
class Embedding(nn.Module):
    def __init__(self, in_dim: int, out_dim: int) -> None:
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.embedding = nn.Embedding(in_dim, out_dim)

    def mean(self, dim=0):
        return self.embedding.weight.mean(dim)

    def forward(self, in_tensor: Tensor) -> Tensor:
        return self.embedding(in_tensor)


class EvsFrameEmbedding(Embedding):
    """R:lse_nerf/lse_embeddings.py:19-70 (per-frame appearance embedding indexed by metadata["appearance_id"])."""

    def __init__(self, config, num_imgs, num_dims) -> None:
        self.config = config
        super().__init__(num_imgs, config.emb_dim)
        self.test_emb = None

    def ray_indices(self, ray_bundle_metadata: Dict[str, Tensor], camera_indices: Optional[Tensor], n_rays: int,
                    device) -> Tensor:
        return ray_bundle_metadata["appearance_id"].reshape(-1).to(torch.int32)

    def forward(self, x: RaySamples, call_from_test=False):
        idxs = x.metadata["appearance_id"]
        return super().forward(idxs)

    def get_test_emb(self, x: RaySamples):
        mode = self.config.eval_mode
        n = len(x)
        dev = x.frustums.directions.device
        if mode == "zero":
            return torch.zeros((*x.frustums.directions.shape[:-1], self.out_dim), device=dev)
        if mode == "mean":
            return torch.ones((n, self.out_dim), device=dev) * self.mean(dim=0)
        assert self.test_emb is not None, "for deblur pretrain test only! need to init test_emb!"
        idxs = x.metadata["appearance_id"]
        return self.test_emb(idxs * 0)

    def init_test_params(self):
        if self.test_emb is not None or (self.embedding.weight.shape[0] <= 1):
            return
        self.test_emb = nn.Embedding(1, self.out_dim).to(self.embedding.weight.device)
        self.test_emb.weight = nn.Parameter(self.embedding(torch.tensor([21]).to(self.embedding.weight.device)))

    def get_emb_dim(self):
        return self.out_dim


class GlobalEmbedding(EvsFrameEmbedding):
    """R:lse_nerf/lse_embeddings.py:73-85: one row, index camera_indices*0."""

    def __init__(self, config, num_imgs, num_dims) -> None:
        super().__init__(config, 1, num_dims)

    def ray_indices(self, ray_bundle_metadata, camera_indices, n_rays, device) -> Tensor:
        # every ray reads row 0: a prefix of one zero vector per device (a fresh torch.zeros is a ~6 us fill per step).  A vector
        # handed out once is NEVER freed: a captured graph bakes its address into the row-bias / embedding-gradient kernels and
        # keeps no reference of its own, so a later call with another ray count must not hand that memory back to the allocator
        # (the next replay would gather and scatter embedding rows through whatever was allocated over it).  Longer requests add a
        # vector of at least twice the length and keep the old ones (geometric growth: at most 2x the largest request in total).
        # Never created inside a graph capture, whose allocations belong to the graph's pool.
        dev = torch.device(device)
        index = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else -1)
        kept = self.__dict__.setdefault("_zero_idx", {}).setdefault((dev.type, index), [])
        if kept and kept[-1].shape[0] >= n_rays:
            return kept[-1][:n_rays]
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return torch.zeros(n_rays, dtype=torch.int32, device=device)
        kept.append(torch.zeros(max(n_rays, 2 * kept[-1].shape[0] if kept else 0), dtype=torch.int32, device=device))
        return kept[-1][:n_rays]

    def forward(self, x: RaySamples, call_from_test=False):
        idxs = x.camera_indices * 0
        return Embedding.forward(self, idxs)

    def get_test_emb(self, x: RaySamples):
        return self.forward(x)


EMBEDDING_TYPE_DICT = {"global_emb": GlobalEmbedding, "evs_emb": EvsFrameEmbedding}


@dataclass
class LSEEmbeddingConfig:
    """R:lse_nerf/lse_embeddings.py:94-107."""
    embedding_type: str = "global_emb"
    metadata: str = "dummy"
    emb_dim: int = 32
    eval_mode: str = "zero"

    def setup(self, **kwargs):
        return EMBEDDING_TYPE_DICT[self.embedding_type.lower()](self, **kwargs)


# ----------------------------------------------------------------------------------------------------
# the field
# ----------------------------------------------------------------------------------------------------
def _contraction_mode(spatial_distortion) -> Optional[str]:
    """Map the reference's ``spatial_distortion`` argument onto the fused position kernel's modes.

    Accepted: ``None`` (aabb normalisation, R:lse_nerf/lse_field.py:271), the string ``"inf"``, or any object shaped like
    nerfstudio's ``SceneContraction`` whose ``order`` is infinity -- what R:lse_nerf/lsenerf.py:163-166 constructs.  The
    object itself is never called: contraction + ``(x + 2) / 4`` + selector run inside ``lse_positions_fwd``."""
    if spatial_distortion is None:
        return None
    if isinstance(spatial_distortion, str):
        if spatial_distortion == "inf":
            return "inf"
        raise ValueError(f"spatial_distortion '{spatial_distortion}': only 'inf' (L-infinity scene contraction) is implemented")
    order = getattr(spatial_distortion, "order", None)
    if order is not None and float(order) == float("inf"):
        return "inf"
    raise NotImplementedError(
        f"spatial_distortion {type(spatial_distortion).__name__}(order={order!r}): the HIP position kernel implements the "
        "L-infinity SceneContraction the reference constructs (R:lse_nerf/lsenerf.py:166) and None")


class LSEField(nn.Module):
    """R:lse_nerf/lse_field.py:94-360: same constructor signature, buffers and methods.

    ``spatial_distortion``: a nerfstudio-style ``SceneContraction(order=inf)`` object (what R:lse_nerf/lsenerf.py:166
    builds), the string ``"inf"``, or ``None`` (aabb normalisation).  ``implementation``: ``"tcnn"`` (the reference's
    value, R:lse_nerf/lsenerf.py:175) and ``"hip"`` both select the gfx950 kernels with tcnn's parameter layouts and
    numerics; ``"torch"`` (nerfstudio's CPU fallback with different layouts) is the oracle's domain and raises.  The heads
    the reference's model leaves switched off (transient / semantics / predicted normals, R:lse_nerf/lsenerf.py:168-176) are
    served by ``get_outputs`` / ``forward`` like the reference's (R:lse_nerf/lse_field.py:313-345): tcnn-layout MLPs on the
    device's library GEMMs, off the fused render path, which returns RGB and density only."""

    def __init__(self, aabb: Tensor, num_images: int, num_layers: int = 2, hidden_dim: int = 64, geo_feat_dim: int = 15,
                 num_levels: int = 16, base_res: int = 16, max_res: int = 2048, log2_hashmap_size: int = 19,
                 num_layers_color: int = 3, num_layers_transient: int = 2, features_per_level: int = 2,
                 hidden_dim_color: int = 64, hidden_dim_transient: int = 64, appearance_embedding_dim: int = 32,
                 embd_config: Optional[LSEEmbeddingConfig] = None, transient_embedding_dim: int = 16,
                 use_transient_embedding: bool = False, use_semantics: bool = False, num_semantic_classes: int = 100,
                 pass_semantic_gradients: bool = False, use_pred_normals: bool = False,
                 use_average_appearance_embedding: bool = False, spatial_distortion=None,
                 average_init_density: float = 1.0, implementation: str = "tcnn") -> None:
        super().__init__()
        if implementation not in ("tcnn", "hip"):
            raise NotImplementedError(f"implementation='{implementation}': this field runs the gfx950 kernels with tcnn "
                                      "semantics ('tcnn' / 'hip'); nerfstudio's torch fallback is restated in oracle/ only")
        assert geo_feat_dim == 15, "the fused head kernel assumes 1 + 15 base outputs"
        self.register_buffer("aabb", aabb.float())
        self.geo_feat_dim = geo_feat_dim
        # state-dict keys of the reference (R:lse_nerf/lse_field.py:158-160)
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.spatial_distortion = spatial_distortion
        self._contraction = _contraction_mode(spatial_distortion)
        self.num_images = num_images
        self.average_init_density = average_init_density
        self.appearance_embedding_dim = appearance_embedding_dim
        self.use_average_appearance_embedding = use_average_appearance_embedding
        self.use_transient_embedding, self.use_semantics, self.use_pred_normals = \
            bool(use_transient_embedding), bool(use_semantics), bool(use_pred_normals)
        self.pass_semantic_gradients = pass_semantic_gradients
        self.base_res = base_res
        self.step = 0
        if self.appearance_embedding_dim > 0:
            embd_config = embd_config or LSEEmbeddingConfig()
            self.embedding_appearance = embd_config.setup(num_imgs=num_images, num_dims=appearance_embedding_dim)
            self.appearance_embedding_dim = self.embedding_appearance.get_emb_dim()
            if not 0 < self.appearance_embedding_dim <= 97:
                raise ValueError(f"appearance_embedding_dim {self.appearance_embedding_dim}: the per-ray feature kernels take 1 .. 97 "
                                 "(head input 16 + 15 + emb padded to at most 128 columns)")
        else:
            self.embedding_appearance = None

        self.mlp_base_grid = Ed_HashEncoding(num_levels=num_levels, min_res=base_res, max_res=max_res,
                                             log2_hashmap_size=log2_hashmap_size, features_per_level=features_per_level)
        self.mlp_base_mlp = MLP(in_dim=self.mlp_base_grid.get_out_dim(), num_layers=num_layers, layer_width=hidden_dim,
                                out_dim=1 + geo_feat_dim, out_activation=None, in_layout=_lib.LSE_IN_LEVELMAJOR)
        self.mlp_head = MLP(in_dim=16 + geo_feat_dim + self.appearance_embedding_dim, num_layers=num_layers_color,
                            layer_width=hidden_dim_color, out_dim=3, out_activation="Sigmoid")
        # side heads (R:lse_nerf/lse_field.py:190-252; off in every LSENeRF preset): same modules, names and parameter layouts
        self.position_encoding = FrequencyEncoding(in_dim=3, num_frequencies=2)
        if self.use_transient_embedding:
            self.transient_embedding_dim = transient_embedding_dim
            self.embedding_transient = Embedding(num_images, transient_embedding_dim)
            self.mlp_transient = DenseMLP(in_dim=geo_feat_dim + transient_embedding_dim, num_layers=num_layers_transient,
                                          layer_width=hidden_dim_transient, out_dim=hidden_dim_transient)
            self.field_head_transient_uncertainty = FieldHead(1, FieldHeadNames.UNCERTAINTY, hidden_dim_transient, nn.Softplus())
            self.field_head_transient_rgb = FieldHead(3, FieldHeadNames.TRANSIENT_RGB, hidden_dim_transient, nn.Sigmoid())
            self.field_head_transient_density = FieldHead(1, FieldHeadNames.TRANSIENT_DENSITY, hidden_dim_transient, nn.Softplus())
        if self.use_semantics:
            self.mlp_semantics = DenseMLP(in_dim=geo_feat_dim, num_layers=2, layer_width=64, out_dim=hidden_dim_transient)
            self.field_head_semantics = FieldHead(num_semantic_classes, FieldHeadNames.SEMANTICS, hidden_dim_transient, None)
        if self.use_pred_normals:
            self.mlp_pred_normals = DenseMLP(in_dim=geo_feat_dim + self.position_encoding.get_out_dim(), num_layers=3,
                                             layer_width=64, out_dim=hidden_dim_transient)
            self.field_head_pred_normals = PredNormalsFieldHead(in_dim=hidden_dim_transient)
        self._aabb6 = None
        self.reuse_prepass = True  # the main pass re-uses the visibility pre-pass's positions / hash features of the survivors
        self._prepass = None
        self._h_ref = None        # weak reference to the last base-MLP output (get_density -> get_outputs hand-over)

    # -- helpers ---------------------------------------------------------------------------------------
    def _aabb_list(self):
        if self._aabb6 is None:
            self._aabb6 = [float(v) for v in self.aabb.flatten().tolist()]
        return self._aabb6

    def _x01(self, rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, n_dev=None):
        contraction = self._contraction == "inf"
        return ops.positions(rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, contraction,
                             None if contraction else self._aabb_list(), n_dev=n_dev)

    def prepass_sigma_fn(self, origins: Tensor, directions: Tensor) -> Callable:
        """``sigma_fn`` of the sampler's visibility pre-pass (nerfstudio VolumetricSampler.get_sigma_fn) on packed samples.
        The returned callable also has ``on_cull``: the estimator calls it with the visibility mask, and the positions /
        selector / hash features the pre-pass computed for the SURVIVING samples are compacted and parked on the field, where
        the main pass of the same step picks them up (``density_packed``) instead of encoding those samples again
        (``reuse_prepass``; the reference evaluates hash grid + base MLP twice on every survivor)."""
        fld = self
        state = {}

        def sigma_fn(t_starts, t_ends, ray_indices, n_dev=None):
            """``n_dev``: device-side count of the candidates (deferred sampling: the arrays then have capacity extent)."""
            ri = ray_indices if ray_indices.dtype == torch.int32 else ray_indices.to(torch.int32)
            with torch.no_grad():
                x01, sel = fld._x01(origins, directions, ri, t_starts, t_ends, None, n_dev)
                y = fld.mlp_base_grid.forward_levelmajor(x01, n_dev)
                sigma = fld._base_mlp(y, sel, x01.shape[0], n_dev)[1]
            state.update(x01=x01, sel=sel, y=y)
            return sigma

        def on_cull(mask, packed_info, new_packed_info, new_ray_indices, new_t_starts, new_t_ends):
            if not (fld.reuse_prepass and state):
                return
            n_new = new_t_starts.shape[0]
            x01, sel, y = ops.compact_features(mask, packed_info, new_packed_info, n_new, state["x01"], state["sel"], state["y"])
            state.clear()
            # keyed on the survivors' own buffers (kept alive here, so the addresses cannot be recycled) and the ray bundle
            fld._prepass = {"key": (new_t_starts.data_ptr(), new_ray_indices.data_ptr(), n_new, origins.data_ptr(),
                                    directions.data_ptr()),
                            "hold": (new_t_starts, new_ray_indices, new_t_ends), "x01": x01, "sel": sel, "y": y,
                            "version": fld.mlp_base_grid.params._version}

        sigma_fn.on_cull = on_cull
        return sigma_fn

    def _base_mlp(self, y: Tensor, sel: Tensor, n: int, n_dev: Optional[Tensor] = None):
        """(h[N,16], sigma[N]) = base MLP on level-major hash features with the fused trunc_exp density head."""
        mlp = self.mlp_base_mlp
        dens = (sel, self.average_init_density)    # trunc_exp density head fused into the MLP epilogue / backward
        if mlp.in_pad == mlp.in_dim:
            return ops.fused_mlp(mlp.params, y, mlp.meta(), n, density=dens, n_dev=n_dev)
        if n_dev is not None:
            raise _lib.LseHipError("a device-side sample count needs the 32-input base MLP (L * F == 32)")
        # small grids (L*F < 16): tcnn's ones-padding columns act as a bias shared by every sample
        kparams, bias = mlp.split_padding()
        idx = torch.zeros(n, dtype=torch.int32, device=y.device)
        seg = torch.tensor([[0, n]], dtype=torch.int64, device=y.device)
        return ops.fused_mlp(kparams, y, mlp.meta(mlp.in_dim), n, bias, idx, seg, density=dens)

    def _take_prepass(self, rays_o, rays_d, ray_idx, t_starts):
        """The parked pre-pass features if they belong to exactly these samples and these parameters, else None."""
        pp, self._prepass = self._prepass, None
        if pp is None or ray_idx is None or rays_d is None:
            return None
        key = (t_starts.data_ptr(), ray_idx.data_ptr(), t_starts.shape[0], rays_o.data_ptr(), rays_d.data_ptr())
        if key != pp["key"] or pp["version"] != self.mlp_base_grid.params._version:
            return None
        return pp

    def density_packed(self, rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, n_dev: Optional[Tensor] = None
                       ) -> Tuple[Tensor, Tensor, Tensor]:
        """Fast path of get_density on packed samples: returns (sigma[N], h[N,16] base-MLP output, selector[N]).
        ``n_dev``: device-side sample count (int64 [1]) when the arrays have capacity extent (deferred sampling)."""
        pp = self._take_prepass(rays_o, rays_d, ray_idx, t_starts) if t_starts is not None else None
        if pp is not None:   # survivors of this step's visibility pre-pass: positions + hash features are already there
            x01, sel = ops.positions(rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, self._contraction == "inf",
                                     None if self._contraction == "inf" else self._aabb_list(), precomputed=(pp["x01"], pp["sel"]),
                                     n_dev=n_dev)
            y = ops.hash_encode(x01, self.mlp_base_grid.params, self.mlp_base_grid.meta, precomputed=pp["y"], n_dev=n_dev)
            h, sigma = self._base_mlp(y, sel, x01.shape[0], n_dev)
            return sigma, h, sel
        x01, sel = self._x01(rays_o, rays_d, ray_idx, t_starts, t_ends, packed_info, n_dev)
        y = self.mlp_base_grid.forward_levelmajor(x01, n_dev)
        h, sigma = self._base_mlp(y, sel, x01.shape[0], n_dev)
        return sigma, h, sel

    def rgb_packed(self, h: Tensor, rays_d: Tensor, emb_idx: Optional[Tensor], ray_idx: Optional[Tensor],
                   packed_info: Optional[Tensor], emb_table: Optional[Tensor], n_dev: Optional[Tensor] = None) -> Tensor:
        """Fast path of get_outputs: h[N,16] from ``density_packed``; per-ray directions/embedding ids.
        Returns the compact head output [N,4] (columns 0..2 = RGB).

        tcnn's head input is [SH16 | geo15 | emb | ones-pad] against W_in[width, in_pad].  The SH / embedding / padding
        columns depend only on the ray: ``ops.ray_bias`` turns them into a per-ray layer-0 bias in one launch.  The 15
        geometry columns are per sample: the fused MLP reads h[N,16] (column 0 = density logit) against columns 15..30 of
        W_in IN PLACE (first-layer view: leading dimension in_pad, column offset 15, column 0 masked) -- the parameter
        vector is never split or copied, and both kernels accumulate their weight gradients straight into its .grad."""
        head = self.mlp_head
        n = h.shape[0]
        emb_dim = 0 if emb_table is None else emb_table.shape[1]
        if emb_dim != self.appearance_embedding_dim:        # eval mode "zero": the embedding columns meet zeros
            emb_table = torch.zeros((1, self.appearance_embedding_dim), dtype=h.dtype, device=h.device) \
                if self.appearance_embedding_dim > 0 else None
            emb_idx = torch.zeros(rays_d.shape[0], dtype=torch.int32, device=h.device) if emb_table is not None else None
        row_bias = ops.ray_bias(rays_d, emb_table, emb_idx, head.params, head.layer_width)        # [R, W]
        meta = ops.MlpMeta(16, head.layer_width, head.n_hidden_layers, head.out_act, _lib.LSE_IN_ROWMAJOR,
                           w0_ld=head.in_pad, w0_col=15, w0_mask_col0=1)
        return ops.fused_mlp(head.params, h, meta, n, row_bias, ray_idx, packed_info, out_cols=4, n_dev=n_dev)

    def _train_emb_table(self):
        if self.embedding_appearance is None:
            return None
        return self.embedding_appearance.embedding.weight

    def _eval_emb(self, n_rays: int, device):
        """(table, idx) implementing get_test_emb for the per-ray feature kernel."""
        emb = self.embedding_appearance
        if emb is None:
            return None, None
        if isinstance(emb, GlobalEmbedding):
            return emb.embedding.weight, torch.zeros(n_rays, dtype=torch.int32, device=device)
        mode = emb.config.eval_mode
        if mode == "zero":
            return None, None
        if mode == "mean":
            return emb.mean(dim=0)[None, :].contiguous(), torch.zeros(n_rays, dtype=torch.int32, device=device)
        assert emb.test_emb is not None, "for deblur pretrain test only! need to init test_emb!"
        return emb.test_emb.weight, torch.zeros(n_rays, dtype=torch.int32, device=device)

    # -- reference interface ---------------------------------------------------------------------------
    @staticmethod
    def _packed_view(ray_samples):
        """(ray_bundle, ray_indices, packed_info) when the samples come from lsenerf_amd's VolumetricSampler (packed,
        ray-sorted, with the owning bundle attached); (None, None, None) for a stock nerfstudio ``RaySamples``, which
        then takes the generic per-sample path."""
        rb = getattr(ray_samples, "ray_bundle", None)
        ridx = getattr(ray_samples, "ray_indices", None)
        if rb is None or ridx is None:
            return None, None, None
        return rb, ridx, getattr(ray_samples, "packed_info", None)

    def get_density(self, ray_samples: RaySamples) -> Tuple[Tensor, Tensor]:
        """R:lse_nerf/lse_field.py:264-288 -> (density [N,1], geo features [N,15])."""
        fr = ray_samples.frustums
        rb, ridx, pinfo = self._packed_view(ray_samples)
        if rb is not None:
            sigma, h, _ = self.density_packed(rb.origins.contiguous(), rb.directions.contiguous(), ridx,
                                              fr.starts.reshape(-1), fr.ends.reshape(-1), pinfo)
        else:
            pos = fr.get_positions().reshape(-1, 3).contiguous()
            sigma, h, _ = self.density_packed(pos, None, None, None, None, None)
        geo = h[:, 1:1 + self.geo_feat_dim]
        shape = fr.shape
        if len(shape) != 1:
            geo = geo.reshape(*shape, self.geo_feat_dim)
        self._h_ref = weakref.ref(h)      # get_outputs recognises `geo` as a view of h (by storage) and skips a copy
        return sigma.view(*shape, 1), geo

    def _full_base_output(self, density_embedding: Tensor, n: int) -> Tensor:
        """The [N,16] head input.  If ``density_embedding`` is (a reshape of) the ``h[:, 1:16]`` view get_density returned,
        h itself is used -- checked on the storage, so any copy or arithmetic on the caller's side simply falls through
        to the padded copy below; nothing rides on Python attributes of the tensor."""
        h = self._h_ref() if self._h_ref is not None else None
        de = density_embedding
        if h is not None and h.shape[0] == n and de.dtype == h.dtype and de.device == h.device \
                and de.untyped_storage().data_ptr() == h.untyped_storage().data_ptr() \
                and de.storage_offset() == h.storage_offset() + 1 and de.numel() == n * self.geo_feat_dim \
                and de.stride(-1) == 1 and all(st % 16 == 0 for st in de.stride()[:-1]) \
                and de.reshape(n, self.geo_feat_dim).stride() == (16, 1):
            return h
        return torch.cat([torch.zeros(n, 1, device=de.device, dtype=de.dtype), de.reshape(n, self.geo_feat_dim)],
                         dim=1).contiguous()

    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None) -> Dict[FieldHeadNames, Tensor]:
        """R:lse_nerf/lse_field.py:290-360 -> {RGB: [N,3]}."""
        assert density_embedding is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        fr = ray_samples.frustums
        n = fr.directions.reshape(-1, 3).shape[0]
        dev = fr.directions.device
        h = self._full_base_output(density_embedding, n)
        rb, ridx, pinfo = self._packed_view(ray_samples)
        if rb is not None:
            dirs, n_rays = rb.directions.contiguous(), len(rb)
            meta, cams = rb.metadata, rb.camera_indices
        else:   # stock RaySamples: arbitrary per-sample directions / ids -- every sample is its own "ray"
            dirs, ridx, pinfo, n_rays = fr.directions.reshape(-1, 3).contiguous(), None, None, n
            meta, cams = (getattr(ray_samples, "metadata", None) or {}), ray_samples.camera_indices
        if self.embedding_appearance is None:
            table, eidx = None, None
        elif self.training:
            table = self._train_emb_table()
            eidx = self.embedding_appearance.ray_indices(meta, cams, n_rays, dev).contiguous()
        else:
            table, eidx = self._eval_emb(n_rays, dev)
        outputs = self._side_heads(ray_samples, density_embedding, n, ridx)
        out16 = self.rgb_packed(h, dirs, eidx, ridx, pinfo, table)
        rgb = out16[:, :3].view(*fr.directions.shape[:-1], 3)
        outputs[FieldHeadNames.RGB] = rgb
        return outputs

    def _side_heads(self, ray_samples, density_embedding: Tensor, n: int, ridx: Optional[Tensor]) -> Dict[FieldHeadNames, Tensor]:
        """R:lse_nerf/lse_field.py:313-345: transient (training only), semantics and predicted-normal outputs of the heads that
        are switched on; ``{}`` for the LSENeRF presets."""
        outputs: Dict[FieldHeadNames, Tensor] = {}
        if not (self.use_transient_embedding or self.use_semantics or self.use_pred_normals):
            return outputs
        fr = ray_samples.frustums
        shape = fr.directions.shape[:-1]
        geo = density_embedding.reshape(n, self.geo_feat_dim)
        if self.use_transient_embedding and self.training:
            cams = ray_samples.camera_indices.reshape(-1).long()
            if cams.shape[0] != n:        # packed samples carry one camera index per RAY
                cams = cams[ridx.long()]
            x = self.mlp_transient(torch.cat([geo, self.embedding_transient(cams)], dim=-1)).view(*shape, -1)
            outputs[FieldHeadNames.UNCERTAINTY] = self.field_head_transient_uncertainty(x)
            outputs[FieldHeadNames.TRANSIENT_RGB] = self.field_head_transient_rgb(x)
            outputs[FieldHeadNames.TRANSIENT_DENSITY] = self.field_head_transient_density(x)
        if self.use_semantics:
            x = self.mlp_semantics(geo if self.pass_semantic_gradients else geo.detach()).view(*shape, -1)
            outputs[FieldHeadNames.SEMANTICS] = self.field_head_semantics(x)
        if self.use_pred_normals:
            pos = self.position_encoding(fr.get_positions().reshape(n, 3))
            x = self.mlp_pred_normals(torch.cat([pos, geo], dim=-1)).view(*shape, -1)
            outputs[FieldHeadNames.PRED_NORMALS] = self.field_head_pred_normals(x)
        return outputs

    def density_fn(self, positions: Tensor) -> Tensor:
        """nerfstudio ``Field.density_fn`` (wired at R:lse_nerf/lsenerf.py:193): positions [..,3] -> density [..,1]."""
        p = positions.reshape(-1, 3).contiguous()
        sigma, _, _ = self.density_packed(p, None, None, None, None, None)
        return sigma.view(*positions.shape[:-1], 1)

    def forward(self, ray_samples: RaySamples) -> Dict[FieldHeadNames, Tensor]:
        """nerfstudio ``Field.forward``: get_density -> get_outputs -> add DENSITY."""
        density, geo = self.get_density(ray_samples)
        out = self.get_outputs(ray_samples, density_embedding=geo)
        out[FieldHeadNames.DENSITY] = density
        return out
