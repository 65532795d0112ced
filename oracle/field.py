"""Oracle: MLPs, SH-4, trunc_exp, L-inf contraction and the LSEField wiring.  TEST INFRASTRUCTURE.

Follows R:lse_nerf/lse_field.py:124-360 for the wiring and SURVEY.md App. A.3/A.4 for the upstream
arithmetic (nerfstudio 0.3.2 MLP / SHEncoding / trunc_exp / SceneContraction, tcnn 1.7 networks).
Parity unpinned (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import hashgrid as hg


# --------------------------------------------------------------------------------------------
# small pieces (App. A.4)
# --------------------------------------------------------------------------------------------
class _TruncExp(torch.autograd.Function):
    """nerfstudio ``trunc_exp``: fwd exp(x); bwd g * exp(clamp(x, -15, 15)).  R:lse_nerf/lse_field.py:286."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def contract_inf(x: torch.Tensor) -> torch.Tensor:
    """nerfstudio ``SceneContraction(order=inf)`` (R:lse_nerf/lsenerf.py:166)."""
    mag = torch.linalg.norm(x, ord=float("inf"), dim=-1)[..., None]
    return torch.where(mag < 1, x, (2 - (1 / mag)) * (x / mag))


def normalized_positions(x: torch.Tensor, aabb: torch.Tensor) -> torch.Tensor:
    """``SceneBox.get_normalized_positions`` (R:lse_nerf/lse_field.py:271)."""
    lengths = aabb[1] - aabb[0]
    return (x - aabb[0]) / lengths


def frustum_positions(origins, directions, starts, ends):
    """nerfstudio ``Frustums.get_positions``: origins + directions * (starts + ends) / 2."""
    return origins + directions * (starts + ends) / 2


SH_C = dict(
    c0=0.28209479177387814, c1=0.4886025119029199, c4=1.0925484305920792, c6a=0.9461746957575601,
    c6b=0.31539156525251999, c8=0.5462742152960396, c9=0.5900435899266435, c10=2.890611442640554,
    c11=0.4570457994644658, c12=0.3731763325901154, c14=1.445305721320277,
)


def sh4_nerfstudio(d: torch.Tensor) -> torch.Tensor:
    """nerfstudio ``components_from_spherical_harmonics(levels=4)`` evaluated at ``d`` (torch path)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    xx, yy, zz = x * x, y * y, z * z
    C = SH_C
    comps = [
        torch.full_like(x, C["c0"]),
        C["c1"] * y, C["c1"] * z, C["c1"] * x,
        C["c4"] * x * y, C["c4"] * y * z, C["c6a"] * zz - C["c6b"], C["c4"] * x * z, C["c8"] * (xx - yy),
        C["c9"] * y * (3 * xx - yy), C["c10"] * x * y * z, C["c11"] * y * (5 * zz - 1),
        C["c12"] * z * (5 * zz - 3), C["c11"] * x * (5 * zz - 1), C["c14"] * z * (xx - yy),
        C["c9"] * x * (xx - 3 * yy),
    ]
    return torch.stack(comps, dim=-1)


def sh4_tcnn(d01: torch.Tensor) -> torch.Tensor:
    """tcnn ``SphericalHarmonics`` degree 4 on inputs in [0,1] (re-centred ``2x-1``), Condon-Shortley signs."""
    v = d01 * 2 - 1
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
    comps = [
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x,
        1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2,
        0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ]
    return torch.stack(comps, dim=-1)


# --------------------------------------------------------------------------------------------
# MLPs (App. A.3)
# --------------------------------------------------------------------------------------------
def _pad16(n: int) -> int:
    return ((n + 15) // 16) * 16


class TcnnMLP:
    """tcnn 1.7 ``Network`` as nerfstudio's ``MLP.get_tcnn_network_config`` configures it.

    Bias-free; weights row-major [out,in] per layer in one flat ``params``; the input is padded to a
    multiple of 16 *with ones* (tcnn's Identity input encoding pads with 1 so that the first layer can
    learn a bias) and the output to 16 (padding outputs are sliced off by the binding).
    ``num_layers`` counts linear layers as nerfstudio does: n_hidden_layers = num_layers - 1.
    """

    def __init__(self, in_dim, num_layers, layer_width, out_dim, out_activation: Optional[str]):
        self.in_dim, self.out_dim = in_dim, out_dim
        self.in_pad, self.out_pad = _pad16(in_dim), _pad16(out_dim)
        self.width = layer_width
        self.n_hidden_layers = num_layers - 1
        self.out_activation = out_activation
        self.shapes = [(layer_width, self.in_pad)]
        self.shapes += [(layer_width, layer_width)] * (self.n_hidden_layers - 1)
        self.shapes += [(self.out_pad, layer_width)]
        self.n_params = sum(a * b for a, b in self.shapes)

    def init_params(self, generator=None) -> torch.Tensor:
        """tcnn xavier_uniform per matrix."""
        chunks = []
        for (o, i) in self.shapes:
            s = math.sqrt(6.0 / (i + o))
            chunks.append(((torch.rand(o * i, generator=generator) * 2 - 1) * s))
        return torch.cat(chunks)

    def matrices(self, params: torch.Tensor) -> List[torch.Tensor]:
        out, off = [], 0
        for (o, i) in self.shapes:
            out.append(params[off:off + o * i].view(o, i))
            off += o * i
        return out

    def forward(self, x: torch.Tensor, params: torch.Tensor, return_hidden=False):
        if x.shape[-1] < self.in_pad:
            x = torch.cat([x, torch.ones(*x.shape[:-1], self.in_pad - x.shape[-1], dtype=x.dtype)], dim=-1)
        Ws = self.matrices(params)
        hidden = []
        h = x
        for W in Ws[:-1]:
            h = torch.relu(h @ W.t())
            hidden.append(h)
        o = h @ Ws[-1].t()
        if self.out_activation == "Sigmoid":
            o = torch.sigmoid(o)
        o = o[..., :self.out_dim]
        return (o, hidden) if return_hidden else o


class TorchMLP(torch.nn.Module):
    """nerfstudio 0.3.2 ``MLP`` torch path: Linear layers with biases, ReLU between, out_activation last."""

    def __init__(self, in_dim, num_layers, layer_width, out_dim, out_activation: Optional[str]):
        super().__init__()
        dims = [in_dim] + [layer_width] * (num_layers - 1) + [out_dim]
        self.layers = torch.nn.ModuleList([torch.nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])
        self.out_activation = out_activation

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < len(self.layers) - 1:
                x = torch.relu(x)
        if self.out_activation == "Sigmoid":
            x = torch.sigmoid(x)
        return x


# --------------------------------------------------------------------------------------------
# LSEField wiring (R:lse_nerf/lse_field.py:124-360)
# --------------------------------------------------------------------------------------------
class FieldOracle:
    """``LSEField`` with ``implementation in {"tcnn","torch"}``, default-off heads omitted
    (R:lse_nerf/lse_field.py:143-148).  Parameters are plain leaf tensors in ``self.params``."""

    def __init__(self, implementation="tcnn", num_levels=16, base_res=16, max_res=2048,
                 log2_hashmap_size=19, features_per_level=2, hidden_dim=64, geo_feat_dim=15,
                 num_layers=2, num_layers_color=3, hidden_dim_color=64, appearance_embedding_dim=32,
                 num_embeddings=1, contraction=True, aabb=None, average_init_density=1.0, seed=96):
        g = torch.Generator().manual_seed(seed)
        self.impl = implementation
        self.geo_feat_dim = geo_feat_dim
        self.contraction = contraction
        self.aabb = aabb if aabb is not None else torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        self.average_init_density = average_init_density
        self.emb_dim = appearance_embedding_dim
        self.training = True
        in_head = 16 + geo_feat_dim + appearance_embedding_dim
        self.params: Dict[str, torch.Tensor] = {}
        if implementation == "tcnn":
            self.meta = hg.tcnn_grid_meta(num_levels, features_per_level, log2_hashmap_size, base_res,
                                          max_res=max_res)
            self.params["grid"] = hg.init_tcnn_table(self.meta, g)
            self.base = TcnnMLP(num_levels * features_per_level, num_layers, hidden_dim, 1 + geo_feat_dim, None)
            self.head = TcnnMLP(in_head, num_layers_color, hidden_dim_color, 3, "Sigmoid")
            self.params["base"] = self.base.init_params(g)
            self.params["head"] = self.head.init_params(g)
        else:
            self.meta = hg.torch_grid_meta(num_levels, base_res, max_res, log2_hashmap_size, features_per_level)
            self.params["grid"] = hg.init_torch_table(self.meta, generator=g)
            torch.manual_seed(seed)
            self.base = TorchMLP(num_levels * features_per_level, num_layers, hidden_dim, 1 + geo_feat_dim, None)
            self.head = TorchMLP(in_head, num_layers_color, hidden_dim_color, 3, "Sigmoid")
            for n, p in list(self.base.named_parameters()):
                self.params["base." + n] = p.data
            for n, p in list(self.head.named_parameters()):
                self.params["head." + n] = p.data
        if appearance_embedding_dim > 0:
            # nerfstudio Embedding -> nn.Embedding default init N(0,1)
            self.params["embedding"] = torch.randn(num_embeddings, appearance_embedding_dim, generator=g)
        for k in self.params:
            self.params[k] = self.params[k].clone().requires_grad_(True)
        if implementation == "torch":
            # re-point module parameters at the tracked leaves
            for n, p in list(self.base.named_parameters()):
                _set_param(self.base, n, self.params["base." + n])
            for n, p in list(self.head.named_parameters()):
                _set_param(self.head, n, self.params["head." + n])

    # -- encodings -------------------------------------------------------------------------
    def encode(self, x01: torch.Tensor) -> torch.Tensor:
        if self.impl == "tcnn":
            return hg.hash_encode_tcnn(x01, self.params["grid"], self.meta)
        return hg.hash_encode_torch(x01, self.params["grid"], self.meta)

    def mlp_base(self, enc):
        return self.base.forward(enc, self.params["base"]) if self.impl == "tcnn" else self.base(enc)

    def mlp_head(self, h):
        return self.head.forward(h, self.params["head"]) if self.impl == "tcnn" else self.head(h)

    # -- R:lse_nerf/lse_field.py:264-288 ---------------------------------------------------
    def normalize(self, positions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.contraction:
            p = contract_inf(positions)
            p = (p + 2.0) / 4.0
        else:
            p = normalized_positions(positions, self.aabb)
        selector = ((p > 0.0) & (p < 1.0)).all(dim=-1)
        p = p * selector[..., None]
        return p, selector

    def get_density(self, positions: torch.Tensor):
        p, selector = self.normalize(positions)
        h = self.mlp_base(self.encode(p.view(-1, 3)))      # mlp_base = Sequential(grid, mlp), R:lse_nerf/lse_field.py:208
        dba, geo = torch.split(h, [1, self.geo_feat_dim], dim=-1)
        density = self.average_init_density * trunc_exp(dba)
        density = density * selector[..., None]
        return density, geo

    def density_fn(self, positions: torch.Tensor) -> torch.Tensor:
        """nerfstudio ``Field.density_fn`` (wired at R:lse_nerf/lsenerf.py:193)."""
        return self.get_density(positions)[0]

    # -- R:lse_nerf/lse_field.py:290-360 ---------------------------------------------------
    def get_outputs(self, directions: torch.Tensor, geo: torch.Tensor, appearance_idx: Optional[torch.Tensor],
                    eval_emb_mode: str = "zero") -> torch.Tensor:
        d = (directions + 1.0) / 2.0                       # shift_directions_for_tcnn, :298
        if self.impl == "tcnn":
            sh = sh4_tcnn(d)
        else:
            with torch.no_grad():                          # nerfstudio torch SH path is no_grad
                sh = sh4_nerfstudio(d)
        parts = [sh, geo.view(-1, self.geo_feat_dim)]
        if self.emb_dim > 0:
            if self.training:
                emb = self.params["embedding"][appearance_idx.view(-1)]     # R:lse_nerf/lse_embeddings.py:36-41
            elif eval_emb_mode == "zero":
                emb = torch.zeros(directions.shape[0], self.emb_dim)        # :51-55
            else:
                emb = torch.ones(directions.shape[0], self.emb_dim) * self.params["embedding"].mean(dim=0)
            parts.append(emb)
        h = torch.cat(parts, dim=-1)
        return self.mlp_head(h)

    def parameters(self):
        return list(self.params.values())


# --------------------------------------------------------------------------------------------
# side heads of LSEField (R:lse_nerf/lse_field.py:190-252, 313-345; switched off by R:lse_nerf/lsenerf.py:168-176)
# --------------------------------------------------------------------------------------------
def frequency_encoding_tcnn(x: torch.Tensor, num_frequencies: int = 2) -> torch.Tensor:
    """nerfstudio ``NeRFEncoding(..., implementation="tcnn")`` (R:lse_nerf/lse_field.py:190-192) = tcnn ``Frequency``
    (tiny-cuda-nn, unpinned in the reference; restated from its published kernel ``frequency_encoding``): encoded feature ``j`` of
    a point reads input dimension ``j // (2 F)``, frequency exponent ``(j // 2) % F`` and phase ``(j % 2) * pi / 2``:
    ``sin(x * 2^e * pi + phase)``.  Loop form on purpose (the HIP-side module is vectorised)."""
    n, d = x.shape
    out = torch.empty(n, d * num_frequencies * 2, dtype=x.dtype)
    for j in range(d * num_frequencies * 2):
        i, e, ph = j // (2 * num_frequencies), (j // 2) % num_frequencies, (j % 2) * (math.pi / 2)
        out[:, j] = torch.sin(x[:, i] * (2.0 ** e) * math.pi + ph)
    return out


class SideHeadsOracle:
    """The three optional head groups on top of ``density_embedding`` (R:lse_nerf/lse_field.py:313-345), tcnn-layout MLPs +
    nerfstudio field heads (``nn.Linear`` + Softplus / Sigmoid / none / Tanh-then-normalise).  ``params`` maps the state-dict
    names of the reference's modules to tensors."""

    def __init__(self, params: Dict[str, torch.Tensor], geo_feat_dim=15, transient_embedding_dim=16, num_layers_transient=2,
                 hidden_dim_transient=64):
        self.p = params
        self.geo = geo_feat_dim
        self.mlp_transient = TcnnMLP(geo_feat_dim + transient_embedding_dim, num_layers_transient, hidden_dim_transient,
                                     hidden_dim_transient, None)                                         # :213-221
        self.mlp_semantics = TcnnMLP(geo_feat_dim, 2, 64, hidden_dim_transient, None)                    # :228-236
        self.mlp_pred_normals = TcnnMLP(geo_feat_dim + 12, 3, 64, hidden_dim_transient, None)            # :243-251

    def _head(self, name, x, act):
        y = x @ self.p[name + ".net.weight"].t() + self.p[name + ".net.bias"]
        return y if act is None else act(y)

    def transient(self, geo, camera_indices):                                                            # :313-326
        emb = self.p["embedding_transient.embedding.weight"][camera_indices.view(-1)]
        x = self.mlp_transient.forward(torch.cat([geo.view(-1, self.geo), emb], dim=-1), self.p["mlp_transient.params"])
        sp = torch.nn.functional.softplus
        return (self._head("field_head_transient_uncertainty", x, sp), self._head("field_head_transient_rgb", x, torch.sigmoid),
                self._head("field_head_transient_density", x, sp))

    def semantics(self, geo, pass_gradients=False):                                                      # :329-335
        inp = geo.view(-1, self.geo)
        if not pass_gradients:
            inp = inp.detach()
        return self._head("field_head_semantics", self.mlp_semantics.forward(inp, self.p["mlp_semantics.params"]), None)

    def pred_normals(self, positions, geo):                                                              # :338-345
        enc = frequency_encoding_tcnn(positions.view(-1, 3))
        x = self.mlp_pred_normals.forward(torch.cat([enc, geo.view(-1, self.geo)], dim=-1), self.p["mlp_pred_normals.params"])
        return torch.nn.functional.normalize(self._head("field_head_pred_normals", x, torch.tanh), dim=-1)


def _set_param(module: torch.nn.Module, name: str, value: torch.Tensor):
    parts = name.split(".")
    for p in parts[:-1]:
        module = getattr(module, p)
    del module._parameters[parts[-1]]
    setattr(module, parts[-1], value)
