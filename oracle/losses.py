"""Oracle: output routing (intensity mappers) and losses of LSENeRFModel -- R:lse_nerf/lsenerf.py:329-439,
R:lse_nerf/intensity_mappers.py:28-94, R:lse_nerf/utils.py:12 (EPS).  TEST INFRASTRUCTURE, plain torch on CPU.
These are O(rays) element-wise ops; they define what "rendered RGB / log-intensity" means in the parity statement."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS = 1e-6


def mlp_mapper(params):
    """The "mlp" / "rgb_mlp" intensity mappers, R:lse_nerf/intensity_mappers.py:28-62: nerfstudio 0.3.2
    ``MLP(in_dim, num_layers=4, layer_width=16, out_dim, activation=ReLU, out_activation=Sigmoid, implementation="torch")`` =
    Linear(in,16) -> ReLU -> Linear(16,16) -> ReLU -> Linear(16,16) -> ReLU -> Linear(16,out) -> Sigmoid, applied to the last axis.
    ``params`` = [W0, b0, W1, b1, W2, b2, W3, b3] in nn.Linear layout ([out, in]).  (The reference fits it to the identity at
    construction, :8-26; the fit is host-side set-up, not part of the path.)"""
    def apply(x):
        for l in range(4):
            x = x @ params[2 * l].T + params[2 * l + 1]
            x = torch.relu(x) if l < 3 else torch.sigmoid(x)
        return x
    return apply


def to_gray(x):
    return (x * x.new_tensor([0.2989, 0.5870, 0.1140])).sum(-1, keepdim=True)


def route_outputs(out_rgb: torch.Tensor, *, training: bool, use_mapping: bool, map_mode: str, ev_out: bool,
                  rgb_loss_type: str, rgb_mapper=lambda x: x, evs_mapper=None, three_to_one_w: Optional[torch.Tensor] = None
                  ) -> Dict[str, torch.Tensor]:
    """R:lse_nerf/lsenerf.py:329-377.  ``three_to_one_w``: the ThreeToOne parameter [1,3] (softmax-normalised inside)."""
    d = {"rgb": out_rgb}

    def one_dim(x):
        if three_to_one_w is None:
            return x
        return F.linear(x, F.softmax(three_to_one_w, dim=-1), None)

    clamp_out = torch.clamp(out_rgb, 1e-5)
    if use_mapping or map_mode == "rgb_evs":
        if map_mode == "rgb_evs":
            if ev_out or not training:
                d["ev_out"] = rgb_mapper(one_dim(clamp_out))
                d["linear"] = torch.cat([d["ev_out"]] * 3, -1) if d["ev_out"].shape[-1] == 1 else d["ev_out"]
        elif map_mode == "evs_rgb":
            d["ev_out"] = one_dim(clamp_out)
            d["linear"] = clamp_out
            d["rgb"] = rgb_mapper(d["linear"])
        elif map_mode == "co_map":
            d["rgb"] = rgb_mapper(clamp_out)
            if ev_out or not training:
                ev_linear = one_dim(clamp_out)
                d["linear"], d["ev_linear"] = clamp_out, ev_linear
                d["ev_out"] = evs_mapper(ev_linear)
    if rgb_loss_type == "deblur" and training:
        if d["rgb"].shape[0] % 4 == 0:
            d["rgb"] = d["rgb"].reshape(-1, 4, 3).mean(dim=1)
    d["rgb"] = torch.clamp(d["rgb"], 1e-5) if training else torch.clamp(d["rgb"], 0, 1)
    return d


def log_loss(evs, prev_rad, next_rad):
    """R:lse_nerf/lsenerf.py:392-399."""
    if prev_rad.shape[-1] != 1:
        prev_rad, next_rad = to_gray(prev_rad), to_gray(next_rad)
    return F.mse_loss(torch.log(next_rad + EPS) - torch.log(prev_rad + EPS), evs)


def enerf_norm_loss(evs, prev_rad, next_rad, e_thresh):
    """R:lse_nerf/lsenerf.py:406-419: both sides of the event MSE divided by their 2-norm over the rays (dim 0); the event side is
    first divided by the event threshold and carries no gradient."""
    if prev_rad.shape[-1] != 1:
        prev_rad, next_rad = to_gray(prev_rad), to_gray(next_rad)
    delta_log = torch.log(next_rad + EPS) - torch.log(prev_rad + EPS)
    log_norm = torch.linalg.norm(delta_log, dim=0, keepdim=True) + EPS
    with torch.no_grad():
        evs = evs / e_thresh
        evs_norm = torch.linalg.norm(evs, dim=0, keepdim=True) + EPS
    return F.mse_loss(delta_log / log_norm, evs / evs_norm)


def loss_dict(col_out, prev_out, next_out, col_gt, evs_gt, *, use_mapping: bool, evs_loss_weight: float = 1.0,
              event_loss: str = "log_loss", e_thresh=None):
    """R:lse_nerf/lsenerf.py:422-439."""
    out = {}
    if col_out is not None:
        out["rgb_loss"] = F.mse_loss(col_gt, col_out["rgb"])
    if prev_out is not None:
        key = "ev_out" if use_mapping else "rgb"
        p, n = prev_out[key], next_out[key]
        evs = evs_gt if p.shape[-1] == 1 else torch.cat([evs_gt] * 3, -1)
        out["event_loss"] = evs_loss_weight * (log_loss(evs, p, n) if event_loss == "log_loss" else enerf_norm_loss(evs, p, n, e_thresh))
    return out
