#!/usr/bin/env python3
"""Request floor of the hash backward as a function of the merge window: distinct 64-byte lines of the gradient table per window of
W consecutive samples (of one ray: 1024 samples per ray at the metric size), per level, on bench.py's headline positions.  W = 64 is
what one wave of the production kernel can merge; larger W prices a kernel that would carry its sector cache across W / 64 chunks."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from lsenerf_amd import ops
dev = torch.device("cuda", 0)
model, sets, _ = bench.build_workload(dev, 96)
rb, _, jitter = sets[0]
cfg = model.config
ri, ts, te, packed = model.occupancy_grid.sampling(rb.origins.detach(), rb.directions.detach(), near_plane=cfg.near_plane,
                                                   far_plane=cfg.far_plane, render_step_size=cfg.render_step_size, stratified=True,
                                                   jitter=jitter, return_packed=True)[:4]
x01 = ops.positions(rb.origins.detach(), rb.directions.detach(), ri, ts, te, packed, True, None)[0]
meta = model.field.mlp_base_grid.meta
n = x01.shape[0]
print(n, "samples")
rows = {}
for W in (64, 128, 256, 512):
    lines = []
    for l in range(16):
        sc = meta.scales[l]; res = meta.resolutions[l]; size = meta.offsets[l + 1] - meta.offsets[l]
        p0 = (x01 * sc + 0.5).floor().to(torch.int64)
        acc = []
        for c in range(8):
            q = p0 + torch.tensor([c & 1, (c >> 1) & 1, (c >> 2) & 1], device=dev)
            if res ** 3 <= size:
                idx = q[:, 0] + q[:, 1] * res + q[:, 2] * res * res
            else:
                idx = (q[:, 0] ^ (q[:, 1] * 2654435761) ^ (q[:, 2] * 805459861)) & 0xFFFFFFFF
            acc.append((idx % size) >> 3)
        ln = torch.stack(acc, 1)
        pad = (-n) % W
        if pad:
            ln = torch.cat([ln, ln[-1:].expand(pad, 8)])
        w = ln.reshape(-1, W * 8).sort(dim=1).values
        lines.append(float(((w[:, 1:] != w[:, :-1]).sum() + w.shape[0]) / n))
    rows[W] = lines
    print(f"W = {W:4d}: per level " + " ".join(f"{v:.2f}" for v in lines) + f"   sum {sum(lines):.2f}", flush=True)
