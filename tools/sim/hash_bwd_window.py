#!/usr/bin/env python3
"""CPU estimate (no GPU): how many distinct 64-byte lines of the gradient table does a window of W consecutive samples of
the M-march workload touch, per level?  = atomic requests per sample if the hash backward merged everything inside a window
of W samples before it sent float atomics (today: W = 64, one wave).  usage: python tools/sim/hash_bwd_window.py [rays]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from oracle import field as ofield, hashgrid as ohash

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = torch.Generator().manual_seed(0)
o, d = bench.sphere_rays(R, g)
aabb = torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
# chord of each ray inside the box; 1024 samples per ray on average (M-march: constant step)
inv = 1.0 / d
t0 = (aabb[0] - o) * inv
t1 = (aabb[1] - o) * inv
tn = torch.minimum(t0, t1).amax(-1).clamp_min(0.05)
tf = torch.maximum(t0, t1).amin(-1)
step = float((tf - tn).clamp_min(0).mean() / 1024)
meta = ohash.tcnn_grid_meta()
res = {}
for W in (64, 128, 256, 512, 1024):
    res[W] = np.zeros(meta.n_levels)
total = 0
for r in range(R):
    n = int((tf[r] - tn[r]) / step)
    if n <= 0:
        continue
    t = tn[r] + (torch.arange(n) + 0.5) * step
    p = o[r] + d[r] * t[:, None]
    x01 = ofield.normalized_positions(p, aabb)
    total += n
    for l in range(meta.n_levels):
        idx = ohash.tcnn_corner_indices(x01, meta, l).numpy().astype(np.int64)     # [n, 8] entry index inside the level
        line = idx >> 3                                                            # 8 entries x 8 B = one 64-B line
        for W in res:
            for s in range(0, n, W):
                res[W][l] += np.unique(line[s:s + W]).size
print(f"{R} rays, {total} samples, step {step:.5f}")
print("level " + " ".join(f"W={W:5d}" for W in res))
for l in range(meta.n_levels):
    print(f"{l:5d} " + " ".join(f"{res[W][l] / total:7.3f}" for W in res))
print("sum   " + " ".join(f"{res[W].sum() / total:7.3f}" for W in res))
