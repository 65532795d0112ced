#!/usr/bin/env python3
"""Launch-size sweep of the two fused-MLP backward calls of the step (head with per-ray bias, base with density head): time = a + b N.
The intercept a is what a launch pays whatever its size: weight staging and the end-of-kernel flush of 2048 waves' weight-gradient
accumulators (float atomics of every wave onto the same ~600 lines).  usage: python tools/mlp_bwd_fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lsenerf_amd import _lib
from lsenerf_amd.field import LSEField

dev = torch.device("cuda", 0)
torch.manual_seed(0)
fld = LSEField(aabb=torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_images=4, spatial_distortion="inf").to(dev).train()
S = 1024
res = {}
for R in (256, 512, 1024, 2048, 4096):
    n = R * S
    g = torch.Generator().manual_seed(R)
    y = torch.randn(16, n, 2, generator=g).to(dev).requires_grad_(True)           # level-major hash features
    sel = torch.ones(n, dtype=torch.uint8, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    ri = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(dev)
    cnt = torch.full((R,), S, dtype=torch.long)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).to(dev).contiguous()
    eidx = torch.zeros(R, dtype=torch.int32, device=dev)
    times = {"head": [], "base": []}
    for it in range(6):
        for p in fld.parameters():
            p.grad = None
        h, sigma = fld._base_mlp(y, sel, n)
        rgb = fld.rgb_packed(h, dirs, eidx, ri, packed, fld._train_emb_table())
        loss = rgb.sum() + sigma.sum()
        _lib.TIMING = {"names": {"lse_mlp_bwd"}, "events": []}
        loss.backward()
        torch.cuda.synchronize()
        ev = _lib.TIMING["events"]; _lib.TIMING = None
        assert len(ev) == 2, [e[0] for e in ev]
        if it:
            times["head"].append(ev[0][1].elapsed_time(ev[0][2]))      # the head's backward runs first
            times["base"].append(ev[1][1].elapsed_time(ev[1][2]))
    res[n] = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    print(f"N = {n:8d}   head bwd {res[n]['head']:.4f} ms   base bwd {res[n]['base']:.4f} ms", flush=True)
ns = sorted(res)
for k in ("head", "base"):
    xs = torch.tensor([float(n) for n in ns], dtype=torch.float64); ys = torch.tensor([res[n][k] for n in ns], dtype=torch.float64)
    A = torch.stack([torch.ones_like(xs), xs], 1)
    a, b = torch.linalg.lstsq(A, ys[:, None]).solution.flatten().tolist()
    print(f"{k}: time = {a * 1e3:.1f} us + {b * 1024 * 4096:.4f} ms per 4096 x 1024 samples   (intercept = {100 * a / res[ns[-1]][k]:.1f} % of the full-size launch)")
