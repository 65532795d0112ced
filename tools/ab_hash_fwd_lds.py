#!/usr/bin/env python3
"""A/B of the LDS-resident variant of the coarsest hash levels (option hash_fwd_lds_levels; BASELINE.json north_star names
"LDS-staged trilinear interpolation") against the L2-resident level-major schedule, on the headline positions and on M-packed
positions: whole lse_hash_fwd, interleaved rounds in one process, outputs compared bit for bit.
usage: python tools/ab_hash_fwd_lds.py [out.txt]"""
import os, sys
os.environ.setdefault("LSE_DEV", "1")      # tuning knobs exist in the development build only (csrc/dev_knobs.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from lsenerf_amd import ops, _lib

dev = torch.device("cuda", 0)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def timed(fn, iters=9):
    ts = []
    for i in range(iters + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


model, _sets, _ = bench.build_workload(dev, 1000)      # (one ray set: the round-1..4 fixed draw)


rb, _, jitter = _sets[0]
cfg = model.config
with torch.no_grad():
    ri, ts_, te_, packed = model.occupancy_grid.sampling(
        rb.origins.detach(), rb.directions.detach(), near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        render_step_size=cfg.render_step_size, stratified=True, jitter=jitter, return_packed=True)[:4]
    x_march = ops.positions(rb.origins.detach(), rb.directions.detach(), ri, ts_, te_, packed, True, None)[0]
    R, S = 4096, 1024
    o, d = bench.sphere_rays(R, torch.Generator().manual_seed(96))
    step = 2 * 3 ** 0.5 / 1000
    tsp = (0.05 + step * torch.arange(S, dtype=torch.float32)).repeat(R).to(dev)
    rip = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(dev)
    cnt = torch.full((R,), S, dtype=torch.long)
    pk = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).to(dev).contiguous()
    x_packed = ops.positions(o.to(dev), d.to(dev), rip, tsp, tsp + step, pk, True, None)[0]
meta = model.field.mlp_base_grid.meta
table = model.field.mlp_base_grid.params.detach()
for name, x in (("headline (M-march, sphere rays)", x_march), ("M-packed", x_packed), ("odd count 1000003", x_packed[:1000003].contiguous())):
    say(f"== {name}: {x.shape[0]} samples")
    ref = None
    res = {}
    for rnd in range(3):
        for k in (0, 1, 2):
            _lib.set_option("hash_fwd_lds_levels", k)
            with torch.no_grad():
                y = ops.hash_encode(x, table, meta)
                if ref is None:
                    ref = y.clone()
                assert torch.equal(y, ref), f"lds_levels={k}: outputs differ"
                res.setdefault(k, []).append(timed(lambda: ops.hash_encode(x, table, meta)))
    for k in (0, 1, 2):
        say(f"   hash_fwd_lds_levels={k}: " + " ".join(f"{t:.4f}" for t in res[k]) + " ms   (bit-identical to the L2-resident schedule)")
_lib.set_option("hash_fwd_lds_levels", 0)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
