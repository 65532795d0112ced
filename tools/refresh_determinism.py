#!/usr/bin/env python3
"""Is the occupancy refresh (positions -> hash forward -> base MLP -> EMA) bit-reproducible?  Rebuilds the model N times in one
process, evaluates the step-0 refresh stage by stage and compares checksums with the first iteration.  Run it alone, and
(for the data-parallel rehearsal's situation) twice concurrently on one GPU:
    python tools/refresh_determinism.py A 300 & python tools/refresh_determinism.py B 300
Observed on MI355X (round 2): alone 0 differing iterations of 300; two processes sharing the GPU: ~0.3 % of the evaluations
return ONE 128-byte line of hash features (16 samples of one level) that a re-evaluation of the same inputs does not reproduce
-- the reason lsenerf_amd.dist.attach_grid_sync broadcasts rank 0's grid after every refresh."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import dp_rehearsal as dr
from lsenerf_amd import _lib, ops
dev = torch.device("cuda", 0)
_lib.load()
tag, n_it = sys.argv[1], int(sys.argv[2])


def cks(t):
    v = t.detach().contiguous().view(-1)
    v = (v.to(torch.uint8) if v.dtype == torch.bool else v)
    v = (v.view(torch.int32) if v.element_size() == 4 else v).to(torch.int64)
    w = (torch.arange(v.numel(), device=v.device) % 1000003) + 1
    return (int(v.sum()), int((v * w).sum()))


ref, bad = None, 0
for it in range(n_it):
    model = dr.make_model(dev)
    est, fld = model.occupancy_grid, model.field
    stages = {}
    for lvl, (indices, x) in enumerate(est._update_samples(0, 256, est._update_generator(0))):
        with torch.no_grad():
            x01, sel = fld._x01(x.reshape(-1, 3).contiguous(), None, None, None, None, None)
            y = fld.mlp_base_grid.forward_levelmajor(x01)
            h, sigma = fld._base_mlp(y, sel, x01.shape[0])
        for nm, t in (("x01", x01), ("y", y), ("h", h), ("sigma", sigma)):
            stages[f"{nm}{lvl}"] = cks(t)
        if ref is not None and stages[f"y{lvl}"] != ref[f"y{lvl}"] and stages[f"x01{lvl}"] == ref[f"x01{lvl}"]:
            y2 = fld.mlp_base_grid.forward_levelmajor(x01)
            ne = (y2 != y).any(dim=2)
            print(tag, "iter", it, "eval", lvl, "hash features differ from iteration 0; re-evaluation reproduces them:", bool(torch.equal(y2, y)),
                  "| samples that differ from the re-evaluation, per table level:", ne.sum(dim=1).tolist(), flush=True)
    if ref is None:
        ref = stages
    elif stages != ref:
        bad += 1
    if it % 25 == 0:
        print(tag, "progress", it, "differing iterations so far", bad, flush=True)
print(tag, "done:", bad, "of", n_it - 1, "iterations differ from the first", flush=True)
