import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lsenerf_amd import ops, _lib
dev = "cuda"
def run(N, R, view, oc, seed=3):
    torch.manual_seed(seed)
    if view:
        meta = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR, 64, 15, 1)
        params0 = torch.randn(64 * 64 + 64 * 64 + 16 * 64) * 0.15
    else:
        meta = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR)
        params0 = torch.randn(16 * 64 + 64 * 64 + 16 * 64) * 0.15
    x0 = torch.randn(N, 16)
    rb0 = torch.randn(R, 64) * 0.3
    ridx = (torch.arange(N) * R // N)
    cnt = torch.bincount(ridx, minlength=R)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    w = torch.randn(N, oc)
    res = {}
    for name, flag in (("gen3", True), ("gen2", False)):
        ops.RECOMPUTE_ALL = flag
        p = params0.clone().to(dev).requires_grad_(True)
        x = x0.clone().to(dev).requires_grad_(True)
        rb = rb0.clone().to(dev).requires_grad_(True)
        out = ops.fused_mlp(p, x, meta, N, rb, ridx.int().to(dev), packed.to(dev), out_cols=oc)
        (out * w.to(dev)).sum().backward()
        res[name] = (p.grad.cpu(), x.grad.cpu(), rb.grad.cpu())
    ops.RECOMPUTE_ALL = True
    e = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    print(f"N={N} R={R} view={view} oc={oc}: dparams {e(res['gen3'][0], res['gen2'][0]):.2e} dx {e(res['gen3'][1], res['gen2'][1]):.2e} dbias {e(res['gen3'][2], res['gen2'][2]):.2e}", flush=True)
for args in ((64, 1, True, 4), (64, 1, False, 16), (64, 1, True, 16), (64, 1, False, 4), (4096, 4, True, 4), (4096, 400, True, 4), (40000, 157, True, 4)):
    run(*args)
