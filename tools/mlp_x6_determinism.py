import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lsenerf_amd import ops, _lib
dev = "cuda"
def run_head(N, R, flag, seed=3):
    torch.manual_seed(seed)
    meta = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR, 64, 15, 1)
    params0 = torch.randn(64 * 64 + 64 * 64 + 16 * 64) * 0.15
    x0 = torch.randn(N, 16); rb0 = torch.randn(R, 64) * 0.3
    ridx = (torch.arange(N) * R // N)
    cnt = torch.bincount(ridx, minlength=R)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
    w = torch.randn(N, 4)
    ops.RECOMPUTE_ALL = flag
    p = params0.clone().to(dev).requires_grad_(True); x = x0.clone().to(dev).requires_grad_(True); rb = rb0.clone().to(dev).requires_grad_(True)
    out = ops.fused_mlp(p, x, meta, N, rb, ridx.int().to(dev), packed.to(dev), out_cols=4)
    (out * w.to(dev)).sum().backward()
    ops.RECOMPUTE_ALL = True
    return out.detach().clone(), p.grad.clone(), x.grad.clone(), rb.grad.clone()
def run_base(N, flag, seed=5):
    torch.manual_seed(seed)
    meta = ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR)
    params0 = torch.randn(meta.n_params) * 0.15
    y0 = torch.randn(16, N, 2); sel = (torch.rand(N) > 0.2).to(torch.uint8); w = torch.randn(N, 16); ws = torch.randn(N)
    ops.RECOMPUTE_ALL = flag
    p = params0.clone().to(dev).requires_grad_(True); y = y0.clone().to(dev).requires_grad_(True)
    out, sigma = ops.fused_mlp(p, y, meta, N, density=(sel.to(dev), 1.0))
    ((out * w.to(dev)).sum() + (sigma * ws.to(dev)).sum()).backward()
    ops.RECOMPUTE_ALL = True
    return out.detach().clone(), p.grad.clone(), y.grad.clone()
e = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
for N, R in ((1 << 20, 1024), (300000, 777)):
    for flag in (True, False):
        a = run_head(N, R, flag); b = run_head(N, R, flag)
        print(f"head N={N} R={R} recompute_all={flag}: run-to-run out {e(a[0], b[0]):.1e} dparams {e(a[1], b[1]):.1e} dx {e(a[2], b[2]):.1e} dbias {e(a[3], b[3]):.1e}", flush=True)
        a = run_base(N, flag); b = run_base(N, flag)
        print(f"base N={N} recompute_all={flag}: run-to-run out {e(a[0], b[0]):.1e} dparams {e(a[1], b[1]):.1e} dy {e(a[2], b[2]):.1e}", flush=True)
