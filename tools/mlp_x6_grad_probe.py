#!/usr/bin/env python3
"""Head MLP with the first-layer view (lsenerf_amd/field.py: rgb_packed): gradients of the third-generation backward against the
second-generation one and a float64 reference, per parameter region."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lsenerf_amd import ops, _lib
dev = "cuda"
torch.manual_seed(3)
N, R = 40000, 157
meta = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR, 64, 15, 1)
params0 = torch.randn(64 * 64 + 64 * 64 + 16 * 64) * 0.15
x0 = torch.randn(N, 16)
rb0 = torch.randn(R, 64) * 0.3
cnt = torch.randint(0, 2 * N // R, (R,))
cnt[-1] = 0
ridx = torch.repeat_interleave(torch.arange(R), cnt)
if len(ridx) < N:
    ridx = torch.cat([ridx, torch.full((N - len(ridx),), R - 1)])
ridx = ridx[:N]
cnt = torch.bincount(ridx, minlength=R)
packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1)
w = torch.randn(N, 4)
res = {}
for name, flag in (("gen3", True), ("gen2", False)):
    ops.RECOMPUTE_ALL = flag
    p = params0.clone().to(dev).requires_grad_(True)
    x = x0.clone().to(dev).requires_grad_(True)
    rb = rb0.clone().to(dev).requires_grad_(True)
    out = ops.fused_mlp(p, x, meta, N, rb, ridx.int().to(dev), packed.to(dev), out_cols=4)
    (out * w.to(dev)).sum().backward()
    res[name] = (out.detach().cpu(), p.grad.cpu(), x.grad.cpu(), rb.grad.cpu())
ops.RECOMPUTE_ALL = True
# float64 reference
p = params0.double().requires_grad_(True)
x = x0.double().requires_grad_(True)
rb = rb0.double().requires_grad_(True)
W0 = p[:4096].view(64, 64)[:, 15:31]
mask = torch.ones(16, dtype=torch.float64); mask[0] = 0
h = torch.relu(x @ (W0 * mask).T + rb[ridx])
h = torch.relu(h @ p[4096:8192].view(64, 64).T)
o = torch.sigmoid(h @ p[8192:].view(16, 64).T)[:, :4]
(o * w.double()).sum().backward()
ref = (o.detach(), p.grad, x.grad, rb.grad)
def err(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())
for name in ("gen3", "gen2"):
    o_, pg, xg, bg = res[name]
    g0 = pg[:4096].view(64, 64)
    r0 = ref[1][:4096].view(64, 64)
    print(name, "out", err(o_, ref[0]), "| dW0 view", err(g0[:, 16:31], r0[:, 16:31]), "col15(masked)", float(g0[:, 15].abs().max()),
          "outside view", float(g0[:, :15].abs().max()), float(g0[:, 31:].abs().max()),
          "| dW1", err(pg[4096:8192], ref[1][4096:8192]), "| dWo", err(pg[8192:], ref[1][8192:]),
          "| dx", err(xg, ref[2]), "| dbias", err(bg, ref[3]), flush=True)

# ---- base MLP with the fused density head (sigma = scale * exp(out[:,0]) * selector), level-major input
torch.manual_seed(5)
for N in (40000, 77, 1):
    meta = ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR)
    params0 = torch.randn(meta.n_params) * 0.15
    y0 = torch.randn(16, N, 2)
    sel = (torch.rand(N) > 0.2).to(torch.uint8)
    w = torch.randn(N, 16)
    ws = torch.randn(N)
    res = {}
    for name, flag in (("gen3", True), ("gen2", False)):
        ops.RECOMPUTE_ALL = flag
        p = params0.clone().to(dev).requires_grad_(True)
        y = y0.clone().to(dev).requires_grad_(True)
        out, sigma = ops.fused_mlp(p, y, meta, N, density=(sel.to(dev), 1.0))
        ((out * w.to(dev)).sum() + (sigma * ws.to(dev)).sum()).backward()
        res[name] = (out.detach().cpu(), sigma.detach().cpu(), p.grad.cpu(), y.grad.cpu())
    ops.RECOMPUTE_ALL = True
    def e(a, b):
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    print(f"base N={N}: gen3 vs gen2: out {e(res['gen3'][0], res['gen2'][0]):.2e} sigma {e(res['gen3'][1], res['gen2'][1]):.2e} "
          f"dparams {e(res['gen3'][2], res['gen2'][2]):.2e} dy {e(res['gen3'][3], res['gen2'][3]):.2e}", flush=True)
