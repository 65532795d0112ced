#!/usr/bin/env python3
"""Marginal cost of every level INSIDE one fused hash-backward launch (M-march positions): time levels [0, k) and [k, 16)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from lsenerf_amd import ops, _lib
from bench_kernels import timeit
dev = torch.device("cuda", 0)
R, S = 4096, 1024
meta = ops.make_grid_meta()
g = torch.Generator().manual_seed(1)
table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 1e-2).to(dev)
desc = meta.desc()
P = lambda t: ctypes.c_void_p(t.data_ptr())
o = (torch.rand(R, 3, generator=g) - 0.5).to(dev)
d = torch.randn(R, 3, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
step = 2 * 3 ** 0.5 / 1000
ts = (0.05 + step * torch.arange(S, dtype=torch.float32)).repeat(R).to(dev)
ri = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(dev)
cnt = torch.full((R,), S, dtype=torch.long)
packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).to(dev).contiguous()
x01 = ops.positions(o, d, ri, ts, ts + step, packed, True, None)[0]
n = x01.shape[0]
dy = torch.randn(16, n, 2, device=dev)
dt = torch.zeros_like(table); dx = torch.empty_like(x01)
opts = _lib.hash_bwd_default_opts()
for k, v in [a.split("=") for a in sys.argv[1:]]:
    setattr(opts, k, int(v))
def run(lo, hi):
    f = lambda: _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(x01), P(dy), P(table), P(dt), P(dx), 0, lo, hi, n,
                          None, ctypes.byref(opts), ops._stream())
    return timeit(f, iters=7, warm=2)[0]
prev = 0.0
print("k   t[0,k)   marginal   t[k,16)")
for k in range(1, 17):
    a = run(0, k); b = run(k, 16) if k < 16 else 0.0
    print(f"{k:2d}  {a:7.4f}  {a - prev:8.4f}   {b:7.4f}", flush=True)
    prev = a
