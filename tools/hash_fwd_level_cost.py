"""Relative cost of each hash-grid level in the forward kernel: time lse_hash_fwd on a 1-level descriptor per level."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lsenerf_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_kernels import timeit

R, S = 4096, 1024
N = R * S
dev = "cuda"
g = torch.Generator().manual_seed(1)
o = (torch.rand(R, 3, generator=g) - 0.5).to(dev)
d = torch.randn(R, 3, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
step = 2 * 3 ** 0.5 / 1000
ts = (0.05 + step * torch.arange(S, dtype=torch.float32)).repeat(R).to(dev)
te = ts + step
ri = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(dev)
cnt = torch.full((R,), S, dtype=torch.long)
packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).to(dev).contiguous()
x01, sel = ops.positions(o, d, ri, ts, te, packed, True, None)
meta = ops.make_grid_meta()
table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 1e-2).to(dev)
full = meta.desc()
y1 = torch.empty((1, N, 2), device=dev)
tot = 0.0
for l in range(meta.n_levels):
    d1 = _lib.GridDesc()
    d1.n_levels, d1.n_features = 1, 2
    d1.offsets[0], d1.offsets[1] = 0, full.offsets[l + 1] - full.offsets[l]
    d1.scales[0], d1.resolutions[0] = full.scales[l], full.resolutions[l]
    tab_l = table[2 * full.offsets[l]:2 * full.offsets[l + 1]]
    f = lambda: _lib.call("lse_hash_fwd", ctypes.byref(d1), ctypes.c_void_p(x01.data_ptr()), ctypes.c_void_p(tab_l.data_ptr()),
                          ctypes.c_void_p(y1.data_ptr()), N, None, ops._stream())
    med, mn = timeit(f)
    tot += med
    print(f"level {l:2d} res {full.resolutions[l]:5d}: {med:.4f} ms (all 8 XCDs on this level)", flush=True)
print(f"sum {tot:.3f} ms")
