#!/usr/bin/env python3
"""Does the hash forward of one half of the samples hide behind the fused MLP forwards of the other half?  Metric-size workload
(4096 x 1024 samples), two streams: hash(A) -> [hash(B) || base(A), head(A)] -> base(B), head(B), against the one-stream sequence."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lsenerf_amd import ops, _lib as L
dev = "cuda"
R, S = 4096, 1024
N = R * S
g = torch.Generator().manual_seed(1)
o = (torch.rand(R, 3, generator=g) - 0.5).to(dev)
d = torch.randn(R, 3, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
step = 2 * 3 ** 0.5 / 1000
ts = (0.05 + step * torch.arange(S, dtype=torch.float32)).repeat(R).to(dev)
ri = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(dev)
cnt = torch.full((R,), S, dtype=torch.long)
packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).to(dev).contiguous()
x01, sel = ops.positions(o, d, ri, ts, ts + step, packed, True, None)
meta = ops.make_grid_meta()
table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 1e-2).to(dev)
mb = ops.MlpMeta(32, 64, 1, L.LSE_ACT_NONE, L.LSE_IN_LEVELMAJOR)
mh = ops.MlpMeta(16, 64, 2, L.LSE_ACT_SIGMOID, L.LSE_IN_ROWMAJOR)
pb = torch.randn(mb.n_params, device=dev) * 0.1
ph = torch.randn(mh.n_params, device=dev) * 0.1

def chain(x, n):
    y = ops.hash_encode(x, table, meta)
    h = ops.fused_mlp(pb, y, mb, n, out_cols=16)
    return ops.fused_mlp(ph, h, mh, n, out_cols=4)

def timed(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts_ = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts_.append(e0.elapsed_time(e1))
    ts_.sort()
    return ts_[len(ts_) // 2]

with torch.no_grad():
    print(f"one stream, whole batch: {timed(lambda: chain(x01, N)):.3f} ms", flush=True)
    for parts in (2, 4, 8):
        xs = [c.contiguous() for c in x01.chunk(parts)]
        def seq():
            for x in xs: chain(x, x.shape[0])
        print(f"one stream, {parts} chunks one after the other: {timed(seq):.3f} ms", flush=True)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        def piped():
            main = torch.cuda.current_stream()
            s1.wait_stream(main); s2.wait_stream(main)
            evs = []
            ys = []
            with torch.cuda.stream(s1):
                for x in xs:
                    ys.append(ops.hash_encode(x, table, meta))
                    e = torch.cuda.Event(); e.record(s1); evs.append(e)
            with torch.cuda.stream(s2):
                for x, y, e in zip(xs, ys, evs):
                    s2.wait_event(e)
                    h = ops.fused_mlp(pb, y, mb, x.shape[0], out_cols=16)
                    ops.fused_mlp(ph, h, mh, x.shape[0], out_cols=4)
            main.wait_stream(s1); main.wait_stream(s2)
        print(f"two streams, {parts} chunks, hash(k+1) beside mlp(k): {timed(piped):.3f} ms", flush=True)
