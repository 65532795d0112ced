"""Micro-benchmarks of individual entry points on the metric-size workload (N = 4096 x 1024), interleaved rounds in one
process (cdna_hip_programming.md rule 24).  Usage: python tools/bench_kernels.py [names...]"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from lsenerf_amd import ops, _lib

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts)//2], ts[0]

def main():
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
    which = set(sys.argv[1:])
    def want(n): return not which or n in which
    if want("hash_fwd"):
        med, mn = timeit(lambda: ops.hash_encode(x01, table, meta))
        print(f"hash_fwd  median {med:.3f} ms  min {mn:.3f} ms  -> {1024*N/med/1e6:.0f} GB/s algorithmic", flush=True)
    y = ops.hash_encode(x01, table, meta)
    if want("hash_bwd"):
        dy = torch.randn_like(y)
        dtab = torch.zeros_like(table); dx = torch.empty_like(x01)
        import ctypes
        desc = meta.desc()
        def bwd(with_dx=True):
            _lib.call("lse_hash_bwd", ctypes.byref(desc), ctypes.c_void_p(x01.data_ptr()), ctypes.c_void_p(dy.data_ptr()),
                      ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(dtab.data_ptr()),
                      ctypes.c_void_p(dx.data_ptr()) if with_dx else None, N, None, ops._stream())
        med, mn = timeit(bwd)
        print(f"hash_bwd(dx) median {med:.3f} ms  min {mn:.3f} ms  -> {1024*N/med/1e6:.0f} GB/s algorithmic", flush=True)
        med, mn = timeit(lambda: bwd(False))
        print(f"hash_bwd(no dx) median {med:.3f} ms  min {mn:.3f}", flush=True)
        for lo, hi in ((0, 4), (4, 8), (8, 12), (12, 16)):
            def rng():
                _lib.call("lse_hash_bwd_levels", ctypes.byref(desc), ctypes.c_void_p(x01.data_ptr()), ctypes.c_void_p(dy.data_ptr()),
                          ctypes.c_void_p(table.data_ptr()), ctypes.c_void_p(dtab.data_ptr()), ctypes.c_void_p(dx.data_ptr()),
                          0, lo, hi, N, None, ops._stream())
            med, mn = timeit(rng)
            print(f"hash_bwd levels [{lo},{hi}) median {med:.3f} ms", flush=True)
    if want("mlp"):
        from lsenerf_amd import _lib as L
        for name, meta, xin in (("head", ops.MlpMeta(16, 64, 2, L.LSE_ACT_SIGMOID, L.LSE_IN_ROWMAJOR), torch.randn(N, 16, device=dev)),
                                ("base", ops.MlpMeta(32, 64, 1, L.LSE_ACT_NONE, L.LSE_IN_LEVELMAJOR), y)):
            p = torch.randn(meta.n_params, device=dev) * 0.1
            with torch.no_grad():
                med, mn = timeit(lambda: ops.fused_mlp(p, xin, meta, N, out_cols=4 if name == "head" else 16))
            print(f"mlp_fwd {name} (no act save) median {med:.3f} ms", flush=True)
            pg = p.clone().requires_grad_(True)
            med, mn = timeit(lambda: ops.fused_mlp(pg, xin, meta, N, out_cols=4 if name == "head" else 16))
            print(f"mlp_fwd {name} (act saved)   median {med:.3f} ms", flush=True)
            xg = xin.clone().requires_grad_(True)
            out = ops.fused_mlp(pg, xg, meta, N, out_cols=4 if name == "head" else 16)
            go = torch.randn_like(out)
            def bwd():
                pg.grad = None; xg.grad = None
                out.backward(go, retain_graph=True)
            med, mn = timeit(bwd)
            print(f"mlp_bwd {name} (autograd backward incl. zero-fills) median {med:.3f} ms  min {mn:.3f}", flush=True)
    if want("traverse"):
        from lsenerf_amd import LSEOccGridEstimator
        est = LSEOccGridEstimator([-1, -1, -1, 1, 1, 1], 128, 4).to(dev); est.mark_all_occupied()
        fars = torch.full((R,), 0.05 + S * step - 0.25 * step, device=dev)
        f = lambda: est.sampling(o, d, near_plane=0.05, far_plane=1e3, t_max=fars, render_step_size=step, return_packed=True)
        med, mn = timeit(f, iters=5)
        print(f"traverse (count+write+sync) median {med:.3f} ms  min {mn:.3f}", flush=True)

if __name__ == "__main__":
    main()
