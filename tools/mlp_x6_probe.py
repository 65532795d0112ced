#!/usr/bin/env python3
"""Fused-MLP forward on the bf16 matrix cores with three-piece operands (mlp_x6.h, lse_mlp_desc.arith = AUTO) next to the f32-MFMA
forward (impl 1): time at the metric size, and error of both against a float64 evaluation of the same network on a subset."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from lsenerf_amd import ops, _lib
from bench_kernels import timeit
dev = "cuda"
R = 4096
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None


def ref64(meta, params, x_rows, rb_rows, head):
    """float64 network on rows (x_rows [M, n_in], rb_rows [M, 64] or None) -> (h_last [M,64], out [M,16])"""
    p = params.double().cpu()
    if head:   # first-layer view: W0[row][c] = p[row*64 + 15 + c], column 0 masked
        W0 = p[:64 * 64].view(64, 64)[:, 15:31].clone()
        W0[:, 0] = 0
        rest = p[64 * 64:]
    else:
        W0 = p[:64 * 32].view(64, 32)
        rest = p[64 * 32:]
    h = x_rows.double().cpu() @ W0.T
    if rb_rows is not None:
        h = h + rb_rows.double().cpu()
    h = h.clamp_min(0)
    if head:
        W1 = rest[:4096].view(64, 64)
        rest = rest[4096:]
        h = (h @ W1.T).clamp_min(0)
    Wo = rest[:16 * 64].view(16, 64)
    o = h @ Wo.T
    if head:
        o = torch.sigmoid(o)
    return h, o


for N in (4096 * 1024, 1000):
    for name, head in (("head 16->64->64->3(4)", True), ("base 32->64->16 (level-major in)", False)):
        torch.manual_seed(1)
        if head:
            meta = ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR, 64, 15, 1)
            x = torch.randn(N, 16, device=dev)
            rb = torch.randn(R, 64, device=dev)
            params = torch.randn(64 * 64 + 64 * 64 + 16 * 64, device=dev) * 0.15
            ridx = (torch.arange(N, device=dev) * R // N).to(torch.int32)
            oc = 4
        else:
            meta = ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR)
            x = torch.randn(16, N, 2, device=dev)
            rb = ridx = None
            params = torch.randn(meta.n_params, device=dev) * 0.15
            oc = 16
        import dataclasses
        nl = meta.n_hidden_layers
        npad = (N + 15) // 16 * 16
        res = {}
        for impl in (1, 2):
            # (the arithmetic route is a descriptor field since ABI 5: 1 = f32 MFMA, 2 = bf16 pieces)
            desc = dataclasses.replace(meta, arith=_lib.LSE_MLP_ARITH_F32_MFMA if impl == 1 else _lib.LSE_MLP_ARITH_AUTO).desc()
            out = torch.zeros(N, oc, device=dev)
            act = torch.zeros(nl, npad, 64, device=dev)

            def run():
                _lib.call("lse_mlp_fwd", ctypes.byref(desc), P(params), P(x), P(rb), P(ridx), P(out), oc, P(act), 1,
                          None, None, 0.0, N, None, ops._stream())
            run()
            torch.cuda.synchronize()
            t = timeit(run)[0] if N > 100000 else float("nan")
            res[impl] = (out.clone(), act.clone(), t)
        # float64 reference on a subset of rows
        M = min(N, 8192)
        rows = torch.randperm(N, device=dev)[:M]
        xr = x[rows] if head else x[:, rows, :].permute(1, 0, 2).reshape(M, 32)
        rbr = rb[ridx[rows].long()] if head else None
        h64, o64 = ref64(meta, params, xr, rbr, head)
        o64 = o64[:, :oc].to(dev)
        line = f"N={N} {name}:"
        for impl in (1, 2):
            out, act, t = res[impl]
            eo = float((out[rows].double() - o64).abs().max() / o64.abs().max())
            # last hidden layer from the tile-major activation image: [tile16][rb][lane(q*16+j)][4]
            a = act[nl - 1].view(npad // 16, 4, 4, 16, 4)          # tile, rb, q, j, r
            hl = a.permute(0, 3, 1, 2, 4).reshape(npad, 64)[:N]    # sample = tile*16+j, neuron = 16rb+4q+r
            eh = float((hl[rows].double().cpu() - h64).abs().max() / h64.abs().max())
            line += f"  impl {impl}: {t:.3f} ms, out err {eo:.2e}, hidden err {eh:.2e};"
        d = float((res[1][0] - res[2][0]).abs().max())
        print(line + f"  max|impl1 - impl2| = {d:.2e}", flush=True)
