#!/usr/bin/env python3
"""How much of the fused-MLP forward is the saved-activation store stream?  Times lse_mlp_fwd at the metric size with and
without the activation workspace (the no_grad path), head and base shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from lsenerf_amd import ops, _lib
from bench_kernels import timeit
dev = "cuda"
N = 4096 * 1024
R = 4096
g = torch.Generator().manual_seed(0)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for name, meta, x, rb in (
        ("head 16->64->64->3(4)", ops.MlpMeta(16, 64, 2, _lib.LSE_ACT_SIGMOID, _lib.LSE_IN_ROWMAJOR, 64, 15, 1),
         torch.randn(N, 16, device=dev), torch.randn(R, 64, device=dev)),
        ("base 32->64->16 (level-major in)", ops.MlpMeta(32, 64, 1, _lib.LSE_ACT_NONE, _lib.LSE_IN_LEVELMAJOR),
         torch.randn(16, N, 2, device=dev), None)):
    params = torch.randn(meta.n_params, device=dev) * 0.1
    oc = 4 if rb is not None else 16
    out = torch.empty(N, oc, device=dev)
    act = torch.empty(meta.n_hidden_layers, N, 64, device=dev)
    ridx = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), 1024).to(dev) if rb is not None else None
    desc = meta.desc()
    def run(with_act):
        _lib.call("lse_mlp_fwd", ctypes.byref(desc), P(params), P(x), P(rb), P(ridx), P(out), oc, P(act) if with_act else None, 1,
                  None, None, 0.0, N, None, ops._stream())
    a, _ = timeit(lambda: run(True))
    b, _ = timeit(lambda: run(False))
    flops = 2 * N * (meta.n_in * 64 + (meta.n_hidden_layers - 1) * 64 * 64 + 64 * 16)
    print(f"{name}: with activations {a:.3f} ms, without {b:.3f} ms;  MFMA time at 157.3 TF: {flops / 157.3e12 * 1e3:.3f} ms; "
          f"activation bytes {meta.n_hidden_layers * N * 256 / 1e9:.2f} GB -> {meta.n_hidden_layers * N * 256 / 1e9 / a:.2f} TB/s of stores", flush=True)
