import sys, ctypes; sys.path.insert(0, '.')
import torch
from lsenerf_amd import ops, _lib, LSEOccGridEstimator
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda"
g = torch.Generator().manual_seed(1)
o = (torch.rand(R, 3, generator=g) - 0.5).to(dev)
d = torch.randn(R, 3, generator=g); d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
step = 2 * 3 ** 0.5 / 1000
for levels, S, occ in ((4, 1024, 1.0), (4, 128, 1.0), (1, 256, 1.0), (4, 1024, 0.0)):
    est = LSEOccGridEstimator([-1, -1, -1, 1, 1, 1], 128, levels).to(dev)
    if occ > 0: est.mark_all_occupied()
    near = torch.full((R,), 0.05, device=dev); far = torch.full((R,), 0.05 + S * step - 0.25 * step, device=dev)
    cnts = torch.empty(R, dtype=torch.int64, device=dev)
    args = (ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(d.data_ptr()), R, ctypes.c_void_p(est._binaries_u8().data_ptr()),
            ctypes.c_void_p(est.aabbs.data_ptr()), levels, 128, 128, 128, ctypes.c_void_p(near.data_ptr()),
            ctypes.c_void_p(far.data_ptr()), float(step), 0.0)
    def count():
        _lib.call("lse_traverse_grids", *args, 0, ctypes.c_void_p(cnts.data_ptr()), None, None, None, None, 0, ops._stream())
    for _ in range(3): count()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): count()
    e1.record(); torch.cuda.synchronize()
    print(f"levels={levels} S={S} occ={occ}: count pass {e0.elapsed_time(e1)/5*1e3:.1f} us, total samples {int(cnts.sum())}", flush=True)
