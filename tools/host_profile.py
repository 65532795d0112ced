#!/usr/bin/env python3
"""Where does the HOST spend a bench step?  cProfile over K steps of bench.train_step (run from a repo / worktree root)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from lsenerf_amd import _lib
from lsenerf_amd.optim import FlatAdam, FlatParams
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
_lib.load()
model, _sets, _ = bench.build_workload(dev, seed=1000)      # (one ray set: the round-1..4 fixed draw)
rb, target, jitter = _sets[0]
flat = FlatParams(model.get_param_groups()["fields"])
opt = FlatAdam(flat, lr=1e-2, eps=1e-15)
for _ in range(5):
    bench.train_step(model, rb, target, jitter, opt, 1)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    bench.train_step(model, rb, target, jitter, opt, 1)
pr.disable()
torch.cuda.synchronize()
print("ms/step under cProfile:", (time.perf_counter() - t0) / K * 1e3)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 22)
print(s.getvalue()[:6000])
