"""Lists the aten::zero_/fill_/zeros calls of one bench train step with their Python call sites (profiling aid)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from lsenerf_amd.optim import FlatAdam, FlatParams
dev = torch.device("cuda", 0)
model, _sets, _ = bench.build_workload(dev, seed=1000)      # (one ray set: the round-1..4 fixed draw)
rb, target, jitter = _sets[0]
flat = FlatParams(model.get_param_groups()["fields"])
opt = FlatAdam(flat, lr=1e-2, eps=1e-15)
for _ in range(2):
    bench.train_step(model, rb, target, jitter, opt, 1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    bench.train_step(model, rb, target, jitter, opt, 1)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::add_", "aten::add", "aten::copy_"):
        st = [f for f in (ev.stack or []) if "lsenerf_amd" in f or "bench.py" in f]
        cnt[(ev.name, str(ev.input_shapes)[:60], st[0][-70:] if st else "(autograd engine / torch internals)")] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(v, k)
