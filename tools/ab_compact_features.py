#!/usr/bin/env python3
"""lse_compact_features (the survivors of the visibility pre-pass keep their hash features) with the levels split over 1 .. 16 groups
of waves per ray (option compact_features_groups), at the size of the reference's default configuration: 3510 rays x ~355 candidate
samples, ~70 % kept.  Outputs must be identical; interleaved rounds in one process."""
import os, sys
os.environ.setdefault("LSE_DEV", "1")      # tuning knobs exist in the development build only (csrc/dev_knobs.h)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from lsenerf_amd import ops, _lib
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
R, L = 3510, 16
cnt = torch.randint(200, 500, (R,), device=dev, generator=g)
packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).contiguous()
N = int(cnt.sum())
# runs of kept / culled samples like a visibility cull (whole stretches of a ray survive)
keep = (torch.rand(N, device=dev, generator=g) < 0.7)
keep = (torch.nn.functional.avg_pool1d(keep.float()[None, None], 9, 1, 4)[0, 0] > 0.5)
mask = keep.to(torch.uint8).contiguous()
seg = torch.repeat_interleave(torch.arange(R, device=dev), cnt)
new_cnt = torch.zeros(R, dtype=torch.long, device=dev).index_add_(0, seg, keep.long())
new_packed = torch.stack([torch.cumsum(new_cnt, 0) - new_cnt, new_cnt], -1).contiguous()
n_new = int(new_cnt.sum())
x01 = torch.rand(N, 3, device=dev, generator=g)
sel = (torch.rand(N, device=dev, generator=g) < 0.9).to(torch.uint8)
y = torch.randn(L, N, 2, device=dev, generator=g)
print(f"{R} rays, {N} candidates, {n_new} survivors; bytes moved {(n_new * (12 + 1 + L * 8)) * 2 / 1e6:.0f} MB")


def timed(fn, iters=20):
    ts = []
    for i in range(iters + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


ref = None
res = {}
for rnd in range(3):
    for k in (1, 2, 4, 8, 16):
        _lib.set_option("compact_features_groups", k)
        out = ops.compact_features(mask, packed, new_packed, n_new, x01, sel, y)
        if ref is None:
            ref = [t.clone() for t in out]
            # the answer by boolean indexing
            assert torch.equal(ref[0], x01[keep]) and torch.equal(ref[1], sel[keep]) and torch.equal(ref[2], y[:, keep])
        assert all(torch.equal(a, b) for a, b in zip(out, ref)), k
        res.setdefault(k, []).append(timed(lambda: ops.compact_features(mask, packed, new_packed, n_new, x01, sel, y)))
for k, v in res.items():
    print(f"   groups {k:2d}: " + " ".join(f"{t:6.1f}" for t in v) + " us")
_lib.set_option("compact_features_groups", 4)
