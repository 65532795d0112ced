#!/usr/bin/env python3
"""One hash-backward configuration on one regime, a few launches (for rocprofv3 --pmc passes).
usage: python tools/hash_bwd_one.py <M-march|M-packed|bench|default> [opt=val ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from lsenerf_amd import ops, _lib
import bench
dev = torch.device("cuda", 0)
R, S = 4096, 1024
regime = sys.argv[1]
meta = ops.make_grid_meta()
g = torch.Generator().manual_seed(1)
table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 1e-2).to(dev)
desc = meta.desc()
P = lambda t: ctypes.c_void_p(t.data_ptr())
def fixed(o, d):
    step = 2 * 3 ** 0.5 / 1000
    ts = (0.05 + step * torch.arange(S, dtype=torch.float32)).repeat(R).to(dev)
    ri = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(dev)
    cnt = torch.full((R,), S, dtype=torch.long)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).to(dev).contiguous()
    return ops.positions(o.to(dev), d.to(dev), ri, ts, ts + step, packed, True, None)[0]
if regime == "M-march":
    o = torch.rand(R, 3, generator=g) - 0.5
    d = torch.randn(R, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
    x01 = fixed(o, d)
elif regime == "M-packed":
    o, d = bench.sphere_rays(R, torch.Generator().manual_seed(96))
    x01 = fixed(o, d)
elif regime == "bench":                 # the positions of bench.py's headline step (SURVEY 8d rays, chord inside the box)
    model, _sets, _ = bench.build_workload(dev, 96)      # (one ray set: the round-1..4 fixed draw)
    rb, _, jitter = _sets[0]
    cfg = model.config
    ri, ts, te, packed = model.occupancy_grid.sampling(
        rb.origins.detach(), rb.directions.detach(), near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        render_step_size=cfg.render_step_size, stratified=True, jitter=jitter, return_packed=True)[:4]
    x01 = ops.positions(rb.origins.detach(), rb.directions.detach(), ri, ts, te, packed, True, None)[0]
else:
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    torch.manual_seed(96)
    model = LSENeRFModel(LSENeRFModelConfig(), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev).train()
    with torch.no_grad():
        model.field.mlp_base_grid.params.mul_(3000.0)
        model.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
    o, d = bench.sphere_rays(R, torch.Generator().manual_seed(7))
    rb = RayBundle(origins=o.to(dev), directions=d.to(dev), camera_indices=torch.zeros(R, 1, dtype=torch.long, device=dev))
    for s_ in range(0, 64, 16):
        model.update_occupancy_grid(s_)
    rs, _ = model.sampler(ray_bundle=rb, near_plane=0.05, far_plane=1e3, render_step_size=model.config.render_step_size,
                          alpha_thre=0.01, cone_angle=0.004)
    x01 = ops.positions(rb.origins, rb.directions, rs.ray_indices, rs.frustums.starts[..., 0].contiguous(),
                        rs.frustums.ends[..., 0].contiguous(), rs.packed_info, True, None)[0]
n = x01.shape[0]
dy = torch.randn(16, n, 2, device=dev)
dt = torch.zeros_like(table); dx = torch.empty_like(x01)
o_ = _lib.hash_bwd_default_opts()
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    setattr(o_, k, int(v))
nb = int(_lib.load().lse_hash_bwd_workspace_bytes(ctypes.byref(desc), ctypes.byref(o_)))
if nb:
    ws = torch.zeros(nb // 4, dtype=torch.float32, device=dev)
    o_.workspace, o_.workspace_bytes = ws.data_ptr(), nb
for _ in range(3):
    _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(x01), P(dy), P(table), P(dt), P(dx), 0, 0, 16, n, None, ctypes.byref(o_), ops._stream())
torch.cuda.synchronize()
# distinct 64-byte lines of the gradient table per 64-sample window (= one wave of the kernel), per level: the request floor
lines = []
for l in range(16):
    sc = meta.scales[l]; res = meta.resolutions[l]; size = meta.offsets[l + 1] - meta.offsets[l]
    p = x01 * sc + 0.5
    p0 = p.floor().to(torch.int64)
    tot = 0
    acc = []
    for c in range(8):
        q = p0 + torch.tensor([c & 1, (c >> 1) & 1, (c >> 2) & 1], device=dev)
        if res ** 3 <= size:
            idx = q[:, 0] + q[:, 1] * res + q[:, 2] * res * res
        else:
            idx = (q[:, 0] ^ (q[:, 1] * 2654435761) ^ (q[:, 2] * 805459861)) & 0xFFFFFFFF
        acc.append((idx % size) >> 3)
    ln = torch.stack(acc, 1)                                     # [n, 8]
    pad = (-n) % 64
    if pad:
        ln = torch.cat([ln, ln[-1:].expand(pad, 8)])
    w = ln.reshape(-1, 64 * 8).sort(dim=1).values
    lines.append(float(((w[:, 1:] != w[:, :-1]).sum() + w.shape[0]) / n))
print(regime, n, "samples; distinct 64-B lines per sample in a 64-sample window, per level:", " ".join(f"{v:.2f}" for v in lines),
      "sum", f"{sum(lines):.2f}")
