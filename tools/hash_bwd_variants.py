#!/usr/bin/env python3
"""Whole-kernel time of hash-backward variants (lse_hash_bwd_ex options) on M-march / M-packed positions, interleaved rounds.
usage: python tools/hash_bwd_variants.py "impl=1" "impl=2 second_probe=1" ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from lsenerf_amd import ops, _lib
import bench
dev = torch.device("cuda", 0)
R, S = 4096, 1024
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
o_in = torch.rand(R, 3, generator=g) - 0.5
d_in = torch.randn(R, 3, generator=g); d_in = d_in / d_in.norm(dim=-1, keepdim=True)
o_sp, d_sp = bench.sphere_rays(R, torch.Generator().manual_seed(96))
def default_config_positions():
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    torch.manual_seed(96)
    model = LSENeRFModel(LSENeRFModelConfig(), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev).train()
    with torch.no_grad():
        model.field.mlp_base_grid.params.mul_(3000.0)
        model.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
    o, d = bench.sphere_rays(R, torch.Generator().manual_seed(7))
    rb = RayBundle(origins=o.to(dev), directions=d.to(dev), camera_indices=torch.zeros(R, 1, dtype=torch.long, device=dev))
    cb = model.update_occupancy_grid
    for s_ in range(0, 64, 16):
        cb(s_)
    rs, _ = model.sampler(ray_bundle=rb, near_plane=0.05, far_plane=1e3, render_step_size=model.config.render_step_size,
                          alpha_thre=0.01, cone_angle=0.004)
    return ops.positions(rb.origins, rb.directions, rs.ray_indices, rs.frustums.starts[..., 0].contiguous(),
                         rs.frustums.ends[..., 0].contiguous(), rs.packed_info, True, None)[0]
def headline_positions():
    """bench.py's headline step (SURVEY 8d exact: sphere rays, one-level grid fully occupied, step chosen for 1024 samples per ray)."""
    model, _sets, _ = bench.build_workload(dev, 1000)      # (one ray set: the round-1..4 fixed draw)
    rb, _, jitter = _sets[0]
    cfg = model.config
    with torch.no_grad():
        ri, ts, te, packed = model.occupancy_grid.sampling(
            rb.origins.detach(), rb.directions.detach(), near_plane=cfg.near_plane, far_plane=cfg.far_plane,
            render_step_size=cfg.render_step_size, stratified=True, jitter=jitter, return_packed=True)[:4]
        return ops.positions(rb.origins.detach(), rb.directions.detach(), ri, ts, te, packed, True, None)[0]
regimes = {"headline": headline_positions(), "M-march(inside box)": fixed(o_in, d_in), "M-packed": fixed(o_sp, d_sp), "default": default_config_positions()}
variants = sys.argv[1:] or ["impl=1", "impl=2"]
for name, x01 in regimes.items():
    n = x01.shape[0]
    dy = torch.randn(16, n, 2, device=dev)
    dt = torch.zeros_like(table); dx = torch.empty_like(x01)
    fns = []
    for v in variants:
        o = _lib.hash_bwd_default_opts()
        nodx = False
        for kv in v.split():
            k, val = kv.split("=")
            if k == "nodx":
                nodx = bool(int(val))
            else:
                setattr(o, k, int(val))
        nb = int(_lib.load().lse_hash_bwd_workspace_bytes(ctypes.byref(desc), ctypes.byref(o)))
        if nb:      # the product path's replica workspace (ops.hash_bwd_opts_with_workspace)
            o._ws = torch.zeros(nb // 4, dtype=torch.float32, device=dev)
            o.workspace, o.workspace_bytes = o._ws.data_ptr(), nb
        fns.append((v, (o, nodx)))
    times = {v: [] for v, _ in fns}
    for rnd in range(6):
        for v, (o, nodx) in fns:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(x01), P(dy), P(table), P(dt), None if nodx else P(dx), 0, 0, 16, n,
                      None, ctypes.byref(o), ops._stream())
            e1.record(); torch.cuda.synchronize()
            if rnd: times[v].append(e0.elapsed_time(e1))
    for v, _ in fns:
        t = sorted(times[v])
        print(f"{name:20s} {v:40s} median {t[len(t)//2]:.3f} ms  min {t[0]:.3f}", flush=True)
