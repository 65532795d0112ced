#!/usr/bin/env python3
"""Per-level cost of the hash backward (the backward twin of tools/hash_fwd_level_cost.py), in the three sample regimes the
bench reports: M-march (origins inside the box, fixed step), M-packed (SURVEY 8d sphere rays, fixed step) and the
default configuration (cone 0.004, visibility culling, carved grid).  For every level: kernel time of lse_hash_bwd_ex
restricted to that level (default kernel and the 16-lanes-per-sample kernel), distinct table entries / 32-B sectors / 64-B
lines per sample inside one 64-sample wave window (= the atomic requests a collision-free per-wave cache would send), and the
request rate that time corresponds to.  Writes a table to stdout (committed under profiles/)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from lsenerf_amd import ops, _lib
from bench_kernels import timeit
import bench

dev = torch.device("cuda", 0)
R, S = 4096, 1024
meta = ops.make_grid_meta()
g = torch.Generator().manual_seed(1)
table = ((torch.rand(meta.n_params, generator=g) * 2 - 1) * 1e-2).to(dev)
desc = meta.desc()
P = lambda t: ctypes.c_void_p(t.data_ptr())


def positions_fixed_step(o, d):
    step = 2 * 3 ** 0.5 / 1000
    ts = (0.05 + step * torch.arange(S, dtype=torch.float32)).repeat(R).to(dev)
    ri = torch.repeat_interleave(torch.arange(R, dtype=torch.int32), S).to(dev)
    cnt = torch.full((R,), S, dtype=torch.long)
    packed = torch.stack([torch.cumsum(cnt, 0) - cnt, cnt], -1).to(dev).contiguous()
    return ops.positions(o.to(dev), d.to(dev), ri, ts, ts + step, packed, True, None)[0]


def positions_default_config():
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    torch.manual_seed(96)
    model = LSENeRFModel(LSENeRFModelConfig(), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev).train()
    with torch.no_grad():
        model.field.mlp_base_grid.params.mul_(3000.0)
        model.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
    gg = torch.Generator().manual_seed(7)
    o, d = bench.sphere_rays(R, gg)
    rb = RayBundle(origins=o.to(dev), directions=d.to(dev), camera_indices=torch.zeros(R, 1, dtype=torch.long, device=dev))
    cb = model.update_occupancy_grid
    for s in range(0, 64, 16):
        cb(s)
    rs, _ = model.sampler(ray_bundle=rb, near_plane=0.05, far_plane=1e3, render_step_size=model.config.render_step_size,
                          alpha_thre=0.01, cone_angle=0.004)
    x01, _ = ops.positions(rb.origins, rb.directions, rs.ray_indices, rs.frustums.starts[..., 0].contiguous(),
                           rs.frustums.ends[..., 0].contiguous(), rs.packed_info, True, None)
    return x01


def window_stats(x01, level, max_waves=4096):
    """distinct entries / sectors / lines per sample within 64-sample windows (first max_waves windows)."""
    n = min(x01.shape[0] // 64 * 64, max_waves * 64)
    x = x01[:n]
    scale, res = meta.scales[level], meta.resolutions[level]
    size = meta.offsets[level + 1] - meta.offsets[level]
    pos = x * scale + 0.5
    p0 = pos.floor().to(torch.int64)
    idx = []
    for c in range(8):
        px, py, pz = p0[:, 0] + (c & 1), p0[:, 1] + ((c >> 1) & 1), p0[:, 2] + ((c >> 2) & 1)
        if res ** 3 <= size:
            i = (px + py * res + pz * res * res) % size
        else:
            i = ((px * 1) ^ (py * 2654435761) ^ (pz * 805459861)) & 0xFFFFFFFF
            i = i % size
        idx.append(i)
    idx = torch.stack(idx, -1).reshape(-1, 64 * 8)                       # [windows, 512]
    out = []
    for shift in (0, 2, 3):
        k = torch.sort(idx >> shift, dim=1).values
        distinct = 1 + (k[:, 1:] != k[:, :-1]).sum(1)
        out.append(float(distinct.float().mean()) / 64)
    return out


def level_time(x01, dy, level, **opt_kw):
    o = _lib.hash_bwd_default_opts()
    for k, v in opt_kw.items():
        setattr(o, k, v)
    n = x01.shape[0]
    dt = torch.zeros_like(table)
    dx = torch.empty_like(x01)
    f = lambda: _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(x01), P(dy), P(table), P(dt), P(dx), 0, level, level + 1, n,
                          None, ctypes.byref(o), ops._stream())
    return timeit(f, iters=7, warm=2)[0]


gen = torch.Generator().manual_seed(1)
o_in = torch.rand(R, 3, generator=gen) - 0.5
d_in = torch.randn(R, 3, generator=gen); d_in = d_in / d_in.norm(dim=-1, keepdim=True)
o_sp, d_sp = bench.sphere_rays(R, torch.Generator().manual_seed(96))
regimes = {"M-march (origins in [-0.5,0.5]^3, fixed step)": positions_fixed_step(o_in, d_in),
           "M-packed (radius-1.5 sphere rays, fixed step)": positions_fixed_step(o_sp, d_sp),
           "default config (cone 0.004, culled, carved grid)": positions_default_config()}
for name, x01 in regimes.items():
    n = x01.shape[0]
    dy = torch.randn(meta.n_levels, n, 2, device=dev)
    full = {}
    for label, kw in (("default", {}), ("lanes16", {"impl": 0})):
        o = _lib.hash_bwd_default_opts()
        for k, v in kw.items():
            setattr(o, k, v)
        dt = torch.zeros_like(table); dx = torch.empty_like(x01)
        f = lambda: _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(x01), P(dy), P(table), P(dt), P(dx), 0, 0, meta.n_levels, n,
                              None, ctypes.byref(o), ops._stream())
        full[label] = timeit(f, iters=7, warm=2)[0]
    print(f"\n== {name}: N = {n} samples ({n / R:.0f} per ray); all 16 levels: default kernel {full['default']:.3f} ms "
          f"({full['default'] * 1e6 / n:.3f} ns/sample), 16-lane kernel {full['lanes16']:.3f} ms", flush=True)
    print(f"{'lvl':>3} {'res':>5} {'entries':>8} | {'t_default':>9} {'t_lanes16':>9} ms | per sample in a 64-sample window: "
          f"{'entries':>7} {'sectors':>7} {'lines':>7} | {'G sector-req/s at t_default':>10}")
    tot = [0.0, 0.0, 0.0, 0.0, 0.0]
    for l in range(meta.n_levels):
        td, t16 = level_time(x01, dy, l), level_time(x01, dy, l, impl=0)
        e, s, ln = window_stats(x01, l)
        tot = [tot[0] + td, tot[1] + t16, tot[2] + e, tot[3] + s, tot[4] + ln]
        print(f"{l:3d} {meta.resolutions[l]:5d} {meta.offsets[l + 1] - meta.offsets[l]:8d} | {td:9.4f} {t16:9.4f}    |"
              f" {'':34s} {e:7.2f} {s:7.2f} {ln:7.2f} | {s * n / (td * 1e-3) / 1e9:10.1f}", flush=True)
    print(f"sum {'':14s} | {tot[0]:9.4f} {tot[1]:9.4f}    | {'':34s} {tot[2]:7.2f} {tot[3]:7.2f} {tot[4]:7.2f} |")
