#!/usr/bin/env python3
"""Does the ray marcher of the NEXT step hide behind the hash backward of the current one?  Times, in the reference's default
configuration (carved grid, cone 0.004): the marcher alone, the hash backward alone, and both issued together on two streams."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from lsenerf_amd import ops, _lib, LSENeRFModel, LSENeRFModelConfig, RayBundle
dev = torch.device("cuda", 0)
R = 3510
torch.manual_seed(96)
model = LSENeRFModel(LSENeRFModelConfig(), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev).train()
with torch.no_grad():
    model.field.mlp_base_grid.params.mul_(3000.0)
    model.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
o, d = bench.sphere_rays(R, torch.Generator().manual_seed(7))
o, d = o.to(dev), d.to(dev)
for s_ in range(0, 64, 16):
    model.update_occupancy_grid(s_)
cfg, est = model.config, model.occupancy_grid
rb = RayBundle(origins=o, directions=d, camera_indices=torch.zeros(R, 1, dtype=torch.long, device=dev))
rs, _ = model.sampler(ray_bundle=rb, near_plane=cfg.near_plane, far_plane=cfg.far_plane, render_step_size=cfg.render_step_size,
                      alpha_thre=cfg.alpha_thre, cone_angle=cfg.cone_angle)
x01 = ops.positions(o, d, rs.ray_indices, rs.frustums.starts[..., 0].contiguous(), rs.frustums.ends[..., 0].contiguous(),
                    rs.packed_info, True, None)[0]
n = x01.shape[0]
meta = model.field.mlp_base_grid.meta
desc = meta.desc()
table = model.field.mlp_base_grid.params.detach()
dy = torch.randn(16, n, 2, device=dev)
dt = torch.zeros_like(table); dx = torch.empty_like(x01)
opts = _lib.hash_bwd_default_opts()
nb = int(_lib.load().lse_hash_bwd_workspace_bytes(ctypes.byref(desc), ctypes.byref(opts)))
if nb:
    ws = torch.zeros(nb // 4 + 1, dtype=torch.float32, device=dev)
    opts.workspace, opts.workspace_bytes = ws.data_ptr(), nb
P = lambda t: ctypes.c_void_p(t.data_ptr())
near, far = ops.ray_planes(R, dev, cfg.near_plane, cfg.far_plane, None, None, torch.rand(R, device=dev), cfg.render_step_size)
cap = est._cap_per_ray(cfg.near_plane, cfg.far_plane, cfg.render_step_size, cfg.cone_angle)
def march():
    return ops.traverse_grids_deferred(o, d, est._binaries_u8(), est.aabbs, near, far, cfg.render_step_size, cfg.cone_angle, cap)
def hash_bwd():
    _lib.call("lse_hash_bwd_ex", ctypes.byref(desc), P(x01), P(dy), P(table), P(dt), P(dx), 0, 0, 16, n, None, ctypes.byref(opts), ops._stream())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]
def both(first_hash=True):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    if first_hash:
        with torch.cuda.stream(s1): hash_bwd()
        with torch.cuda.stream(s2): march()
    else:
        with torch.cuda.stream(s2): march()
        with torch.cuda.stream(s1): hash_bwd()
    cur.wait_stream(s1); cur.wait_stream(s2)
print(f"samples {n}, rays {R}, cap per ray {cap}")
tm, th = timed(march), timed(hash_bwd)
print(f"marcher alone {tm:.3f} ms   hash backward alone {th:.3f} ms   sum {tm + th:.3f}")
print(f"both, hash backward issued first {timed(lambda: both(True)):.3f} ms   marcher issued first {timed(lambda: both(False)):.3f} ms")
