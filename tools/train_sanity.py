#!/usr/bin/env python3
"""End-to-end sanity of the training path: a student field is fitted to renders of a teacher field (synthetic scene, the reference's
default sampler configuration, occupancy refresh every 16 steps, colour + event bundles through train_step_bundles; second argument
"mlp": with the two MLP intensity mappers instead of identity / powpow) for a few hundred
steps, once with the eager step, once with the captured step (lsenerf_amd.graph.GraphedTrainStep) and once with the captured step that
marches the next step's rays on a side stream.  Prints the loss curves; they must fall together (same rays, same targets; the jitter streams differ).  usage: python tools/train_sanity.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle, _lib
from lsenerf_amd.graph import GraphedTrainStep
from lsenerf_amd.optim import FlatAdam, FlatParams

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
_lib.load()
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg = dict(use_mapping=True, mapping_method="identity", map_mode="co_map", evs_mapping_method="powpow")
if len(sys.argv) > 2 and sys.argv[2] == "mlp":       # the MLP intensity mappers of R:lse_nerf/intensity_mappers.py:28-62, trained inside the
    cfg = dict(use_mapping=True, mapping_method="rgb_mlp", map_mode="co_map", evs_mapping_method="mlp")     # fused loss epilogue (ABI 6)
torch.manual_seed(1)
teacher = LSENeRFModel(LSENeRFModelConfig(**cfg), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev).train()
with torch.no_grad():
    teacher.field.mlp_base_grid.params.mul_(3000.0)
    teacher.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
for s_ in range(0, 64, 16):
    teacher.update_occupancy_grid(s_)
g = torch.Generator().manual_seed(7)
N_POOL, sizes = 32768, (2316, 597, 597)
o, d = bench.sphere_rays(N_POOL, g)
aid = torch.randint(0, 64, (N_POOL,), generator=g)
pool = RayBundle(origins=o.to(dev), directions=d.to(dev), camera_indices=torch.zeros(N_POOL, 1, dtype=torch.long, device=dev),
                 metadata={"appearance_id": aid.to(dev)})
teacher.eval()
with torch.no_grad():
    tgt = torch.cat([teacher.exec_get_outputs(RayBundle(origins=pool.origins[i:i + 4096], directions=pool.directions[i:i + 4096],
                                                         camera_indices=pool.camera_indices[i:i + 4096],
                                                         metadata={"appearance_id": pool.metadata["appearance_id"][i:i + 4096]}))["rgb"]
                     for i in range(0, N_POOL, 4096)]).clamp(1e-5, 1.0)
print("teacher render: mean", float(tgt.mean()), "std", float(tgt.std()))


def pick(idx):
    return RayBundle(origins=pool.origins[idx], directions=pool.directions[idx], camera_indices=pool.camera_indices[idx],
                     metadata={"appearance_id": pool.metadata["appearance_id"][idx]})


def batch_of(it):
    gi = torch.Generator().manual_seed(1000 + it)
    ic = torch.randint(0, N_POOL, (sizes[0],), generator=gi).to(dev)
    ip = torch.randint(0, N_POOL - 1, (sizes[1],), generator=gi).to(dev)
    col, prev, nxt = pick(ic), pick(ip), pick(ip + 1)
    gray = lambda x: (x * x.new_tensor([0.2989, 0.5870, 0.1140])).sum(-1, keepdim=True)
    evs = torch.log(gray(tgt[ip + 1]) + 1e-6) - torch.log(gray(tgt[ip]) + 1e-6)
    return col, prev, nxt, {"col_batch": {"image": tgt[ic]}, "evs_batch": {"image": evs}}


results = {}
for mode in ("eager", "graphed", "graphed_prefetch"):
    torch.manual_seed(2)
    student = LSENeRFModel(LSENeRFModelConfig(**cfg), torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev).train()
    opt = FlatAdam(FlatParams(student.get_param_groups()["fields"]), lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=20000)
    step = None
    curve = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    upcoming = batch_of(0)
    for it in range(STEPS):
        student.update_occupancy_grid(it)
        col, prev, nxt, batch = upcoming          # the very tensors that were announced (the graph checks their identity)
        upcoming = batch_of(it + 1)
        if mode == "graphed_prefetch":      # the next step's rays are announced one step early and marched on the side stream
            if step is None:
                step = GraphedTrainStep(student, opt, col, prev, nxt, batch, prefetch_march=True)
            losses = step(col, prev, nxt, batch, next_bundles=upcoming[:3])
        elif mode == "eager":
            opt.zero_grad()
            _, losses, _ = student.train_step_bundles(col, prev, nxt, batch)
            sum(losses.values()).backward()
            opt.step()
        else:
            if step is None:
                step = GraphedTrainStep(student, opt, col, prev, nxt, batch)
            losses = step(col, prev, nxt, batch)
        if it % 25 == 0 or it == STEPS - 1:
            curve.append((it, float(losses["rgb_loss"]), float(losses["event_loss"])))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if step is not None:
        step.check_overflow()
        if mode == "graphed_prefetch":
            print("   calls whose rays were not the announced ones (re-marched):", step.remarched_unannounced)
            assert step.remarched_unannounced == 0
    student.occupancy_grid.check_deferred_overflow()
    finite = bool(torch.isfinite(opt.flat.data).all())
    results[mode] = {"seconds": dt, "ms_per_step": dt / STEPS * 1e3, "finite_parameters": finite, "curve": curve,
                     "occupied_fraction": float(student.occupancy_grid.binaries.float().mean())}
    print(mode, f"{dt / STEPS * 1e3:.2f} ms/step  finite={finite}  occupied={results[mode]['occupied_fraction']:.3f}")
    for it, a, b in curve:
        print(f"   step {it:4d}  rgb_loss {a:.5f}  event_loss {b:.5f}")
ok = all(r["finite_parameters"] and r["curve"][-1][1] < 0.35 * r["curve"][0][1] for r in results.values())
print("train_sanity:", "OK" if ok else "FAILED")
print(json.dumps(results))
sys.exit(0 if ok else 1)
