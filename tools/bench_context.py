"""Context runs next to bench.py ("a third run with the reference defaults ... reported as default-config
context", and the M-packed variant).  Not the contract benchmark: bench.py stays the number of record.

default-config: the reference's own training configuration of the hot path -- 4-level 128^3 occupancy grid, cone_angle
0.004, alpha_thre 0.01, early_stop_eps 1e-4, stratified jitter, the sigma_fn visibility pre-pass ON -- on a synthetic
"trained-like" scene: the density head is biased so that a band of the volume is opaque, the occupancy grid is carved by
the reference's own update rule (warm-up sweeps over all cells), and rays come from a sphere of radius 1.5 toward random
targets in [-0.5, 0.5]^3.  Reports rays/s, samples per ray before / after culling, and the occupied fraction.

Usage: python tools/bench_context.py [--steps K] [--warmup W] [--rays R]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rays", type=int, default=4096)
    args = ap.parse_args()
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig, RayBundle
    from lsenerf_amd.optim import FlatAdam, FlatParams
    dev = torch.device("cuda", 0)
    torch.manual_seed(96)
    cfg = LSENeRFModelConfig()                                   # reference defaults (R:lse_nerf/lsenerf.py config)
    model = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev)
    model.train()
    with torch.no_grad():                                        # make the field spatially varied and partly opaque
        model.field.mlp_base_grid.params.mul_(3000.0)
        w = model.field.mlp_base_mlp.params
        w[-16 * 64:-15 * 64].mul_(6.0)                           # output neuron 0 = log-density
    flat = FlatParams(model.get_param_groups()["fields"])
    opt = FlatAdam(flat, lr=1e-3, eps=1e-15)
    g = torch.Generator().manual_seed(7)
    R = args.rays
    o = torch.randn(R, 3, generator=g)
    o = 1.5 * o / o.norm(dim=-1, keepdim=True)
    tgt = torch.rand(R, 3, generator=g) - 0.5
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    rb = RayBundle(origins=o.to(dev).requires_grad_(True), directions=d.to(dev).requires_grad_(True),
                   camera_indices=torch.zeros(R, 1, dtype=torch.long, device=dev),
                   metadata={"appearance_id": torch.randint(0, 64, (R,), generator=g).to(dev)})
    target = torch.rand(R, 3, generator=g).to(dev)
    cb = model.update_occupancy_grid
    for step in range(0, 64, 16):                                # carve the grid with the reference's update rule
        cb(step)
    occ_frac = float(model.occupancy_grid.binaries.float().mean())

    def train_step(step):
        cb(step)                                                 # every 16th step refreshes the grid (inside the timing)
        opt.zero_grad()
        rb.origins.grad = None
        rb.directions.grad = None
        out = model.exec_get_outputs(rb)
        loss = torch.nn.functional.mse_loss(out["rgb"], target)
        loss.backward()
        opt.step()
        return out

    step = 65
    for _ in range(args.warmup):
        out = train_step(step)
        step += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_tot = 0
    for _ in range(args.steps):
        out = train_step(step)
        step += 1
        n_tot += int(out["num_samples_per_ray"].sum())
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"context": "default-config (cone 0.004, alpha_thre 0.01, pre-pass on, carved 4-level grid)",
                      "rays_per_s": R * args.steps / el, "ms_per_step": el / args.steps * 1e3, "rays": R,
                      "samples_per_ray_after_culling": n_tot / args.steps / R, "occupied_fraction": occ_frac,
                      "steps": args.steps}))


if __name__ == "__main__":
    main()
