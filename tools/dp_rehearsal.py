#!/usr/bin/env python3
"""Data-parallel rehearsal of BASELINE config 5 with the REAL HIP model (R:lse_nerf/lse_pipeline.py:95-98,
R:train.py:104,146-167): W ranks share one MI355X over gloo (the collectives stage through the host -- this checks
values and control flow, not speed), each renders its shard of one ray batch, and after a few optimizer steps the
parameters must equal those of ONE process trained on the whole batch.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        tools/dp_rehearsal.py --out gpurun_out/dp_rehearsal.json

For every exchange of lsenerf_amd.dist -- plain (one blocking all-reduce), pipelined (GradPipeline: all-reduce + Adam
finished behind the next step's ray marcher, before the visibility pre-pass reads the parameters), sharded
(reduce-scatter -> Adam on 1/W -> all-gather) and overlap (two-launch hash backward, early all-reduce of the fine levels)
-- the script runs: occupancy refresh at step 0 (warm-up branch: all cells), 3 train steps in the reference's default
configuration (cone 0.004, alpha_thre 0.01 => sigma_fn pre-pass on, 4-level 128^3 grid), occupancy refresh at step 320
(sampled branch), and asserts
  * the first step's rank-averaged gradient == the single-process full-batch gradient within 3e-6 * max|g| per parameter
    tensor (the clean statement of "sharding + one all-reduce == full batch": nothing has amplified the float-atomic
    summation noise yet; two runs of the SAME single process differ by 1e-6 * max|g|),
  * flat parameters after 3 Adam steps == single-process parameters within 1e-6 * max|p| for all but a vanishing
    fraction of the elements, and nowhere further away than a second single-process run is from the first (Adam with
    eps 1e-15 turns a gradient element that is pure summation noise into a +-lr step, and float atomics make that noise
    differ from run to run even inside ONE process -- the rehearsal measures that floor and reports it),
  * parameters bit-identical on all ranks,
  * occs / binaries bit-identical on all ranks after both refreshes although every rank seeds its global RNG
    differently: the estimator's own (update_seed, step) stream makes the ranks compute the same grid (recorded as
    "grids_identical_before_the_broadcast": true in a quiet process, occasionally false in the last bits when two
    processes share one GPU as here), and dist.sync_grid then broadcasts rank 0's grid, which is what is asserted
    (the reference relies on DDP's buffer broadcast alone).
Launch it BEFORE anything else touches the GPU in the calling shell command; ranks must not be spawned from a process
that has initialised the GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RAYS = 4096
STEPS = 3


def make_model(dev):
    from lsenerf_amd import LSENeRFModel, LSENeRFModelConfig
    torch.manual_seed(96)                                          # identical initial parameters everywhere
    cfg = LSENeRFModelConfig()                                     # reference defaults: pre-pass on, cone 0.004, 4x128^3
    model = LSENeRFModel(cfg, torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), num_train_data=64).to(dev)
    with torch.no_grad():                                          # a field with visible structure (see tools/bench_context.py)
        model.field.mlp_base_grid.params.mul_(3000.0)
        model.field.mlp_base_mlp.params[-16 * 64:-15 * 64].mul_(6.0)
    model.train()
    return model


def make_batch(dev):
    g = torch.Generator().manual_seed(7)
    o = torch.randn(RAYS, 3, generator=g)
    o = 1.5 * o / o.norm(dim=-1, keepdim=True)
    d = (torch.rand(RAYS, 3, generator=g) - 0.5) - o
    d = d / d.norm(dim=-1, keepdim=True)
    return {"o": o.to(dev), "d": d.to(dev), "target": torch.rand(RAYS, 3, generator=g).to(dev),
            "aid": torch.randint(0, 64, (RAYS,), generator=g).to(dev), "jitter": torch.rand(RAYS, generator=g).to(dev)}


def run(mode: str, rank: int, world: int, dev, batch):
    """Returns (flat parameters, samples rendered by this rank per step, grid consistency flags)."""
    from lsenerf_amd import RayBundle, dist as ldist
    from lsenerf_amd.optim import FlatAdam, FlatParams
    single = mode == "single"
    w = 1 if single else world
    sl = slice(0, RAYS) if single else ldist.shard_rays(RAYS, rank, world)
    model = make_model(dev)
    torch.manual_seed(1000 + rank)                                 # R:train.py:104: every rank its own global RNG from here on
    flat = FlatParams(model.get_param_groups()["fields"], total_multiple=world * 64)
    if not single:
        ldist.broadcast_params(flat.data)
    opt = FlatAdam(flat, lr=1e-2, eps=1e-15, lr_final=1e-4, max_steps=200000)
    pipe = sharded = exchange = None
    if mode == "pipelined":
        pipe = ldist.GradPipeline(opt, w).attach(model.occupancy_grid)
    elif mode == "sharded":
        sharded = ldist.ShardedAdamExchange(flat, lr=1e-2, eps=1e-15)
    elif mode == "overlap":
        grid = model.field.mlp_base_grid
        exchange = ldist.OverlappedGradExchange(flat, grid.params, grid.meta.offsets, split_level=6)
        exchange.install()
    rb = RayBundle(origins=batch["o"][sl].clone().requires_grad_(True), directions=batch["d"][sl].clone().requires_grad_(True),
                   camera_indices=torch.zeros(sl.stop - sl.start, 1, dtype=torch.long, device=dev),
                   metadata={"appearance_id": batch["aid"][sl]})
    refresh = model.update_occupancy_grid
    grids_ok = []
    pre_sync = []
    if not single:     # rank 0's grid after every refresh; what the ranks computed on their own is recorded first
        model.occupancy_grid.after_update_hook = lambda: (pre_sync.append(ldist.check_grid_consistency(model.occupancy_grid)),
                                                          ldist.sync_grid(model.occupancy_grid))

    def refresh_grid(step):
        if pipe is not None:
            pipe.flush()                                           # the refresh reads the parameters
        refresh(step)
        grids_ok.append(True if single else ldist.check_grid_consistency(model.occupancy_grid))

    if not single and os.environ.get("LSE_DIAG"):
        # which stage of the density evaluation differs between the ranks (same parameters, same cells)?
        est, fld = model.occupancy_grid, model.field

        def cks(t):
            v = t.detach().contiguous().view(-1)
            v = (v.to(torch.uint8) if v.dtype == torch.bool else v)
            v = (v.view(torch.int32) if v.element_size() == 4 else v).to(torch.int64)
            w = (torch.arange(v.numel(), device=v.device) % 1000003) + 1
            return torch.stack([v.sum(), (v * w).sum()])
        names, vals, keep = [], [], {}
        names.append("params"); vals.append(cks(flat.data))
        for lvl, (indices, x) in enumerate(est._update_samples(0, 256, est._update_generator(0))):
            with torch.no_grad():
                x01, sel = fld._x01(x.reshape(-1, 3).contiguous(), None, None, None, None, None)
                y = fld.mlp_base_grid.forward_levelmajor(x01)
                h, sigma = fld._base_mlp(y, sel, x01.shape[0])
            for nm, t in (("x", x), ("x01", x01), ("y", y), ("h", h), ("sigma", sigma)):
                names.append(f"{nm}{lvl}"); vals.append(cks(t))
            keep[lvl] = (x01, y)
        mine = torch.stack(vals)
        other = mine.clone()
        tdist.broadcast(other, src=0)
        bad = [names[i] for i in range(len(names)) if not torch.equal(other[i], mine[i])]
        if rank == 1:
            print("DIAG", mode, "stages that differ from rank 0:", bad, flush=True)
            for lvl, (x01, y) in keep.items():
                if f"y{lvl}" in bad and f"x01{lvl}" not in bad:
                    y2 = fld.mlp_base_grid.forward_levelmajor(x01)
                    ne = (y2 != y).any(dim=2)
                    print("DIAG   level-eval", lvl, ": re-evaluated y equals first y:", bool(torch.equal(y2, y)), "differing samples per table level",
                          ne.sum(dim=1).tolist(), flush=True)
    refresh_grid(0)
    n_samples = []
    first_grad = None
    for step in range(STEPS):
        rb.origins.grad = rb.directions.grad = None
        if exchange is not None:
            exchange.begin_step(1)
        # pipelined: sampling() fires GradPipeline.flush after the marcher and before the sigma_fn pre-pass
        out = model.exec_get_outputs(rb, jitter=batch["jitter"][sl])
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(out["rgb"], batch["target"][sl])
        loss.backward()
        n_samples.append(int(out["num_samples_per_ray"].sum()))
        if step == 0:                                              # rank-averaged gradient of the first step
            first_grad = flat.grad.clone()
            if not single:
                ldist.allreduce_grads(first_grad)
                first_grad /= w
        if pipe is not None:
            pipe.start()
        elif sharded is not None:
            sharded.lr = opt.current_lr()
            opt.step_count += 1
            sharded.step()
        else:
            if exchange is not None:
                exchange.finish()
            elif not single:
                ldist.allreduce_grads(flat.grad)
            opt.step(grad_scale=1.0 / w)
    if pipe is not None:
        pipe.flush()
    refresh_grid(320)
    if exchange is not None:
        exchange.uninstall()
    spans = [(o, o + p.numel()) for p, o in zip(flat.params, flat.offsets)]
    run.pre_sync = pre_sync
    return flat.data.clone(), n_samples, grids_ok, float(model.occupancy_grid.binaries.float().mean()), first_grad, spans


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dp_rehearsal.json"))
    ap.add_argument("--modes", default="plain,pipelined,sharded,overlap")
    args = ap.parse_args()
    from lsenerf_amd import _lib, dist as ldist
    rank, world, _ = ldist.init_from_env("gloo")
    assert world >= 2, "launch with torch.distributed.run --nproc-per-node 2"
    assert torch.cuda.is_available(), "the rehearsal trains the HIP model: it needs the MI355X"
    dev = torch.device("cuda", 0)                                  # all ranks share the one GPU of the box
    torch.cuda.set_device(dev)
    _lib.load()
    batch = make_batch(dev)

    ref, n_ref, _, occ_ref, g_ref, spans = run("single", rank, world, dev, batch)   # every rank: the same full-batch reference
    ref2, _, _, _, g_ref2, _ = run("single", rank, world, dev, batch)               # ... twice: the run-to-run floor
    scale = float(ref.abs().max())

    def grad_err(g):        # per parameter tensor: max|g - g_ref| / max|g_ref|
        return max(float((g[a:b] - g_ref[a:b]).abs().max()) / max(float(g_ref[a:b].abs().max()), 1e-30) for a, b in spans)

    if rank == 0 and os.environ.get("LSE_DIAG"):
        for (a_, b_) in spans:
            print("DIAG span", a_, b_, "ref2 vs ref", float((g_ref2[a_:b_] - g_ref[a_:b_]).abs().max()) / max(float(g_ref[a_:b_].abs().max()), 1e-30),
                  "n_ref", n_ref, flush=True)
    floor_g = grad_err(g_ref2)
    floor_p = float((ref2 - ref).abs().max())
    floor_frac = float(((ref2 - ref).abs() > 1e-6 * scale).float().mean())
    report = {"world": world, "rays": RAYS, "steps": STEPS, "backend": "gloo (2 ranks on one MI355X)",
              "config": "reference defaults: cone 0.004, alpha_thre 0.01 (sigma_fn pre-pass on), 4-level 128^3 grid",
              "single_process": {"samples_per_step": n_ref, "occupied_fraction_after_refresh": occ_ref,
                                 "run_to_run_first_grad_err": grad_err(g_ref2), "run_to_run_max_abs_param_diff": floor_p,
                                 "run_to_run_fraction_of_params_beyond_1e-6_max": floor_frac},
              "tolerance": "first-step gradient: per-tensor max err <= max(3e-6, 3 x run-to-run) * max|g|; parameters after 3 "
                           "Adam steps: fraction beyond 1e-6 * max|p| <= max(1e-4, 2 x single-process run-to-run fraction) "
                           "and max|p_dp - p_single| <= max(1e-6 * max|p|, 3 x run-to-run max diff); samples per step "
                           "within 1e-5 relative (visibility-threshold flips on parameters that differ in the last bits)",
              "max_abs_param": scale, "modes": {},
              "note": "Two processes share ONE GPU here, which the production layout (one process per GPU) never does.  In that "
                      "situation ~0.3 % of the kernel launches of EITHER process were seen to return one 128-byte line of hash "
                      "features that a re-evaluation of the same inputs does not reproduce (tools/refresh_determinism.py: 0 of "
                      "1200 evaluations differ when the process has the GPU to itself).  Such an event flips a visibility decision "
                      "or an occupancy bit on one rank; the grid broadcast (dist.sync_grid) contains the second, the first can push "
                      "a mode over the parameter tolerance -- rerun in that case (about one run in four is affected)."}
    ok_all = True
    for mode in args.modes.split(","):
        p, n_s, grids_ok, occ, g1, _ = run(mode, rank, world, dev, batch)
        err = float((p - ref).abs().max())
        frac = float(((p - ref).abs() > 1e-6 * scale).float().mean())
        gerr = grad_err(g1)
        n_tot = torch.tensor(n_s, dtype=torch.int64, device=dev)
        tdist.all_reduce(n_tot)
        p0 = p.clone()
        tdist.broadcast(p0, src=0)
        same = torch.tensor([1 if torch.equal(p0, p) else 0], dtype=torch.int32, device=dev)
        tdist.all_reduce(same, op=tdist.ReduceOp.MIN)
        entry = {"first_step_grad_err_vs_single": gerr, "max_abs_err_vs_single": err, "rel_to_max_param": err / scale,
                 "fraction_of_params_beyond_1e-6_max": frac,
                 "within_tolerance": gerr <= max(3e-6, 3 * floor_g) and frac <= max(1e-4, 2 * floor_frac)
                 and err <= max(1e-6 * scale, 3 * floor_p),
                 "params_bit_identical_across_ranks": bool(same.item()),
                 "grids_bit_identical_across_ranks_after_refresh_step0_step320": grids_ok,
                 "grids_identical_before_the_broadcast": list(getattr(run, "pre_sync", [])),
                 "samples_per_step_all_ranks": n_tot.tolist(), "samples_match_single_process": all(abs(a - b) <= 1e-5 * b for a, b in zip(n_tot.tolist(), n_ref)),
                 "occupied_fraction_after_refresh": occ}
        report["modes"][mode] = entry
        ok = entry["within_tolerance"] and entry["params_bit_identical_across_ranks"] and all(grids_ok) \
            and entry["samples_match_single_process"]
        ok_all = ok_all and ok
        if rank == 0:
            print(mode, json.dumps(entry), flush=True)
    report["all_ok"] = ok_all
    if rank == 0:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)
        print("dp_rehearsal:", "OK" if ok_all else "FAILED", "->", args.out, flush=True)
    tdist.barrier()
    tdist.destroy_process_group()
    if not ok_all:
        sys.exit(1)


if __name__ == "__main__":
    main()
