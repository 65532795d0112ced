#!/usr/bin/env python3
"""Does capturing a step survive EARLIER eager losses of the same model that are still alive?  (Round 4: it crashed the HIP runtime
in hipStreamEndCapture through stale AccumulateGrad nodes of the epilogue's scalar parameters; since those gradients are accumulated
directly, no AccumulateGrad node of a model parameter runs inside the step.)  Run as its own process: a failure is a segfault."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tests.test_gpu_fullsize import _build
from lsenerf_amd.graph import GraphedTrainStep
for kind in ("cfg2", "cfg3"):
    m, opt, bundles, batch, jit = _build(kind)
    keep = []
    for _ in range(2):          # eager steps whose losses / outputs stay alive, on the default stream
        opt.zero_grad()
        out, losses, _ = m.train_step_bundles(*bundles, batch, jitter=torch.cat([j for j in jit if j is not None]))
        sum(losses.values()).backward(retain_graph=True)
        keep.append((out, losses))
    step = GraphedTrainStep(m, opt, *bundles, batch, ray_grads=True, jitter="input")
    l = step(*bundles, batch, jitter=torch.cat([j for j in jit if j is not None]))
    torch.cuda.synchronize()
    print(kind, "captured and replayed with", len(keep), "eager graphs alive:", {k: float(v) for k, v in l.items()}, flush=True)
    step.close()
print("OK")
