#!/usr/bin/env python3
"""Does capturing a step survive EARLIER eager losses of the same model that are still alive?  (Round 4: it crashed the HIP runtime
in hipStreamEndCapture through stale AccumulateGrad nodes of the epilogue's scalar parameters.)  Since round 5 nothing inside a
captured step hands a gradient to an AccumulateGrad node that the step did not create itself: parameters of the fast path and of
the torch route accumulate directly (ops._direct_grad, ops.direct_grad_params), the rays are fresh leaves per run, warm-up and
capture share a stream (lsenerf_amd/graph.py).  torch's own diagnosis of the hazard -- "The AccumulateGrad node's stream does not
match the stream of the node that produced the incoming gradient" -- is an ERROR here, and TORCH_WARN_ONCE is lifted so that every
occurrence counts.  Run as its own process: a regression may be a segfault.

Cases: the cfg 2 and cfg 3 compositions at full size on the fused epilogue, and cfg 2 on the TORCH route of the loss
(a mapper subclass the fused epilogue does not know: Powpow / ThreeToOne parameters consumed by plain torch ops), each with ray gradients."""
import os, sys, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
torch.set_warn_always(True)
warnings.filterwarnings("error", message=".*AccumulateGrad node's stream.*")
from tests.test_gpu_fullsize import _build
from lsenerf_amd.graph import GraphedTrainStep
for kind in ("cfg2", "cfg3", "cfg2_torch_route"):
    m, opt, bundles, batch, jit = _build(kind.split("_")[0])
    if kind.endswith("torch_route"):
        # a mapper class the fused epilogue does not know (a user's subclass): the plan falls back to the torch routing, where the
        # Powpow / ThreeToOne parameters are consumed by plain torch ops
        m.evs_mapper.__class__ = type("UserPowpow", (type(m.evs_mapper),), {})
        assert m._epilogue_desc() is None, "expected the torch route of the loss"
    keep = []
    for _ in range(2):          # eager steps whose losses / outputs stay alive, on the default stream
        opt.zero_grad()
        out, losses, _ = m.train_step_bundles(*bundles, batch, jitter=torch.cat([j for j in jit if j is not None]))
        sum(losses.values()).backward(retain_graph=True)
        keep.append((out, losses))
    step = GraphedTrainStep(m, opt, *bundles, batch, ray_grads=True, jitter="input")
    for _ in range(2):
        l = step(*bundles, batch, jitter=torch.cat([j for j in jit if j is not None]))
    torch.cuda.synchronize()
    rg = step.ray_grads["col"]
    assert rg is not None and bool(torch.isfinite(rg[0]).all()) and float(rg[0].abs().max()) > 0
    print(kind, "captured and replayed with", len(keep), "eager graphs alive:", {k: float(v) for k, v in l.items()}, flush=True)
    step.close()
print("OK")
