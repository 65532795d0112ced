"""Flat-buffer optimiser state for the hot path.

All trainable tensors of a model are re-pointed into ONE contiguous fp32 buffer (``FlatParams``): their ``.data``
and ``.grad`` become views, so that
  * the gradient of the whole model is one tensor -> one RCCL all-reduce per step over xGMI (lsenerf_amd.dist),
  * Adam is one streaming HIP kernel over (p, g, m, v)  (lse_adam_step),
  * zeroing gradients is one memset.
Semantics are torch.optim.Adam's with the reference's hyper-parameters (lr 1e-2, eps 1e-15, exponential decay to
1e-4 over 200k steps: R:lse_nerf/lse_config.py:29-33).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
from torch import nn

from . import ops


class FlatParams:
    def __init__(self, params: Iterable[nn.Parameter], align: int = 64, total_multiple: int = 1):
        """``total_multiple``: round the buffer length up (e.g. W * 64 so that dist.ShardedAdamExchange shards it in place)."""
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        assert len(self.params) > 0
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            offs.append(total)
            total += (p.numel() + align - 1) // align * align
        total = (total + total_multiple - 1) // total_multiple * total_multiple
        self.offsets, self.numel = offs, total
        self.data = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):   # keep .grad a view (autograd then accumulates in place)
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FlatAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps) on a FlatParams, one HIP kernel per step."""

    def __init__(self, flat: FlatParams, lr: float = 1e-2, betas=(0.9, 0.999), eps: float = 1e-15,
                 lr_final: Optional[float] = None, max_steps: Optional[int] = None):
        self.flat = flat
        self.lr_init, self.betas, self.eps = lr, betas, eps
        self.lr_final, self.max_steps = lr_final, max_steps
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = 0

    def current_lr(self) -> float:
        """nerfstudio ExponentialDecayScheduler (no warm-up): lr_init * (lr_final/lr_init)^(t/max_steps)."""
        if self.lr_final is None or not self.max_steps:
            return self.lr_init
        t = min(max(self.step_count / self.max_steps, 0.0), 1.0)
        import math
        return math.exp(math.log(self.lr_init) * (1 - t) + math.log(self.lr_final) * t)

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, grad_scale: float = 1.0):
        lr = self.current_lr()
        self.step_count += 1
        ops.adam_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1],
                      self.eps, self.step_count, grad_scale)

    # -- captured steps: the launch inside a HIP graph cannot carry this step's scalars, so they live in device memory ----------
    # The optimizer clock is ON THE DEVICE: an int64 step counter that `lse_adam_schedule_dev` -- captured in front of the Adam
    # kernel -- advances and turns into (lr, 1 - beta1^t, 1 / sqrt(1 - beta2^t)).  A replay therefore reads nothing the host writes
    # per step.  (Until round 3 the host staged the three floats in ONE pinned buffer and queued an asynchronous copy per step: a
    # host that runs ahead of the device -- which is the point of replaying a graph -- overwrote the buffer before the queued copy
    # of an earlier step had executed, so early steps could take a later step's bias corrections.)
    def _device_clock(self):
        if not hasattr(self, "_hyper_dev"):
            dev = self.flat.data.device
            self._hyper_dev = torch.zeros(6, dtype=torch.float32, device=dev)
            self._step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
            self._sched_dev = torch.zeros(6, dtype=torch.float64, device=dev)
            self._step_dev_expected = None
            self._sched_uploaded = None
        return self._step_dev, self._hyper_dev

    def _schedule_constants(self):
        return (float(self.lr_init), float(self.lr_final) if self.lr_final else 0.0, float(self.max_steps) if self.max_steps else 0.0,
                float(self.betas[0]), float(self.betas[1]), float(self.eps))

    def prepare_step(self) -> None:
        """Host bookkeeping of one captured step: makes sure the device-side step counter equals ``step_count`` (a stream-ordered
        fill with the value in the launch arguments, only after the host-side count was changed from outside: construction,
        load_state_dict, a restored snapshot), that the device-side schedule constants are the host's current ones (lr_init,
        lr_final, max_steps, betas, eps: six doubles, uploaded when they change -- a loaded param group or a manual lr drop reaches
        the replays of a graph captured earlier), and advances the host's mirror.  Call before replaying -- or capturing -- a graph
        that contains ``step_staged``."""
        step_dev, _ = self._device_clock()
        if self._step_dev_expected != self.step_count:
            step_dev.fill_(int(self.step_count))
        consts = self._schedule_constants()
        if self._sched_uploaded != consts:
            if not (consts[0] > 0.0 and 0.0 <= consts[3] < 1.0 and 0.0 <= consts[4] < 1.0):
                raise ValueError(f"FlatAdam: need lr > 0 and betas in [0, 1), got lr {consts[0]}, betas {consts[3:5]}")
            # (stream-ordered like the fill above: replays queued earlier still read the old constants)
            self._sched_dev.copy_(torch.tensor(consts, dtype=torch.float64), non_blocking=False)
            self._sched_uploaded = consts
        self.step_count += 1
        self._step_dev_expected = self.step_count

    def step_staged(self, grad_scale: float = 1.0) -> None:
        """Device clock tick + the Adam update with the scalars it derived (lse_adam_schedule_dev, lse_adam_step_dev): capturable.
        Nothing step- or schedule-dependent is a launch argument: a captured pair follows ``prepare_step``'s device-side state."""
        step_dev, hyper = self._device_clock()
        ops.adam_schedule_dev(step_dev, hyper, self._sched_dev)
        ops.adam_step_dev(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, hyper, grad_scale)

    # -- resume (nerfstudio saves ``optimizers: {"fields": optimizer.state_dict()}``, R:lse_nerf/lse_trainer.py:85-122) ------
    def state_dict(self) -> dict:
        """torch.optim.Adam-style state dict (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` + one param group), so
        a checkpoint written from here loads into ``torch.optim.Adam`` over the same parameter list and vice versa."""
        state = {}
        for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone()}
        group = {"lr": self.current_lr(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.flat.params))),
                 "initial_lr": self.lr_init, "lr_final": self.lr_final, "max_steps": self.max_steps}
        # (the marker tells checkpoint.load_nerfstudio_checkpoint that the indices are positions in THIS parameter list)
        return {"state": state, "param_groups": [group], "lsenerf_amd_layout": True}

    def load_state_dict(self, sd: dict, index_map: Optional[dict] = None, names: Optional[dict] = None) -> None:
        """Load a torch.optim.Adam-style state dict.

        Without ``index_map`` the state indices are positions in THIS optimizer's parameter list -- i.e. a state dict written by
        ``state_dict()`` above (or by ``torch.optim.Adam`` over the same list).  A state dict written by the REFERENCE indexes
        its own parameter list, which contains parameters this model does not have (the dead torch-layout hash table,
        R:lse_nerf/lse_field.py:63-65) and orders the rest by its own module registration order; pass
        ``index_map = {reference index: local index or None}`` (``checkpoint.reference_optimizer_index_map`` builds it from
        the checkpoint's pipeline keys; ``load_nerfstudio_checkpoint`` does so) and ``names`` (reference index -> key, for the
        error messages).  A state entry that maps to no local parameter and is not known to be dead, or whose shape differs
        from the parameter it maps to, raises: moments are never matched by position alone across the two layouts."""
        state = sd["state"]
        n_local = len(self.flat.params)
        names = names or {}
        if index_map is None:
            if len(state) not in (0, n_local) or (state and {int(k) for k in state} != set(range(n_local))):
                raise ValueError(f"optimizer state has {len(state)} entries for {n_local} parameters: not a state dict of this "
                                 "parameter list (for a reference-written checkpoint pass index_map, see the docstring)")
            index_map = {i: i for i in range(n_local)}
        steps, seen = set(), set()
        for key, st in state.items():
            ri = int(key)
            if ri not in index_map:
                raise ValueError(f"optimizer state index {ri} ({names.get(ri, '?')}) is not covered by the index map")
            li = index_map[ri]
            if li is None:      # a parameter this model does not hold (dead table / zero-sized tcnn parameter)
                continue
            if li in seen:
                raise ValueError(f"two optimizer states map to local parameter {li} ({names.get(ri, '?')})")
            seen.add(li)
            p, o = self.flat.params[li], self.flat.offsets[li]
            if tuple(st["exp_avg"].shape) != tuple(p.shape) and st["exp_avg"].numel() != p.numel():
                raise ValueError(f"optimizer state {ri} ({names.get(ri, '?')}): shape {tuple(st['exp_avg'].shape)} does not fit "
                                 f"local parameter {li} of shape {tuple(p.shape)}")
            n = p.numel()
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"parameters disagree on the step count: {sorted(steps)}")
        if steps:
            self.step_count = steps.pop()
        g = sd["param_groups"][0]
        self.lr_init = g.get("initial_lr", self.lr_init)
        self.lr_final = g.get("lr_final", self.lr_final)
        self.max_steps = g.get("max_steps", self.max_steps)
