"""world_size-2 gloo tests of the data-parallel exchange (lsenerf_amd.dist): ray sharding + one all-reduce of the flat
gradient buffer + averaged Adam == single-process training on the full batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _adam_cpu(flat, state, lr, scale, step):
    g = flat.grad * scale
    state["m"].mul_(0.9).add_(g, alpha=0.1)
    state["v"].mul_(0.999).addcmul_(g, g, value=0.001)
    bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
    flat.data.addcdiv_(state["m"], (state["v"].sqrt() / bc2 ** 0.5).add_(1e-15), value=-lr / bc1)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.optim import FlatParams
    r, w, _ = ldist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    if rank == 1:   # de-synchronise on purpose: broadcast_params must repair it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    flat = FlatParams(model.parameters())
    ldist.broadcast_params(flat.data)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
    sl = ldist.shard_rays(64, rank, world)
    state = {"m": torch.zeros_like(flat.data), "v": torch.zeros_like(flat.data)}
    for step in range(1, 4):
        flat.zero_grad()
        loss = ((model(x[sl]) - y[sl]) ** 2).mean()
        loss.backward()
        ldist.allreduce_grads(flat.grad)
        _adam_cpu(flat, state, 1e-2, 1.0 / world, step)
    mx = ldist.max_over_ranks(float(rank), "cpu")
    if rank == 0:
        torch.save({"params": flat.data.clone(), "max": mx}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_equals_full_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from lsenerf_amd.optim import FlatParams
    model = _model()
    flat = FlatParams(model.parameters())
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
    state = {"m": torch.zeros_like(flat.data), "v": torch.zeros_like(flat.data)}
    for step in range(1, 4):
        flat.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        _adam_cpu(flat, state, 1e-2, 1.0, step)
    assert torch.allclose(got["params"], flat.data, atol=1e-6)
    assert got["max"] == 1.0


def test_shard_rays_partition():
    from lsenerf_amd.dist import shard_rays
    parts = [shard_rays(32768, r, 8) for r in range(8)]
    assert parts[0] == slice(0, 4096) and parts[7] == slice(28672, 32768)
    assert sum(p.stop - p.start for p in parts) == 32768


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lsenerf_amd import dist as ldist, ops
    from lsenerf_amd.optim import FlatParams
    ldist.init_from_env("gloo")
    torch.manual_seed(0)
    other = torch.nn.Parameter(torch.randn(100))
    table = torch.nn.Parameter(torch.randn(2 * 96))         # 3 "levels" of 32 entries x 2 features
    tail = torch.nn.Parameter(torch.randn(7))
    flat = FlatParams([other, table, tail])
    ex = ldist.OverlappedGradExchange(flat, table, level_offsets=(0, 32, 64, 96), split_level=1)
    ex.install()
    assert ops.get_hash_bwd_hook(table, "split")[0] == 1
    assert ops.get_hash_bwd_hook(other, "split") is None          # keyed by the table: nothing else in the process is hooked
    res = []
    for use_split in (True, False):
        flat.zero_grad()
        g = torch.Generator().manual_seed(10 + rank)
        # what the two hash-backward launches do: levels >= split first, callback, then the rest; other grads arrive around it
        other.grad.add_(torch.randn(100, generator=g))
        table.grad[64:].add_(torch.randn(128, generator=g))
        if use_split:
            ops.get_hash_bwd_hook(table, "split")[1]()
        table.grad[:64].add_(torch.randn(64, generator=g))
        tail.grad.add_(torch.randn(7, generator=g))
        ex.finish()           # without the callback: falls back to one plain all-reduce
        res.append(flat.grad.clone())
    ex.uninstall()
    assert ops.get_hash_bwd_hook(table, "split") is None and not ops._HASH_BWD_HOOKS
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_equals_plain_allreduce(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    split, plain = torch.load(out)
    assert torch.equal(split, plain)
    # and both are the sum over the two ranks' gradients
    exp = torch.zeros_like(plain)
    offs = (0, 128, 128 + 192)     # FlatParams aligns every parameter to 64 floats
    for rank in range(2):
        g = torch.Generator().manual_seed(10 + rank)
        exp[offs[0]:offs[0] + 100] += torch.randn(100, generator=g)
        hi = torch.randn(128, generator=g)
        lo = torch.randn(64, generator=g)
        exp[offs[1]:offs[1] + 64] += lo
        exp[offs[1] + 64:offs[1] + 192] += hi
        exp[offs[2]:offs[2] + 7] += torch.randn(7, generator=g)
    assert torch.allclose(plain, exp)


def _adam_fn_cpu(p, g, m, v, lr, b1, b2, eps, step, scale):
    g = g * scale
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.addcdiv_(m, (v.sqrt() / (1 - b2 ** step) ** 0.5).add_(eps), value=-lr / (1 - b1 ** step))


def _sharded_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.optim import FlatParams
    ldist.init_from_env("gloo")
    model = _model()
    flat = FlatParams(model.parameters())
    ldist.broadcast_params(flat.data)
    ex = ldist.ShardedAdamExchange(flat, lr=1e-2, adam_fn=_adam_fn_cpu)
    assert ex.per % 64 == 0 and ex.padded >= flat.data.numel() and ex.exp_avg.numel() == ex.per
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
    sl = ldist.shard_rays(64, rank, world)
    for _ in range(3):
        flat.zero_grad()
        ((model(x[sl]) - y[sl]) ** 2).mean().backward()
        ex.step()
    torch.save(flat.data.clone(), out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adam_equals_full_batch_and_replicates(tmp_path):
    out = str(tmp_path / "p")
    mp.spawn(_sharded_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    p0, p1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(p0, p1)                        # identical parameters on every rank
    from lsenerf_amd.optim import FlatParams
    model = _model()
    flat = FlatParams(model.parameters())
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
    state = {"m": torch.zeros_like(flat.data), "v": torch.zeros_like(flat.data)}
    for step in range(1, 4):
        flat.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        _adam_cpu(flat, state, 1e-2, 1.0, step)
    assert torch.allclose(p0, flat.data, atol=1e-6)


def _multi_backward_worker(rank, world, port, out):
    """Event configs: three field passes (colour / previous / next bundle) -> three hash backwards per loss.backward(),
    all accumulating into the same table gradient.  The early all-reduce may only start after the LAST one."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lsenerf_amd import dist as ldist, ops
    from lsenerf_amd.optim import FlatParams
    ldist.init_from_env("gloo")
    torch.manual_seed(0)
    other = torch.nn.Parameter(torch.randn(100))
    table = torch.nn.Parameter(torch.randn(2 * 96))
    flat = FlatParams([other, table])
    ex = ldist.OverlappedGradExchange(flat, table, level_offsets=(0, 32, 64, 96), split_level=1)
    ex.install()
    res = []
    for mode in ("armed", "plain"):
        flat.zero_grad()
        g = torch.Generator().manual_seed(20 + rank)
        if mode == "armed":
            ex.begin_step(3)
        for _ in range(3):                          # three accumulating hash backwards: fine levels, callback, coarse levels
            table.grad[64:].add_(torch.randn(128, generator=g))
            if mode == "armed":
                ops.get_hash_bwd_hook(table, "split")[1]()
            table.grad[:64].add_(torch.randn(64, generator=g))
        other.grad.add_(torch.randn(100, generator=g))
        if mode == "armed":
            ex.finish()
        else:
            ldist.allreduce_grads(flat.grad)
        res.append(flat.grad.clone())
    # an un-armed second callback in one step must raise instead of reducing the slice twice
    ex.begin_step(1)
    ops.get_hash_bwd_hook(table, "split")[1]()
    try:
        ops.get_hash_bwd_hook(table, "split")[1]()
        raised = False
    except RuntimeError:
        raised = True
    ex.finish()
    ex.uninstall()
    if rank == 0:
        torch.save({"res": res, "raised": raised}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_exchange_with_three_hash_backwards_per_step(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_multi_backward_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    armed, plain = got["res"]
    assert torch.equal(armed, plain)
    assert got["raised"]


class _ToyOpt:
    """FlatAdam's interface on CPU tensors (the HIP Adam kernel needs a GPU)."""

    lr = 1e-2

    def __init__(self, flat):
        self.flat, self.step_count = flat, 0
        self.state = {"m": torch.zeros_like(flat.data), "v": torch.zeros_like(flat.data)}

    def current_lr(self):
        return self.lr

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        self.step_count += 1
        _adam_cpu(self.flat, self.state, 1e-2, grad_scale, self.step_count)


class _ToyEstimator:
    """Only what GradPipeline touches: the hook slot and a `sampling` that fires it between marcher and sigma_fn."""

    def __init__(self):
        self.after_march_hook = None

    def sampling(self, sigma_fn):
        if self.after_march_hook is not None:
            self.after_march_hook()
        return sigma_fn()


def _pipeline_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.optim import FlatParams
    ldist.init_from_env("gloo")
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
    sl = ldist.shard_rays(64, rank, world)
    finals, seen = [], []
    for mode in ("pipelined", "plain", "pipelined_sharded"):
        model = _model()
        flat = FlatParams(model.parameters(), total_multiple=world * 64 if mode == "pipelined_sharded" else 1)
        ldist.broadcast_params(flat.data)
        opt = _ToyOpt(flat)
        est = _ToyEstimator()
        sharded = ldist.ShardedAdamExchange(flat, lr=opt.lr, adam_fn=_adam_fn_cpu) if mode == "pipelined_sharded" else None
        pipe = ldist.GradPipeline(opt, world, sharded=sharded).attach(est) if mode != "plain" else None
        if pipe is not None:
            pipe.exposed_events = []      # the measurement hook of bench.py: no device here, so nothing is recorded and nothing breaks
        for step in range(4):
            # the visibility pre-pass reads the parameters: it must see the update of the previous step
            seen.append(est.sampling(lambda: float(flat.data.sum())))
            opt.zero_grad()
            ((model(x[sl]) - y[sl]) ** 2).mean().backward()
            if pipe is not None:
                pipe.start()
            else:
                ldist.allreduce_grads(flat.grad)
                opt.step(grad_scale=1.0 / world)
        if pipe is not None:
            pipe.flush()
            pipe.flush()      # idempotent
            assert opt.step_count == 4 and pipe.exposed_events == []
        finals.append(flat.data.clone())
    if rank == 0:
        torch.save({"finals": finals, "seen": seen}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_pipeline_is_value_identical_to_the_blocking_exchange(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert torch.equal(got["finals"][0], got["finals"][1])
    assert got["seen"][:4] == got["seen"][4:8]         # sigma_fn saw the same parameters in both schedules
    # GradPipeline over ShardedAdamExchange (reduce-scatter started behind backward, Adam on 1/W + all-gather behind the next
    # step's marcher): the same parameters up to the summation order of the sharded reduction
    n = got["finals"][1].numel()
    assert torch.allclose(got["finals"][2][:n], got["finals"][1], rtol=0, atol=1e-6)
    assert all(abs(a - b) < 1e-4 for a, b in zip(got["seen"][8:], got["seen"][4:8]))


def _sharded_inplace_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.optim import FlatParams
    ldist.init_from_env("gloo")
    res = []
    for mult in (1, world * 64):
        model = _model()
        flat = FlatParams(model.parameters(), total_multiple=mult)
        ex = ldist.ShardedAdamExchange(flat, lr=1e-2, adam_fn=_adam_fn_cpu)
        if mult > 1:
            assert ex._pad == 0 and ex.grad_full is None and ex.param_full is None      # collectives run in place
        g = torch.Generator().manual_seed(1)
        x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
        sl = ldist.shard_rays(64, rank, world)
        for _ in range(3):
            flat.zero_grad()
            ((model(x[sl]) - y[sl]) ** 2).mean().backward()
            ex.step()
        res.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adam_in_place_buffers_match_the_staged_ones(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_sharded_inplace_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    staged, inplace = torch.load(out)
    assert torch.equal(staged, inplace)


def _grid_rng_worker(rank, world, port, out):
    """R:train.py:104 seeds every rank differently; the occupancy refresh must not care."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.grid_estimator import LSEOccGridEstimator
    ldist.init_from_env("gloo")
    torch.manual_seed(1234 + rank)                        # per-rank global RNG, as in the reference
    est = LSEOccGridEstimator([-1.0, -1, -1, 1, 1, 1], resolution=16, levels=2)
    est.occs.copy_(torch.rand(est.occs.shape, generator=torch.Generator().manual_seed(5)))
    est.binaries.copy_((est.occs > 0.5).view(est.binaries.shape))
    torch.rand(rank + 1)                                  # ranks have consumed different amounts of global randomness
    same = []
    for step in (0, 320):                                 # warm-up branch (all cells) and the sampled branch
        cells = est._update_samples(step, 256, est._update_generator(step))
        blob = torch.cat([torch.cat([i.float(), x.reshape(-1)]) for i, x in cells])
        n = torch.tensor([blob.numel()])
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        ok = all(int(v) == int(n) for v in ns)
        if ok:
            ref = blob.clone()
            dist.broadcast(ref, src=0)
            ok = torch.equal(ref, blob)
        same.append(ok)
    consistent = ldist.check_grid_consistency(est)
    est.occs[0] += float(rank)                            # and the checker notices a divergence
    est.binaries.view(-1)[3] = bool(rank)
    broken = ldist.check_grid_consistency(est)
    ldist.sync_grid(est)                                  # ... which the broadcast of rank 0's grid repairs
    repaired = ldist.check_grid_consistency(est)
    # attach_grid_sync: the hook runs at the end of every refresh
    ldist.attach_grid_sync(est)
    est.occs[5] -= float(rank)                            # (the refresh itself runs HIP kernels: the hook is called directly here)
    est.after_update_hook()
    hooked = ldist.check_grid_consistency(est)
    if rank == 1:
        torch.save({"same": same, "consistent": consistent, "broken": broken, "repaired": repaired, "hooked": hooked}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_occupancy_refresh_is_rank_independent(tmp_path):
    out = str(tmp_path / "r1.pt")
    mp.spawn(_grid_rng_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["same"] == [True, True]
    assert got["consistent"] is True and got["broken"] is False
    assert got["repaired"] is True and got["hooked"] is True


def test_shard_rays_rejects_uneven_splits():
    from lsenerf_amd.dist import shard_rays
    with pytest.raises(AssertionError):
        shard_rays(4097, 0, 2)


def _single_rank_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.optim import FlatParams
    calls = []
    for name in ("all_reduce", "broadcast", "all_gather"):
        orig = getattr(dist, name)
        setattr(dist, name, (lambda o, n: lambda *a, **k: (calls.append(n), o(*a, **k))[1])(orig, name))
    ldist.SINGLE_RANK_COLLECTIVES = True
    assert ldist.init_from_env("gloo") == (0, 1, 0) and dist.is_initialized()      # a one-rank group IS created when forced
    model = _model()
    flat = FlatParams(model.parameters())
    ldist.broadcast_params(flat.data)
    ex = ldist.ShardedAdamExchange(flat, lr=1e-2, adam_fn=_adam_fn_cpu)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
    for _ in range(3):
        flat.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        ldist.allreduce_grads(flat.grad.clone())
        ex.step()
    assert ldist.max_over_ranks(2.5, torch.device("cpu")) == 2.5
    torch.save({"p": flat.data.clone(), "calls": calls}, out)
    dist.destroy_process_group()


def test_single_rank_collectives_switch_runs_every_collective_and_changes_nothing(tmp_path):
    """dist.SINGLE_RANK_COLLECTIVES (how tests/test_gpu_dp.py executes the RCCL calls on a one-GPU box): with one rank every
    collective is the identity, so the sharded exchange must give exactly the parameters of the loop without torch.distributed."""
    out = str(tmp_path / "p")
    mp.spawn(_single_rank_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    res = torch.load(out)
    assert {"all_reduce", "broadcast", "all_gather"} <= set(res["calls"])
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.optim import FlatParams
    assert ldist.SINGLE_RANK_COLLECTIVES is False and not ldist._active()              # off by default: no group, no collectives
    model = _model()
    flat = FlatParams(model.parameters())
    ex = ldist.ShardedAdamExchange(flat, lr=1e-2, adam_fn=_adam_fn_cpu)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(64, 6, generator=g), torch.randn(64, 3, generator=g)
    for _ in range(3):
        flat.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        ex.step()
    assert torch.equal(res["p"], flat.data)


# ---- config 5's exact split on CPU: 8 ranks x 4096 of 32768 rays (R:lse_nerf/lse_pipeline.py:95-98) ------------------
class _TinyField(torch.nn.Module):
    """A field-shaped toy: a 'table' gathered by position (odd size), two small MLP matrices and a 200-float tail -- 1565 parameters
    in a 2048-float buffer (each parameter 64-aligned, the total a multiple of 8 x 64), which the 8 shards of 256 floats cut in the
    MIDDLE of the table, of a weight matrix and of the tail; the last rank's shard is padding only."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.table = torch.nn.Parameter(torch.randn(1101, generator=g) * 0.1)
        self.w0 = torch.nn.Parameter(torch.randn(24, 8, generator=g) * 0.3)
        self.w1 = torch.nn.Parameter(torch.randn(3, 24, generator=g) * 0.3)
        self.tail = torch.nn.Parameter(torch.randn(200, generator=g) * 0.1)

    def forward(self, o, d):
        idx = ((o.abs().sum(-1) * 997.0).long() % 1101)
        feat = torch.cat([o, d, self.table[idx][:, None], self.tail[idx % 200][:, None]], dim=-1)       # [R, 8]
        return torch.sigmoid(torch.relu(feat @ self.w0.t()) @ self.w1.t())


def _world8_rays():
    g = torch.Generator().manual_seed(96)
    o, d = torch.randn(32768, 3, generator=g), torch.randn(32768, 3, generator=g)
    return o, d / d.norm(dim=-1, keepdim=True), torch.rand(32768, 3, generator=g)


def _world8_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from lsenerf_amd import dist as ldist
    from lsenerf_amd.optim import FlatParams
    r, w, _ = ldist.init_from_env("gloo")
    assert (r, w) == (rank, 8)
    o, d, y = _world8_rays()
    sl = ldist.shard_rays(32768, rank, world)
    assert sl == slice(4096 * rank, 4096 * (rank + 1))
    finals = {}
    for mode in ("plain", "pipelined_sharded", "sharded"):
        model = _TinyField()
        if rank:        # de-synchronise on purpose: broadcast_params must repair it
            with torch.no_grad():
                model.tail.add_(float(rank))
        flat = FlatParams(model.parameters(), total_multiple=world * 64)
        assert flat.data.numel() == 2048 and sum(p.numel() for p in model.parameters()) == 1565
        ldist.broadcast_params(flat.data)
        opt = _ToyOpt(flat)
        est = _ToyEstimator()
        sharded = ldist.ShardedAdamExchange(flat, lr=opt.lr, adam_fn=_adam_fn_cpu) if mode != "plain" else None
        if sharded is not None:
            # in place on the flat buffers (no staging copies), and every shard border lies strictly INSIDE a parameter: table | w0 |
            # w1 | tail start at 0, 1152, 1344, 1472 and end at 1672; the borders are the multiples of 256, the last of them (1792)
            # lies in the padding: rank 7 reduces, updates and gathers a shard that holds no parameter at all
            assert sharded.per == 256 and sharded.padded == 2048 and sharded.param_full is None and sharded.grad_full is None
            assert flat.offsets == [0, 1152, 1344, 1472]
            for b in range(256, 1672, 256):
                assert any(o_ < b < o_ + p_.numel() for o_, p_ in zip(flat.offsets, flat.params)), b
        pipe = ldist.GradPipeline(opt, world, sharded=sharded).attach(est) if mode == "pipelined_sharded" else None
        for step in range(3):
            est.sampling(lambda: None)                  # (the marcher of this step: the pipeline finishes the previous step here)
            opt.zero_grad()
            ((model(o[sl], d[sl]) - y[sl]) ** 2).mean().backward()
            if pipe is not None:
                pipe.start()
            elif sharded is not None:
                sharded.lr = opt.current_lr()
                opt.step_count += 1
                sharded.step()
            else:
                ldist.allreduce_grads(flat.grad)
                opt.step(grad_scale=1.0 / world)
        if pipe is not None:
            pipe.flush()
        assert opt.step_count == 3
        finals[mode] = flat.data.clone()
    torch.save(finals, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_world_8_split_of_config_5_on_cpu(tmp_path):
    """BASELINE config 5 as named -- 32768 rays over 8 ranks, one exchange of the flat gradient per step -- over gloo on the CPU (no
    8-GPU node has been available in any round): ``shard_rays(32768, r, 8)``, ``FlatParams(total_multiple=8 * 64)``,
    ``ShardedAdamExchange`` (blocking and behind ``GradPipeline``) and the plain all-reduce, on a field-shaped toy whose shard borders
    fall inside parameters and whose flat buffer has a padded tail.  Every rank ends with the same bits; all three exchanges equal
    ONE process on the full batch."""
    out = str(tmp_path / "p")
    mp.spawn(_world8_worker, args=(8, _free_port(), out), nprocs=8, join=True)
    got = [torch.load(out + f".{r}") for r in range(8)]
    for mode in ("plain", "pipelined_sharded", "sharded"):
        for r in range(1, 8):
            assert torch.equal(got[r][mode], got[0][mode]), (mode, r)        # replicas stay replicas
    from lsenerf_amd.optim import FlatParams
    model = _TinyField()
    flat = FlatParams(model.parameters(), total_multiple=8 * 64)
    o, d, y = _world8_rays()
    state = {"m": torch.zeros_like(flat.data), "v": torch.zeros_like(flat.data)}
    for step in range(1, 4):
        flat.zero_grad()
        ((model(o, d) - y) ** 2).mean().backward()
        _adam_cpu(flat, state, 1e-2, 1.0, step)
    for mode in ("plain", "pipelined_sharded", "sharded"):
        assert torch.allclose(got[0][mode], flat.data, rtol=0, atol=2e-6), (mode, float((got[0][mode] - flat.data).abs().max()))
    assert torch.equal(got[0]["pipelined_sharded"], got[0]["sharded"])      # the pipeline reorders launches, not arithmetic
    assert bool((got[0]["plain"][1672:] == 0).all())                         # the padded tail never moves
