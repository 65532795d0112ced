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
