"""CPU, world_size 2 over gloo: the data-parallel plumbing (one flat-bucket gradient all-reduce,
1/world folded into the optimizer, buffer broadcast, shard arithmetic).  The same code runs over
RCCL ("nccl") on the GPUs; only the backend string differs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel(torch.nn.Module):
    """Stands in for ModelB_2 on CPU: exposes flat_grad() like the real module."""

    def __init__(self, n):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(n))
        self.register_buffer("running", torch.zeros(4))
        self._g = None

    def flat_grad(self):
        return self._g


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sifsr
    from sifsr import distributed as dp
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dp.world_size() == world
    n = 282705
    m = _FakeModel(n)
    g = torch.full((n,), float(rank + 1))
    m._g = g

    class Opt:
        grad_scale = 1.0
    opt = Opt()
    dp.allreduce_gradients(m, opt)
    ok = bool(torch.all(g == sum(range(1, world + 1)))) and opt.grad_scale == 1.0 / world
    g2 = torch.full((n,), float(rank + 1)); m._g = g2
    dp.allreduce_gradients(m, None)                  # no FlatAdam: mean applied here
    ok = ok and bool(torch.allclose(g2, torch.full((n,), sum(range(1, world + 1)) / world)))
    m.running.fill_(float(rank))
    dp.broadcast_buffers(m, src=0)
    ok = ok and bool(torch.all(m.running == 0))
    lo, hi = dp.shard_range(324, rank, world)
    t = torch.tensor([hi - lo]); dist.all_reduce(t)
    ok = ok and int(t) == 324
    f = torch.ones(8) * (rank + 1)
    dp.allreduce_flat_(f, average=True)
    ok = ok and bool(torch.allclose(f, torch.full((8,), (world + 1) / 2)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_world_size_1_is_noop():
    from sifsr import distributed as dp
    m = _FakeModel(8); m._g = torch.ones(8)
    dp.allreduce_gradients(m, None)
    assert torch.all(m._g == 1) and dp.world_size() == 1
