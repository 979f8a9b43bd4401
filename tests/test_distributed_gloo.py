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
    m.w.grad = g.view_as(m.w)                        # what autograd does when p.grad was None: adopt the view

    class Opt:
        grad_scale = 1.0
    opt = Opt()
    assert dp.grads_alias_flat(m, g)
    dp.allreduce_gradients(m, opt)
    ok = bool(torch.all(g == sum(range(1, world + 1)))) and opt.grad_scale == 1.0 / world
    g2 = torch.full((n,), float(rank + 1)); m._g = g2; m.w.grad = g2.view_as(m.w)
    dp.allreduce_gradients(m, None)                  # no FlatAdam: mean applied here
    ok = ok and bool(torch.allclose(g2, torch.full((n,), sum(range(1, world + 1)) / world)))
    # p.grad does NOT alias the last backward's buffer (zero_grad(set_to_none=False) / accumulation): the optimizer
    # steps on p.grad, so p.grad is what must come back reduced -- reducing only flat_grad() would be silently wrong
    stale = torch.full((n,), 100.0 * (rank + 1)); m._g = stale
    m.w.grad = torch.full((n,), float(rank + 1))
    assert not dp.grads_alias_flat(m, stale)
    dp.allreduce_gradients(m, None)
    ok = ok and bool(torch.allclose(m.w.grad, torch.full((n,), sum(range(1, world + 1)) / world)))
    opt2 = Opt(); m.w.grad = torch.full((n,), float(rank + 1))
    dp.allreduce_gradients(m, opt2)
    ok = ok and bool(torch.all(m.w.grad == sum(range(1, world + 1)))) and opt2.grad_scale == 1.0 / world
    m.w.grad = None
    try:
        dp.allreduce_gradients(m, None); ok = False
    except RuntimeError:
        pass
    m.running.fill_(float(rank))
    dp.broadcast_buffers(m, src=0)
    ok = ok and bool(torch.all(m.running == 0))
    # replicas seeded differently (or a checkpoint loaded on one rank only) are made identical by broadcast_parameters,
    # optimizer moments and step count included
    torch.manual_seed(100 + rank)
    m2 = _FakeModel(64); m2.w.data.normal_(); m2.running.normal_()

    class FakeAdam:
        capturable = False
        def __init__(self):
            self._m, self._v, self._step, self._step_dev = torch.randn(64), torch.rand(64), 3 + rank, None
    o2 = FakeAdam()
    dp.broadcast_parameters(m2, o2, src=0)
    torch.manual_seed(100)
    exp = _FakeModel(64); exp.w.data.normal_(); exp.running.normal_(); exp_m, exp_v = torch.randn(64), torch.rand(64)
    ok = ok and torch.equal(m2.w.data, exp.w.data) and torch.equal(m2.running, exp.running)
    ok = ok and torch.equal(o2._m, exp_m) and torch.equal(o2._v, exp_v) and o2._step == 3
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
    m = _FakeModel(8); m._g = torch.ones(8); m.w.grad = m._g.view_as(m.w)
    dp.allreduce_gradients(m, None)
    assert torch.all(m._g == 1) and dp.world_size() == 1
