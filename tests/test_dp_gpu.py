"""Data-parallel step on the GPU (SURVEY.md §8 e): two ranks (gloo rehearsal on ONE device -- RCCL needs one GPU
per rank, the 8-GPU run is the driver's) must end up with identical parameters, equal to a single-process run
that processes the same two micro-batches with per-micro-batch BatchNorm statistics and averaged gradients.

Covered: SR2 and SR1 losses; replicas that were seeded DIFFERENTLY and are made identical by
``dp.broadcast_parameters``; the ``zero_grad(set_to_none=False)`` path in which ``p.grad`` does not alias the
backward's flat buffer (the all-reduce must then act on ``p.grad``); ``dp.broadcast_buffers`` before a save."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["SIFSR_ROOT"])
import sifsr
from sifsr import distributed as dp
kind, keep_grads = os.environ["SIFSR_KIND"], os.environ["SIFSR_KEEP_GRADS"] == "1"
alpha, gamma = (0.5, -0.25) if kind == "sr2" else (0.99, -0.5)
rank, world, local = dp.init_from_env()
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
torch.manual_seed(rank)                       # replicas start DIFFERENT; rank 0 is made authoritative below
model = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
opt = sifsr.FlatAdam(model.parameters(), lr=1e-3)
dp.broadcast_parameters(model, opt, src=0)
stats = dict(sifsr.dataset.DEFAULT_STATS)
lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(4, dev, seed=77)
lo, hi = dp.shard_range(4, rank, world)
a, b, c = lst[lo:hi].contiguous(), lst_up[lo:hi].contiguous(), ndvi[lo:hi].contiguous()
aliased = []
for it in range(2):
    if keep_grads:                            # the reference's own call: optimizer.zero_grad() with tensors kept
        model.train()
        opt.zero_grad(set_to_none=False)
        sr = model(torch.cat((b, c), dim=1))
        _, _, loss = sifsr.sif_loss(kind, sr, a, c, stats["mean_lst"], stats["std_lst"], alpha, gamma)
        loss.backward()
        aliased.append(dp.grads_alias_flat(model, model.flat_grad()))
        dp.allreduce_gradients(model, opt)
        opt.step()
    else:
        sifsr.train.train_step(model, opt, a, b, c, stats, alpha, gamma, kind)
        aliased.append(dp.grads_alias_flat(model, model.flat_grad()))
torch.cuda.synchronize()
flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
run_before = model.inbloc.bloc[1].running_mean.detach().cpu().clone()
dp.broadcast_buffers(model, src=0)
run_after = model.inbloc.bloc[1].running_mean.detach().cpu().clone()
torch.save({"flat": flat, "aliased": aliased, "run_before": run_before, "run_after": run_after},
           os.path.join(os.environ["SIFSR_OUT"], f"rank{rank}.pt"))
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
'''


@pytest.mark.parametrize("kind,keep_grads", [("sr2", False), ("sr1", False), ("sr2", True)])
def test_two_rank_step_matches_micro_batched_single_process(tmp_path, kind, keep_grads):
    import sifsr
    from sifsr import distributed as dp
    port = 29533 + (0 if kind == "sr2" else 1) + (2 if keep_grads else 0)
    env = dict(os.environ, SIFSR_ROOT=ROOT, SIFSR_OUT=str(tmp_path), SIFSR_DIST_BACKEND="gloo", SIFSR_KIND=kind,
               SIFSR_KEEP_GRADS="1" if keep_grads else "0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e))
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    a, b = r0["flat"], r1["flat"]
    assert torch.equal(a, b), "ranks diverged"
    # first step: p.grad was None -> autograd adopts the flat views (fast path); with set_to_none=False the second
    # backward ACCUMULATES into the kept tensors, which no longer alias the new flat buffer (gather path)
    assert r0["aliased"] == ([True, False] if keep_grads else [True, True])
    # BN running statistics differ per rank (different shards) until broadcast_buffers makes rank 0's authoritative
    assert not torch.equal(r0["run_before"], r1["run_before"])
    assert torch.equal(r0["run_after"], r1["run_after"]) and torch.equal(r0["run_after"], r0["run_before"])

    # single process: same two micro-batches, per-micro-batch BN statistics, averaged gradients, one Adam step
    alpha, gamma = (0.5, -0.25) if kind == "sr2" else (0.99, -0.5)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)                      # rank 0's seed: broadcast_parameters made it everybody's
    model = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).to(dev)
    opt = sifsr.FlatAdam(model.parameters(), lr=1e-3)
    stats = dict(sifsr.dataset.DEFAULT_STATS)
    lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(4, dev, seed=77)
    for _ in range(2):
        acc = None
        for r in range(2):
            lo, hi = dp.shard_range(4, r, 2)
            model.train()
            opt.zero_grad(set_to_none=True)
            sr = model(torch.cat((lst_up[lo:hi], ndvi[lo:hi]), dim=1))
            _, _, loss = sifsr.sif_loss(kind, sr, lst[lo:hi].contiguous(), ndvi[lo:hi].contiguous(), stats["mean_lst"], stats["std_lst"], alpha, gamma)
            loss.backward()
            g = model.flat_grad().clone()
            acc = g if acc is None else acc + g
        model.flat_grad().copy_(acc)
        opt.grad_scale = 0.5
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    # the parameters must agree: the all-reduce adds the same two fp32 gradient buffers this loop adds, and every kernel
    # is deterministic.  (keep_grads: zero_grad(set_to_none=False) zeroes and autograd adds -> same values)
    assert torch.allclose(a, ref, rtol=0, atol=5e-6)


RCCL_WORKER = r'''
import os, sys, torch
sys.path.insert(0, os.environ["SIFSR_ROOT"])
import torch.distributed as dist
import sifsr
from sifsr import distributed as dp
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(3)
model = sifsr.ModelB_2(2, [16, 32, 64, 128], "replicate", "ReLU", 1, 1).cuda()
opt = sifsr.FlatAdam(model.parameters(), lr=1e-3)
dp.broadcast_parameters(model, opt, src=0)                     # ncclBroadcast of the flat buffers
stats = dict(sifsr.dataset.DEFAULT_STATS)
lst, lst_up, ndvi = sifsr.dataset.synthetic_device_batch(2, "cuda", seed=5)
model.train(); opt.zero_grad(set_to_none=True)
sr = model(torch.cat((lst_up, ndvi), 1))
_, _, loss = sifsr.sif_loss("sr2", sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], 0.5, -0.25)
loss.backward()
g0 = model.flat_grad().clone()
flat = model.flat_grad()
dp._sum_all_reduce_(flat)                                      # ncclAllReduce(sum) over the 1.13 MB gradient bucket
torch.cuda.synchronize()
assert dist.get_backend() == "nccl" and torch.equal(flat, g0) and flat.numel() == 282705
dp.broadcast_buffers(model, src=0)
opt.step(); torch.cuda.synchronize()
print("rccl single-rank ok", float(loss))
dist.barrier(); dist.destroy_process_group()
'''


def test_rccl_backend_single_rank(tmp_path):
    """RCCL needs one GPU per rank, so the multi-rank exchange cannot run on a one-GPU box; what can is the backend
    itself: a world of ONE rank on backend "nccl" (= RCCL on ROCm) creates the communicator and runs the very calls of
    the data-parallel step -- ncclBroadcast of the flat parameter / optimizer buffers, ncclAllReduce(sum) of the flat
    282,705-float gradient, the buffer broadcast -- on device memory, and must leave the values unchanged."""
    env = dict(os.environ, SIFSR_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rccl single-rank ok" in r.stdout


def test_bench_gpus_2_as_the_driver_starts_it():
    """`python bench.py --gpus 2 --steps 3` with NO launcher environment -- the driver's command form -- must start its own
    two rank processes (a child torch.distributed.run, before the parent touches HIP), run the data-parallel step and let
    rank 0 print the one JSON line with the N > 1 fields.  Two ranks share this box's single GPU, so the collective backend is
    the gloo rehearsal (RCCL needs one GPU per rank); everything else is the N = 8 code path."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SIFSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["value"] > 0
    assert j["dist_backend"] == "gloo" and j["rccl_ranks"] == 0            # "nccl" / 2 on a node with one GPU per rank
    assert j["allreduce_ms"] > 0 and j["rank_ms_per_step_max"] >= j["rank_ms_per_step_min"] > 0
    assert abs(j["value"] - 2 * 8 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-2 * j["value"]   # whole-job patches / max-rank time
    assert "cpu_baseline" not in j and "also" not in j                      # N = 1 only
