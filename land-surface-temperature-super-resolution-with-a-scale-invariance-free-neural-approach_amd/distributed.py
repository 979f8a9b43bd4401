"""Pure data parallelism over the GPUs of one node (SURVEY.md §8 e): one process per GPU
(``torchrun``), patch minibatches sharded over ranks, ONE sum all-reduce of the flat 282,705-float
gradient buffer per step over RCCL/xGMI (backend "nccl" on ROCm), 1/world folded into the Adam
kernel.  BatchNorm statistics stay per replica (DistributedDataParallel's default semantics).
The reference has no distributed code at all; this module is the whole of it.  ``gloo`` works for
CPU tests of the plumbing (world_size 2).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC handles for RCCL on this driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("SIFSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()   # rehearsals with more ranks than GPUs (gloo); identity on a full node
    return rank, world, local


def _sum_all_reduce_(flat: torch.Tensor):
    """Sum all-reduce in place.  RCCL reduces device memory directly; the gloo rehearsal path stages device
    tensors through the host (gloo in this build has no device support)."""
    if flat.is_cuda and dist.get_backend() == "gloo":
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard of ``n_items`` units (patches / tiles) for ``rank``; covers everything once."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_flat_(flat: torch.Tensor, average: bool = False):
    """In-place sum (or mean) all-reduce of one flat tensor; a no-op for world_size 1."""
    if world_size() > 1:
        _sum_all_reduce_(flat)
        if average:
            flat.div_(world_size())
    return flat


def allreduce_gradients(model, optimizer=None):
    """Sum-all-reduce the model's flat gradient buffer (ONE collective).  With a ``FlatAdam``
    optimizer the 1/world_size is applied inside the Adam kernel (``grad_scale``); otherwise the
    gradients are divided here."""
    w = world_size()
    if w == 1:
        return
    flat = model.flat_grad() if hasattr(model, "flat_grad") else None
    if flat is None:
        raise RuntimeError("no flat gradient: run backward first")
    _sum_all_reduce_(flat)
    if optimizer is not None and hasattr(optimizer, "grad_scale"):
        optimizer.grad_scale = 1.0 / w
    else:
        flat.div_(w)


def broadcast_buffers(model, src: int = 0):
    """BN running statistics differ slightly per replica; make rank ``src``'s authoritative before a
    save (DDP ``broadcast_buffers`` semantics)."""
    if world_size() == 1:
        return
    for b in model.buffers():
        if b.is_cuda and dist.get_backend() == "gloo":
            host = b.detach().cpu()
            dist.broadcast(host, src=src)
            b.copy_(host)
        else:
            dist.broadcast(b, src=src)
