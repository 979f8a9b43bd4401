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


class AllreduceTimer:
    """Optional timing of the gradient exchange (bench.py, N > 1): a pool of event pairs created up front, one pair
    recorded on the caller's stream around every collective while installed (``set_allreduce_timer``).  The collective
    runs on RCCL's own stream, which waits for the caller's stream and is waited for by it, so the pair brackets the
    exchange *including* the wait for the slowest rank.  ``mean_ms()`` synchronises the recorded pairs."""

    def __init__(self, capacity: int = 4096):
        self._ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(capacity)]
        self._n = 0

    def reset(self):
        self._n = 0

    def bracket(self):
        if self._n >= len(self._ev):
            return None
        pair = self._ev[self._n]
        self._n += 1
        return pair

    def times_ms(self):
        out = []
        for a, b in self._ev[:self._n]:
            b.synchronize()
            out.append(a.elapsed_time(b))
        return out

    def mean_ms(self):
        t = self.times_ms()
        return sum(t) / len(t) if t else 0.0


_timer: AllreduceTimer | None = None


def set_allreduce_timer(timer: AllreduceTimer | None):
    """Install (or remove, with None) the timer the gradient all-reduce records into."""
    global _timer
    _timer = timer


def _sum_all_reduce_(flat: torch.Tensor):
    """Sum all-reduce in place.  RCCL reduces device memory directly; the gloo rehearsal path stages device
    tensors through the host (gloo in this build has no device support)."""
    pair = _timer.bracket() if (_timer is not None and flat.is_cuda) else None
    if pair is not None:
        pair[0].record()
    if flat.is_cuda and dist.get_backend() == "gloo":
        host = flat.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if pair is not None:
        pair[1].record()


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


def grads_alias_flat(model, flat: torch.Tensor) -> bool:
    """True when every ``p.grad`` IS the corresponding slice of ``flat`` (autograd adopted the views the backward
    returned: ``p.grad`` was None before it).  False after ``zero_grad(set_to_none=False)`` or gradient accumulation:
    ``p.grad`` then owns other memory, into which autograd *added* the new gradient."""
    off, esz = 0, flat.element_size()
    for p in model.parameters():
        if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + off * esz or p.grad.device != flat.device:
            return False
        off += p.numel()
    return off == flat.numel()


def allreduce_gradients(model, optimizer=None):
    """Sum-all-reduce the gradients the optimizer will step on, as ONE collective over a flat 282,705-float buffer.
    With a ``FlatAdam`` optimizer the 1/world_size is applied inside the Adam kernel (``grad_scale``); otherwise the
    gradients are divided here.

    The fast path reduces the buffer the last backward wrote (``model.flat_grad()``) in place -- valid only if every
    ``p.grad`` aliases it, which is checked.  Otherwise (accumulated gradients, ``set_to_none=False``) the ``p.grad``
    tensors themselves are gathered into a flat buffer, reduced, and scattered back, so that no optimizer can step
    on un-reduced gradients."""
    w = world_size()
    if w == 1:
        return
    flat = model.flat_grad() if hasattr(model, "flat_grad") else None
    params = list(model.parameters())
    if any(p.grad is None for p in params):
        raise RuntimeError("allreduce_gradients: a parameter has no gradient -- run backward first")
    if flat is not None and grads_alias_flat(model, flat):
        _sum_all_reduce_(flat)
        if optimizer is not None and hasattr(optimizer, "grad_scale"):
            optimizer.grad_scale = 1.0 / w
        else:
            flat.div_(w)
        return
    gathered = torch.cat([p.grad.reshape(-1) for p in params])
    _sum_all_reduce_(gathered)
    if optimizer is not None and hasattr(optimizer, "grad_scale"):
        optimizer.grad_scale = 1.0 / w
    else:
        gathered.div_(w)
    off = 0
    for p in params:
        p.grad.copy_(gathered[off:off + p.numel()].view_as(p.grad))
        off += p.numel()


def _broadcast_(t: torch.Tensor, src: int):
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.detach().cpu()
        dist.broadcast(host, src=src)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src)


@torch.no_grad()
def broadcast_parameters(model, optimizer=None, src: int = 0):
    """Make rank ``src`` authoritative for everything a replica steps on: the flat parameter buffer, the BatchNorm
    buffers and -- for ``FlatAdam`` -- the moment buffers and the step count (what DistributedDataParallel does for the
    module at construction, plus the optimizer state).  Call once after building model + optimizer (and after loading a
    checkpoint on any rank): replicas that were seeded differently, or of which only one loaded a checkpoint, would
    otherwise diverge silently, since only gradients are exchanged per step.  A no-op for world_size 1."""
    if world_size() == 1:
        return
    if hasattr(model, "flat_parameters") and next(model.parameters()).is_cuda:
        _broadcast_(model.flat_parameters(), src)          # one 1.13 MB broadcast; the nn.Parameters are views of it
    else:
        for p in model.parameters():
            _broadcast_(p.data, src)
    for b in model.buffers():
        _broadcast_(b, src)
    if optimizer is not None and hasattr(optimizer, "_m"):
        dev = next(model.parameters()).device
        n = sum(p.numel() for p in model.parameters())
        if optimizer._m is None or optimizer._m.device != dev:
            optimizer._m = torch.zeros(n, dtype=torch.float32, device=dev)
            optimizer._v = torch.zeros(n, dtype=torch.float32, device=dev)
        _broadcast_(optimizer._m, src)
        _broadcast_(optimizer._v, src)
        if getattr(optimizer, "capturable", False) and optimizer._step_dev is not None:
            optimizer._step = int(optimizer._step_dev.item())
        step = torch.tensor([optimizer._step], dtype=torch.int64)
        if dist.get_backend() != "gloo":
            step = step.to(dev)
        dist.broadcast(step, src=src)
        optimizer._step = int(step.item())
        optimizer._step_dev = None


def broadcast_buffers(model, src: int = 0):
    """BN running statistics differ slightly per replica; make rank ``src``'s authoritative before a
    save (DDP ``broadcast_buffers`` semantics)."""
    if world_size() == 1:
        return
    for b in model.buffers():
        _broadcast_(b, src)
