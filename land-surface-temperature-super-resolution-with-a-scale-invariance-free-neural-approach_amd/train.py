"""Training-step harness with the reference's step semantics (train_model_B_gradFTM.py:86-121 for
'sr2', train_model_B_predef_filters.py:98-137 for 'sr1'), minus the per-batch host metrics
(psnr/ssim on .cpu().numpy(), :126-127) which are outside the fwd+bwd metric (SURVEY.md §8 d).

    lst_ndvi = cat(lst_up, ndvi); sr = model(lst_ndvi)
    ds, pl, loss = SIF loss(sr, lst, ndvi; mean, std, alpha, gamma);  loss.backward();  optimizer.step()
"""
from __future__ import annotations

import torch

from . import distributed as dp
from .sif_ops import huber_loss, sif_loss


def train_step(model, optimizer, lst, lst_up, ndvi, stats, alpha, gamma, kind="sr2", sync_grads=True):
    """One optimisation step.  Returns device scalars (ds_loss, percep_loss, loss) -- no host sync.

    ``stats`` is the dataset's ``.stats`` dict (the reference reads the module-global
    ``train_ds.stats``, train_model_B_gradFTM.py:99-100)."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    lst_ndvi = torch.cat((lst_up, ndvi), dim=1)
    sr = model(lst_ndvi)
    ds, pl, loss = sif_loss(kind, sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], alpha, gamma)
    loss.backward()
    if sync_grads:
        dp.allreduce_gradients(model, optimizer)
    optimizer.step()
    return ds, pl, loss


@torch.inference_mode()
def eval_step(model, lst, lst_up, ndvi, stats, alpha, gamma, kind="sr2"):
    """test_step semantics (train_model_B_gradFTM.py:141-237): eval mode, no gradient."""
    model.eval()
    sr = model(torch.cat((lst_up, ndvi), dim=1))
    return sif_loss(kind, sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], alpha, gamma)


def si_train_step(model, optimizer, lst_4km_up, ndvi_1km, lst_1km, sync_grads=True):
    """The scale-invariance baseline's step (train_model_B_scale_invariance.py:86-103): the model is trained one scale
    down (4 km -> 1 km, 64x64 patches) with a plain ``nn.HuberLoss`` against the 1 km LST.  Returns the device loss."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    sr = model(torch.cat((lst_4km_up, ndvi_1km), dim=1))
    loss = huber_loss(sr, lst_1km)
    loss.backward()
    if sync_grads:
        dp.allreduce_gradients(model, optimizer)
    optimizer.step()
    return loss


class GraphedTrainStep:
    """One whole training step -- cat, forward, SIF loss, backward, Adam -- captured once into a hipGraph
    (``torch.cuda.CUDAGraph``) and replayed per batch.  Every kernel of the step is enqueued by two C-ABI calls plus
    the loss and optimizer launches; none allocates outside torch's graph pool, synchronises, or reads a host value that
    changes between steps (``FlatAdam(capturable=True)`` keeps its step count on the device), so replay removes the
    host cost of ~190 launches per step.  That matters at small batch (at batch 64 the GPU is the bottleneck).
    Single GPU: the gradient all-reduce of data-parallel runs is not part of the captured region."""

    def __init__(self, model, optimizer, batch, stats, alpha, gamma, kind="sr2", hr=256, device=None):
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FlatAdam(..., capturable=True)")
        dev = device or next(model.parameters()).device
        self.model, self.opt = model, optimizer
        self.lst = torch.zeros((batch, 1, hr // 4, hr // 4), dtype=torch.float32, device=dev)
        self.lst_up = torch.zeros((batch, 1, hr, hr), dtype=torch.float32, device=dev)
        self.ndvi = torch.zeros((batch, 1, hr, hr), dtype=torch.float32, device=dev)
        args = (stats, alpha, gamma, kind)

        def step():
            # detached: a caller holding the returned loss must not keep the step's autograd graph (and with it the
            # parameters' AccumulateGrad nodes, bound to the stream they were made on) alive into the capture --
            # torch then syncs the capture stream with that stream and the capture is invalid
            out = train_step(model, optimizer, self.lst, self.lst_up, self.ndvi, *args, sync_grads=False)
            return tuple(t.detach() for t in out)

        self._eager = step
        self.graph = None
        self._warm = 0
        self.out = None

    def __call__(self, lst, lst_up, ndvi):
        """Copies the batch into the static inputs and runs the step; returns device scalars (ds, pl, loss).  The first
        three calls run eagerly (flat-buffer set-up, allocator pools, optimizer state), the fourth captures."""
        self.lst.copy_(lst); self.lst_up.copy_(lst_up); self.ndvi.copy_(ndvi)
        if self.graph is not None:
            self.graph.replay()
            return self.out
        if self._warm < 3:
            self._warm += 1
            return self._eager()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):          # records the step's kernels; nothing runs yet
            self.out = self._eager()
        self.graph = g
        g.replay()                         # this call's step
        return self.out
