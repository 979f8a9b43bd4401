"""Training-step harness with the reference's step semantics (train_model_B_gradFTM.py:86-121 for
'sr2', train_model_B_predef_filters.py:98-137 for 'sr1'), minus the per-batch host metrics
(psnr/ssim on .cpu().numpy(), :126-127) which are outside the fwd+bwd metric (SURVEY.md §8 d).

    lst_ndvi = cat(lst_up, ndvi); sr = model(lst_ndvi)
    ds, pl, loss = SIF loss(sr, lst, ndvi; mean, std, alpha, gamma);  loss.backward();  optimizer.step()
"""
from __future__ import annotations

import copy

import torch

from . import distributed as dp
from . import metrics as _metrics
from .sif_ops import huber_loss, sif_loss, sif_loss_with_grad


def train_step(model, optimizer, lst, lst_up, ndvi, stats, alpha, gamma, kind="sr2", sync_grads=True, return_sr=False):
    """One optimisation step.  Returns device scalars (ds_loss, percep_loss, loss) -- no host sync --
    and, with ``return_sr``, the detached training-mode prediction as a fourth item (the tensor the reference scores
    with PSNR / SSIM, train_model_B_gradFTM.py:126-127).

    ``stats`` is the dataset's ``.stats`` dict (the reference reads the module-global
    ``train_ds.stats``, train_model_B_gradFTM.py:99-100)."""
    model.train()
    optimizer.zero_grad(set_to_none=True)
    lst_ndvi = torch.cat((lst_up, ndvi), dim=1)
    sr = model(lst_ndvi)
    # loss.backward() == sr.backward(d loss / d sr): the fused loss op returns that gradient directly (sif_ops.sif_loss_with_grad)
    ds, pl, loss, dsr = sif_loss_with_grad(kind, sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], alpha, gamma)
    sr.backward(dsr)
    if sync_grads:
        dp.allreduce_gradients(model, optimizer)
    optimizer.step()
    if return_sr:
        return ds, pl, loss, sr.detach()
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


class ModelCheckpoint:
    """Early stopping with the semantics of the reference's ``us.model_checkpoint`` (utils.py:667-714): keeps a deep copy
    of the ``state_dict`` of the best epoch by a monitored metric (lower is better; ``>=`` counts as no improvement),
    counts the epochs without improvement, and sets ``train_state`` to ``'break'`` when the patience is used up or when
    the last epoch arrives with a non-zero counter; the training loop then restores ``saved_state``."""

    def __init__(self, n_epochs, patience=5):
        self.patience, self.max_epochs = patience, n_epochs
        self.curr_patience = 0
        self.saved_state = self.saved_best_value = self.curr_epoch = self.best_epoch = self.train_state = None

    def test_update(self, model, metrics, val_monitored, epoch):
        self.curr_epoch = epoch
        value = metrics[val_monitored][-1]
        if epoch == 1:                                   # first epoch: always the best so far, state untouched
            self.best_epoch, self.saved_best_value = epoch, value
            self.saved_state = copy.deepcopy(model.state_dict())
        elif value >= self.saved_best_value:             # the reference's own test (utils.py:687): a NaN metric fails it and
            self.curr_patience += 1                      # therefore counts as an IMPROVEMENT there -- mirrored, not "fixed"
            spent = self.curr_patience >= self.patience
            self.train_state = "break" if spent or epoch == self.max_epochs else "continue"
        else:
            self.best_epoch, self.saved_best_value, self.curr_patience = epoch, value, 0
            self.saved_state = copy.deepcopy(model.state_dict())
            self.train_state = "continue"


def train_epoch(model, loader, optimizer, stats, alpha, gamma, kind="sr2", device="cuda", with_metrics=True):
    """The reference's ``train_step`` over a DataLoader (train_model_B_gradFTM.py:84-138): per batch ``.to(device)`` x3
    (:89) and one optimisation step; returns the epoch means (ds_loss, percep_loss, loss, psnr, ssim), PSNR / SSIM of
    the training-mode prediction against ``lst_up`` as the reference scores them (:126-127) but on the device
    (``metrics.psnr_ssim``).  The per-batch scalars stay on the device and are read back ONCE per
    epoch; the reference calls ``.item()`` three times and copies two full tensors to the host per batch."""
    acc = torch.zeros(5, dtype=torch.float64, device=device)
    n = 0
    for lst, lst_up, ndvi in loader:
        lst, lst_up, ndvi = lst.to(device), lst_up.to(device), ndvi.to(device)
        ds, pl, loss, sr = train_step(model, optimizer, lst, lst_up, ndvi, stats, alpha, gamma, kind, return_sr=True)
        acc[0] += ds.detach(); acc[1] += pl.detach(); acc[2] += loss.detach()
        if with_metrics:
            ps, ss = _metrics.psnr_ssim(sr, lst_up)
            acc[3] += ps; acc[4] += ss
        n += 1
    return tuple((acc / max(n, 1)).tolist())


@torch.inference_mode()
def eval_epoch(model, loader, stats, alpha, gamma, kind="sr2", device="cuda", with_metrics=True):
    """The reference's ``test_step`` (train_model_B_gradFTM.py:141-237): eval mode, no gradient; epoch means of
    (ds_loss, percep_loss, loss, psnr, ssim)."""
    model.eval()
    acc = torch.zeros(5, dtype=torch.float64, device=device)
    n = 0
    for lst, lst_up, ndvi in loader:
        lst, lst_up, ndvi = lst.to(device), lst_up.to(device), ndvi.to(device)
        sr = model(torch.cat((lst_up, ndvi), dim=1))
        ds, pl, loss = sif_loss(kind, sr, lst, ndvi, stats["mean_lst"], stats["std_lst"], alpha, gamma)
        acc[0] += ds; acc[1] += pl; acc[2] += loss
        if with_metrics:
            ps, ss = _metrics.psnr_ssim(sr, lst_up)
            acc[3] += ps; acc[4] += ss
        n += 1
    return tuple((acc / max(n, 1)).tolist())


def fit(model, train_dataset, val_dataset, n_epochs, batch_size, optimizer, alpha, gamma, kind="sr2", device="cuda",
        checkpoint=None, shuffle=True, generator=None):
    """The reference's ``train()`` (train_model_B_gradFTM.py:240-354): two shuffled DataLoaders (:295-296), per epoch
    one training and one validation pass, the same ``metrics`` dict (``train_loss, train_dsloss, train_perceploss,
    train_psnr, train_ssim, val_*``, ``best_epoch``) and early stopping through ``checkpoint`` (``ModelCheckpoint`` or
    the reference's own ``us.model_checkpoint`` object) with restore of the best state (:338-352).  The normalisation
    statistics come from ``train_dataset.stats`` as in the reference's step (:99-100).  Returns (model, metrics)."""
    from torch.utils.data import DataLoader
    tl = DataLoader(train_dataset, batch_size=batch_size, shuffle=shuffle, generator=generator)
    vl = DataLoader(val_dataset, batch_size=batch_size, shuffle=shuffle, generator=generator)
    stats = train_dataset.stats
    keys = ("dsloss", "perceploss", "loss", "psnr", "ssim")
    metrics = {f"{split}_{k}": [] for split in ("train", "val") for k in keys}
    for epoch in range(1, n_epochs + 1):
        for k, v in zip(keys, train_epoch(model, tl, optimizer, stats, alpha, gamma, kind, device)):
            metrics[f"train_{k}"].append(v)
        for k, v in zip(keys, eval_epoch(model, vl, stats, alpha, gamma, kind, device)):
            metrics[f"val_{k}"].append(v)
        if checkpoint is not None:
            checkpoint.test_update(model, metrics, "val_loss", epoch)
            if checkpoint.train_state == "break":
                metrics["best_epoch"] = checkpoint.best_epoch
                model.load_state_dict(checkpoint.saved_state)
                break
            if checkpoint.train_state == "continue" and epoch == n_epochs:    # train_model_B_gradFTM.py:342-344: the key exists
                metrics["best_epoch"] = n_epochs                              # only then (a one-epoch run leaves train_state None)
    return model, metrics


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
