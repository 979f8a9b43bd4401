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
