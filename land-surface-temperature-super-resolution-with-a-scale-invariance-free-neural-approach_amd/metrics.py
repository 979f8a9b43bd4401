"""On-device train-time metrics: ``us.psnr_skimage`` / ``us.ssim_skimage`` (utils.py:548-578)
without the per-batch ``.detach().cpu().numpy()`` + scikit-image host stall of train_model_B_gradFTM.py:126-127.

Semantics follow scikit-image 0.22 (the reference's pinned version, environment.yml:402) as the reference calls
it: ``data_range = targets.max() - targets.min()`` over the WHOLE target batch, ``structural_similarity`` defaults
(7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03, mean over the window-valid interior),
``peak_signal_noise_ratio = 10 log10(range^2 / mse)``, then the mean over the batch.  scikit-image is not
installed in the build container: the oracle is a numpy/scipy restatement of that published algorithm and the
parity with scikit-image itself is *unpinned*.
"""
from __future__ import annotations

import torch

from . import _lib


def psnr_ssim(predictions, targets):
    """(B,1,H,W) x2 -> (psnr, ssim) as 0-d device tensors (batch means); no host synchronisation."""
    _lib.require_gpu(predictions, "predictions"); _lib.require_gpu(targets, "targets")
    if predictions.shape != targets.shape or predictions.dim() != 4 or predictions.shape[1] != 1:
        raise _lib.SifsrError("psnr_ssim expects two (B,1,H,W) tensors")
    B, _, H, W = predictions.shape
    nbytes = _lib.call("sifsr_psnr_ssim_scratch_bytes", B, H, W)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=predictions.device)
    out = torch.empty(2, dtype=torch.float32, device=predictions.device)
    _lib.call("sifsr_psnr_ssim", predictions.detach(), targets.detach(), B, H, W, scratch, nbytes, out,
              _lib.stream_ptr(predictions.device))
    return out[0], out[1]


def psnr_skimage(predictions, targets):
    """Drop-in for us.psnr_skimage on device tensors (returns a 0-d device tensor; call .item() when needed)."""
    return psnr_ssim(predictions, targets)[0]


def ssim_skimage(predictions, targets):
    """Drop-in for us.ssim_skimage on device tensors (returns a 0-d device tensor)."""
    return psnr_ssim(predictions, targets)[1]
