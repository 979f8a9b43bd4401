"""Fourier-domain evaluation on the device -- the only place the reference uses an FFT:

    fourier_dict[m] = np.fft.fftshift(np.abs(sp.fft.fft2(LST_m)))       compare_methods.py:312-324
    us.compute_2D_attenuation_spectra(fourier_dict[m])                   utils.py:598-636
    us.get_FRR / get_FRO / get_FRU (pb, rb, xb)                          utils.py:638-662

``fft2_magnitude`` and ``attenuation_spectra`` run as hand-written HIP kernels (radix-2 FFT in LDS, float64);
the three frequency-restoration scores are a few hundred scalar operations on the 1-D spectra and stay on the
host, vectorised.  Image sides must be powers of two (the reference's 256x256 evaluation tiles).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def _run(img, want_mag, want_spec):
    _lib.require_gpu(img, "image")
    x = img if img.dim() == 3 else img[None]
    if x.dim() != 3:
        raise _lib.SifsrError("expected an (H,W) image or a (B,H,W) stack")
    x = x.contiguous()
    B, H, W = x.shape
    nbytes = _lib.call("sifsr_fft2_attenuation_scratch_bytes", B, H, W)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    nr = min(H // 2, W // 2) - 1
    mag = torch.empty((B, H, W), dtype=torch.float32, device=x.device) if want_mag else None
    spec = torch.empty((B, nr + 1), dtype=torch.float32, device=x.device) if want_spec else None
    _lib.call("sifsr_fft2_attenuation", x, B, H, W, scratch, nbytes, mag, spec, _lib.stream_ptr(x.device))
    squeeze = img.dim() == 2
    return (mag[0] if squeeze and mag is not None else mag), (spec[0] if squeeze and spec is not None else spec)


def fft2_magnitude(img):
    """fftshift(|fft2(img)|) of an (H,W) image or (B,H,W) stack (compare_methods.py:312)."""
    return _run(img, True, False)[0]


def attenuation_spectra(img):
    """1-D attenuation spectrum [dB] of the IMAGE(s): compute_2D_attenuation_spectra(fftshift(|fft2(img)|)),
    utils.py:598-636, FFT and ring means fused on the device.  Element 0 is 1 (the reference's f0/f0)."""
    return _run(img, False, True)[1]


def compute_2D_attenuation_spectra(im):
    """Drop-in name of utils.py:598: takes the shifted magnitude ``im`` like the reference does.  The rings are
    then evaluated with torch ops on ``im``'s device (one pass, vectorised) -- use ``attenuation_spectra`` on
    the image itself to run FFT + rings in the hand-written kernels."""
    t = torch.as_tensor(im)
    H, W = t.shape
    cy, cx = H // 2, W // 2
    yy = torch.arange(H, device=t.device)[:, None] - cy
    xx = torch.arange(W, device=t.device)[None, :] - cx
    d2 = (yy * yy + xx * xx).to(torch.float64)
    ring = torch.ceil(torch.sqrt(d2)).to(torch.int64) - 1          # r^2 < d^2 <= (r+1)^2  (exact for these sizes)
    nr = min(cy - 1, cx - 1)
    td = t.to(torch.float64)
    f0 = td[cy, cx]
    valid = (ring >= 0) & (ring < nr)
    sums = torch.zeros(nr, dtype=torch.float64, device=t.device).index_add_(0, ring[valid], td[valid])
    cnts = torch.zeros(nr, dtype=torch.float64, device=t.device).index_add_(0, ring[valid], torch.ones_like(td[valid]))
    out = 10 * (torch.log10(sums / cnts) - torch.log10(f0))
    return [1.0] + out.cpu().tolist()


def _arr(x):
    return np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x, dtype=np.float64)


def get_PFR(rb, xb):
    rb, xb = _arr(rb), _arr(xb)
    return float(np.maximum(rb - xb, 0).sum())


def get_AFR(pb, rb, xb):
    pb, rb, xb = _arr(pb), _arr(rb), _arr(xb)
    low = np.minimum(xb, rb)
    return float((np.maximum(np.minimum(pb, rb), low) - low).sum())


def get_FRR(pb, rb, xb):
    return get_AFR(pb, rb, xb) / get_PFR(rb, xb)


def get_FRO(pb, rb, xb):
    pb, rb = _arr(pb), _arr(rb)
    return float((rb - np.maximum(pb, rb)).sum() / rb.sum())


def get_FRU(pb, rb, xb):
    pb, xb = _arr(pb), _arr(xb)
    return float((xb - np.minimum(pb, xb)).sum() / xb.sum())
