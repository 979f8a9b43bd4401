"""SIF loss operators -- drop-ins for the reference's ``utils`` hot-path functions, on MI355X.

``downscale_LST_SR_to_LR`` (utils.py:1671-1706) and ``get_output_ftm`` (utils.py:1833-1860) keep the
reference signatures and return autograd-capable tensors on ``data.device``; the arithmetic runs in
hand-written gfx950 kernels behind the C ABI (include/sifsr_hip.h).  ``sif_loss`` is the fused form
of the whole loss block of the two training scripts (train_model_B_gradFTM.py:99-117,
train_model_B_predef_filters.py:111-133): two launches for the three loss values and one for
d loss / d sr.  No CPU path.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch

from . import _lib


def psf_taps_1d(mtf: float, factor: float = 4.0, hkw=None) -> np.ndarray:
    """Separable factor of ``generate_psf_kernel(1.0, factor, mtf, hkw)`` (utils.py:1615-1639).

    The reference builds a (2h+1)^2 kernel exp(-(i^2+j^2)/2s^2) / sum in float64 and casts to fp32;
    that kernel is the outer product of these normalised 1-D taps (to 1e-9, tests/golden)."""
    fc = 0.5 / factor
    sigma = math.sqrt(-math.log(mtf) / 2) / (math.pi * fc)
    h = int(math.ceil(factor / 1.0)) if hkw is None else int(hkw)
    g = np.exp(-(np.arange(-h, h + 1, dtype=np.float64) ** 2) / (2 * sigma * sigma))
    return g / g.sum()


_TAPS_CACHE = {}


def _taps_c(mtf, factor, hkw):
    key = (float(mtf), float(factor), hkw)
    if key not in _TAPS_CACHE:
        t = psf_taps_1d(mtf, factor, hkw)
        if len(t) != 9:
            raise NotImplementedError("only the 9x9 PSF (factor=4, hkw=None) has a gfx950 kernel")
        _TAPS_CACHE[key] = (ctypes.c_float * 9)(*[float(v) for v in t])
    return _TAPS_CACHE[key]


def _as_images(data):
    _lib.require_gpu(data, "data")
    if data.dim() != 4:
        raise _lib.SifsrError("expected a (B,C,H,W) tensor")
    B, C, H, W = data.shape
    return B * C, H, W


class _Blur(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, taps):
        n, H, W = _as_images(data)
        out = torch.empty_like(data)
        _lib.call("sifsr_gauss9_reflect_fwd", data, taps, out, n, H, W, _lib.stream_ptr(data.device))
        ctx.taps = taps
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n, H, W = _as_images(g)
        gx = torch.empty_like(g)
        _lib.call("sifsr_gauss9_reflect_bwd", g, ctx.taps, gx, n, H, W, _lib.stream_ptr(g.device))
        return gx, None


class _BlurDecimate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, taps):
        n, H, W = _as_images(data)
        out = torch.empty(data.shape[:2] + (H // 4, W // 4), dtype=torch.float32, device=data.device)
        _lib.call("sifsr_gauss9_decimate4_fwd", data, taps, out, n, H, W, _lib.stream_ptr(data.device))
        ctx.taps, ctx.hw = taps, (H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        H, W = ctx.hw
        gx = torch.empty(g.shape[:2] + (H, W), dtype=torch.float32, device=g.device)
        _lib.call("sifsr_gauss9_decimate4_bwd", g, ctx.taps, gx, g.shape[0] * g.shape[1], H, W, _lib.stream_ptr(g.device))
        return gx, None


def downscale_LST_SR_to_LR(data, factor=4, mtf=0.1, padding="same", hkw=None, deci_type="bic"):
    """utils.py:1671-1706.  (B,C,H,W) -> (B,C,H/4,W/4): Gaussian PSF low-pass (reflect border) then
    bicubic /4 (A=-0.75) -- fused in one kernel."""
    if factor != 4 or padding != "same" or deci_type != "bic":
        raise NotImplementedError("only factor=4, padding='same', deci_type='bic' (the hot-path call) is implemented")
    return _BlurDecimate.apply(data.contiguous(), _taps_c(mtf, factor, hkw))


def get_output_ftm(data, factor=4, mtf=0.1, padding="same", hkw=None):
    """utils.py:1833-1860.  Reflect-border Gaussian low-pass of the MTF ('FTM') at ``mtf``."""
    if factor != 4 or padding != "same":
        raise NotImplementedError("only factor=4, padding='same' is implemented")
    return _Blur.apply(data.contiguous(), _taps_c(mtf, factor, hkw))


class _Sobel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, H, W = _as_images(x)
        if x.shape[1] != 1:
            raise _lib.SifsrError("sobel_bank expects (B,1,H,W)")
        out = torch.empty((x.shape[0], 4, H, W), dtype=torch.float32, device=x.device)
        _lib.call("sifsr_sobel4_fwd", x, out, n, H, W, _lib.stream_ptr(x.device))
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, _, H, W = g.shape
        gx = torch.empty((B, 1, H, W), dtype=torch.float32, device=g.device)
        _lib.call("sifsr_sobel4_bwd", g, gx, B, H, W, _lib.stream_ptr(g.device))
        return gx


def sobel_bank(x):
    """``F.conv2d(x, filters_tensor, padding='same')`` with the 4 fixed filters of
    train_model_B_predef_filters.py:38-42,120-128.  (B,1,H,W) -> (B,4,H,W)."""
    return _Sobel.apply(x.contiguous())


class _Huber(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, bscale):
        _lib.require_gpu(a, "input"); _lib.require_gpu(b, "target")
        if a.shape != b.shape:
            raise _lib.SifsrError("huber_loss: shape mismatch")
        n = a.numel()
        nblk = _lib.call("sifsr_huber_partial_blocks", n)
        partials = torch.empty(nblk, dtype=torch.float32, device=a.device)
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        _lib.call("sifsr_huber_fwd", a, b, float(bscale), n, partials, out, _lib.stream_ptr(a.device))
        ctx.save_for_backward(a, b)
        ctx.bscale = float(bscale)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        go = gout.reshape(1).contiguous().float()
        _lib.call("sifsr_huber_bwd", a, b, ctx.bscale, go, a.numel(), ga, _lib.stream_ptr(a.device))
        gb = -ctx.bscale * ga if ctx.needs_input_grad[1] else None
        return ga, gb, None


def huber_loss(input, target, target_scale: float = 1.0):
    """``nn.HuberLoss(reduction='mean', delta=1.0)(input, target_scale*target)`` (train_model_B_gradFTM.py:454)."""
    return _Huber.apply(input.contiguous(), target.contiguous(), target_scale)


class _SifLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sr, lst, ndvi, kind, mean, std, alpha, gamma):
        for t, n in ((sr, "sr"), (lst, "lst"), (ndvi, "ndvi")):
            _lib.require_gpu(t, n)
        B, C, H, W = sr.shape
        if C != 1 or tuple(ndvi.shape) != (B, 1, H, W) or tuple(lst.shape) != (B, 1, H // 4, W // 4):
            raise _lib.SifsrError("sif_loss expects sr (B,1,H,W), lst (B,1,H/4,W/4), ndvi (B,1,H,W)")
        k = {"sr2": 2, "sr1": 1}[kind]
        ws_bytes = _lib.call("sifsr_sif_loss_workspace_bytes", k, B, H, W)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=sr.device)
        losses = torch.empty(3, dtype=torch.float32, device=sr.device)
        need = ctx.needs_input_grad[0]
        dsr = torch.empty_like(sr) if need else None
        _lib.call("sifsr_sif_loss", k, sr, lst, ndvi, B, H, W, float(mean), float(std), float(alpha), float(gamma),
                  _taps_c(0.1, 4, None), _taps_c(0.25, 4, None), ws, ws_bytes, losses, dsr, _lib.stream_ptr(sr.device))
        ctx.dsr = dsr
        ds, pl, loss = losses[0], losses[1], losses[2]
        ctx.mark_non_differentiable(ds, pl)
        ctx.set_materialize_grads(False)     # no zero-filled gradients for ds / pl (two fill launches per step)
        return ds, pl, loss

    @staticmethod
    def backward(ctx, g_ds, g_pl, g_loss):
        dsr = ctx.dsr
        ctx.dsr = None
        if g_loss is None:                   # only ds / pl were used downstream: they carry no gradient
            return (None,) * 8
        if dsr is None:
            raise _lib.SifsrError("sif_loss: backward called twice (d loss / d sr was released after the first call)")
        return dsr * g_loss, None, None, None, None, None, None, None


def sif_loss_with_grad(kind, sr, lst, ndvi, mean, std, alpha, gamma):
    """The same fused loss block WITHOUT an autograd node: returns (ds_loss, percep_loss, loss, d loss / d sr).  A training step
    that ends in ``loss.backward()`` seeds the graph with ones_like(loss) and multiplies d loss / d sr by it -- a fill and a
    16 MB elementwise launch on the serial chain; ``sr.backward(dsr)`` with this gradient is the same step without them
    (train.train_step).  ``sr`` may carry a graph; it is only read."""
    srd, lst, ndvi = sr.detach().contiguous(), lst.contiguous(), ndvi.contiguous()
    for t, n in ((srd, "sr"), (lst, "lst"), (ndvi, "ndvi")):
        _lib.require_gpu(t, n)
    B, C, H, W = srd.shape
    if C != 1 or tuple(ndvi.shape) != (B, 1, H, W) or tuple(lst.shape) != (B, 1, H // 4, W // 4):
        raise _lib.SifsrError("sif_loss expects sr (B,1,H,W), lst (B,1,H/4,W/4), ndvi (B,1,H,W)")
    k = {"sr2": 2, "sr1": 1}[kind]
    ws_bytes = _lib.call("sifsr_sif_loss_workspace_bytes", k, B, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=srd.device)
    losses = torch.empty(3, dtype=torch.float32, device=srd.device)
    dsr = torch.empty_like(srd)
    _lib.call("sifsr_sif_loss", k, srd, lst, ndvi, B, H, W, float(mean), float(std), float(alpha), float(gamma),
              _taps_c(0.1, 4, None), _taps_c(0.25, 4, None), ws, ws_bytes, losses, dsr, _lib.stream_ptr(srd.device))
    return losses[0], losses[1], losses[2], dsr


def sif_loss(kind, sr, lst, ndvi, mean, std, alpha, gamma):
    """Fused loss block of the training step.  kind='sr2': train_model_B_gradFTM.py:99-117;
    kind='sr1': train_model_B_predef_filters.py:111-133.  Returns (ds_loss, percep_loss, loss) as
    0-d device tensors; only ``loss`` carries a gradient (to ``sr``)."""
    return _SifLoss.apply(sr.contiguous(), lst.contiguous(), ndvi.contiguous(), kind, mean, std, alpha, gamma)
