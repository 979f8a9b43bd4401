"""On-device input pipeline: what the reference does per tile on the host before the
model -- ``ModisDatasetB.__getitem__`` (dataset.py:134-142) and the block loop of predict.py:84-103.

    z-score of the LST tile -> us.upsampling (cv2.resize INTER_CUBIC x4, utils.py:163-180)
    NDVI clip to [-1, 1] (predict.py:88-89) + z-score -> torch.cat((lst_up, ndvi), 1)

fused into one kernel (``sifsr_tiles_prepare``), plus the paste-back with ``* std + mean``
(``sifsr_tiles_paste``, predict.py:101-103).  OpenCV is absent in the build container, so the resampler is
pinned against ``F.interpolate(mode='bicubic', align_corners=False)`` (same A = -0.75 kernel, half-pixel
centres and edge clamp as INTER_CUBIC); parity with cv2 itself is *unpinned*.
"""
from __future__ import annotations

import torch

from . import _lib

_UNIT = {"mean_lst": 0.0, "std_lst": 1.0, "mean_ndvi": 0.0, "std_ndvi": 1.0}


def prepare_tiles(lst, ndvi, stats=None, clip_ndvi=False):
    """lst (T,1,w,w), ndvi (T,1,4w,4w) -> model input (T,2,4w,4w) = cat(bicubic4(z(lst)), z(clip(ndvi))).
    ``stats=None``: inputs are already normalised (the training DataLoader's tensors)."""
    _lib.require_gpu(lst, "lst"); _lib.require_gpu(ndvi, "ndvi")
    T, c, w, w2 = lst.shape
    if c != 1 or w != w2 or tuple(ndvi.shape) != (T, 1, 4 * w, 4 * w):
        raise _lib.SifsrError(f"prepare_tiles expects lst (T,1,w,w) and ndvi (T,1,4w,4w); got {tuple(lst.shape)}, {tuple(ndvi.shape)}")
    st = stats or _UNIT
    x = torch.empty((T, 2, 4 * w, 4 * w), dtype=torch.float32, device=lst.device)
    _lib.call("sifsr_tiles_prepare", lst, ndvi, x, T, 1, w, 0, 0, 0, float(st["mean_lst"]), float(st["std_lst"]),
              float(st["mean_ndvi"]), float(st["std_ndvi"]), 1 if clip_ndvi else 0, _lib.stream_ptr(lst.device))
    return x


def bicubic_up4(img):
    """``us.upsampling(img, (4, 4))`` (utils.py:163-180, cv2.resize INTER_CUBIC) for a batch (T,1,w,w) -> (T,1,4w,4w):
    the resampler of ``prepare_tiles`` alone (unit statistics, the NDVI half of the kernel's output discarded)."""
    T, _, w, _ = img.shape
    ndvi = torch.zeros((T, 1, 4 * w, 4 * w), dtype=torch.float32, device=img.device)
    return prepare_tiles(img.contiguous(), ndvi)[:, 0:1]


def granule_to_tiles(lst_g, ndvi_g, stats, window=64, clip_ndvi=True):
    """Raw granule rasters lst_g (h,w) [K] and ndvi_g (4h,4w) -> (x (T,2,4win,4win), (tiles_y, tiles_x)) for the
    non-overlapping full tiles of predict.py:84-95 (ragged edge tiles are skipped, as in the reference)."""
    _lib.require_gpu(lst_g, "lst granule"); _lib.require_gpu(ndvi_g, "ndvi granule")
    h, w = lst_g.shape
    if tuple(ndvi_g.shape) != (4 * h, 4 * w):
        raise _lib.SifsrError("ndvi granule must be 4x the LST granule")
    ty, tx = h // window, w // window
    if ty < 1 or tx < 1:
        raise _lib.SifsrError("granule smaller than one window")
    x = torch.empty((ty * tx, 2, 4 * window, 4 * window), dtype=torch.float32, device=lst_g.device)
    _lib.call("sifsr_tiles_prepare", lst_g, ndvi_g, x, ty, tx, window, h, w, 1, float(stats["mean_lst"]), float(stats["std_lst"]),
              float(stats["mean_ndvi"]), float(stats["std_ndvi"]), 1 if clip_ndvi else 0, _lib.stream_ptr(lst_g.device))
    return x, (ty, tx)


def tiles_to_granule(sr, out, tiles, window, stats):
    """sr (T,1,4win,4win) normalised -> out (4h,4w) [K] at the tiles' positions (predict.py:101-103)."""
    _lib.require_gpu(sr, "sr"); _lib.require_gpu(out, "output granule")
    ty, tx = tiles
    _lib.call("sifsr_tiles_paste", sr, out, ty, tx, window, out.shape[1] // 4, float(stats["mean_lst"]), float(stats["std_lst"]),
              _lib.stream_ptr(sr.device))
    return out


def l4pool4(x):
    """us.downsampling (utils.py:183-213): (mean of x**4 over 4x4 blocks)**0.25 of a (B,1,H,W) batch."""
    _lib.require_gpu(x, "x")
    B, c, H, W = x.shape
    if c != 1:
        raise _lib.SifsrError("l4pool4 expects (B,1,H,W)")
    out = torch.empty((B, 1, H // 4, W // 4), dtype=torch.float32, device=x.device)
    _lib.call("sifsr_l4pool4", x, out, B, H, W, _lib.stream_ptr(x.device))
    return out


_DELTA9 = None


def decimate4_bic(x):
    """us.downscale_LST_SR_to_LR_test(..., deci_type='bic') (utils.py:1716-1748): reflect-pad 4, bicubic /4, crop --
    the consistency operator of the SIF loss WITHOUT the Gaussian blur (that function pads but never convolves).
    Runs the fused blur+decimate kernel with an identity PSF.  (B,1,H,W), H and W multiples of 32, >= 64."""
    import ctypes
    global _DELTA9
    if _DELTA9 is None:
        _DELTA9 = (ctypes.c_float * 9)(0, 0, 0, 0, 1, 0, 0, 0, 0)
    _lib.require_gpu(x, "x")
    B, c, H, W = x.shape
    out = torch.empty((B, c, H // 4, W // 4), dtype=torch.float32, device=x.device)
    _lib.call("sifsr_gauss9_decimate4_fwd", x, _DELTA9, out, B * c, H, W, _lib.stream_ptr(x.device))
    return out


def scale_invariance_inputs(lst, ndvi, stats):
    """ModisDatasetB_scale_invariance.__getitem__ (dataset.py:240-263) for a batch, on the device:
    normalised lst (B,1,64,64) and ndvi (B,1,256,256) -> (lst_4km_up (B,1,64,64), ndvi_1km (B,1,64,64), lst).
      ndvi_1km   = decimate4_bic(ndvi)
      lst_4km    = l4pool4(lst*std + mean)                              [K, 16x16]
      lst_4km_up = (bicubic x4 of lst_4km - mean) / std"""
    ndvi_1km = decimate4_bic(ndvi)
    lst_4km = l4pool4(lst * stats["std_lst"] + stats["mean_lst"])
    x = prepare_tiles(lst_4km, ndvi_1km, {"mean_lst": stats["mean_lst"], "std_lst": stats["std_lst"], "mean_ndvi": 0.0,
                                          "std_ndvi": 1.0})
    return x[:, 0:1], x[:, 1:2], lst
