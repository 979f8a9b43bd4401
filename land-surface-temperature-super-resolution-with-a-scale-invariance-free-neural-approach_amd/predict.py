"""Inference call pattern of the reference's predict.py:84-103, batched.

The reference runs one eval-mode forward per 64x64 LST tile (batch 1) and de-normalises with
``* std + mean``; tiles are independent and do not overlap, so here they are stacked and pushed
through the network in large batches (config 4 of BASELINE.json: batch 256).  HDF/GeoTIFF I/O is out
of scope; inputs are the already normalised ``lst_up`` / ``ndvi`` tiles.
"""
from __future__ import annotations

import torch


@torch.inference_mode()
def predict_tiles(model, lst_up, ndvi, stats, batch=256):
    """(N,1,256,256) x2 -> (N,1,256,256) de-normalised LST [K]  (predict.py:100-101)."""
    model.eval()
    out = torch.empty_like(lst_up)
    for i in range(0, lst_up.shape[0], batch):
        x = torch.cat((lst_up[i:i + batch], ndvi[i:i + batch]), dim=1)
        out[i:i + batch] = model(x) * stats["std_lst"] + stats["mean_lst"]
    return out


def tile_granule(lst_norm, ndvi_norm, window=64):
    """Cut a normalised LST raster (h,w) and its 4x NDVI raster into the non-overlapping tiles of
    predict.py:84-95 (ragged edge tiles are skipped, as in the reference).  Returns
    (lst_tiles (N,1,64,64), ndvi_tiles (N,1,256,256), [(i,j)...])."""
    h, w = lst_norm.shape
    lst_t, ndvi_t, pos = [], [], []
    for i in range(0, h - window + 1, window):
        for j in range(0, w - window + 1, window):
            lst_t.append(lst_norm[i:i + window, j:j + window])
            ndvi_t.append(ndvi_norm[4 * i:4 * (i + window), 4 * j:4 * (j + window)])
            pos.append((i, j))
    return torch.stack(lst_t)[:, None], torch.stack(ndvi_t)[:, None], pos
