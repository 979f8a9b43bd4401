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


@torch.inference_mode()
def predict_granule(model, lst_g, ndvi_g, stats, window=64, batch=256):
    """The whole block loop of predict.py:84-103 on the device: raw LST raster (h,w) [K] + raw NDVI raster
    (4h,4w) -> super-resolved LST raster (4h,4w) [K].  Tiles are cut, normalised, bicubic-upsampled and
    concatenated by one kernel, pushed through the network in batches, de-normalised and pasted by another;
    pixels of ragged edge tiles stay 0, as in the reference (``LST_SR = np.zeros(...)``)."""
    from . import pipeline
    model.eval()
    x, tiles = pipeline.granule_to_tiles(lst_g, ndvi_g, stats, window=window, clip_ndvi=True)
    sr = torch.empty((x.shape[0], 1, 4 * window, 4 * window), dtype=torch.float32, device=x.device)
    for i in range(0, x.shape[0], batch):
        sr[i:i + batch] = model(x[i:i + batch])
    out = torch.zeros((4 * lst_g.shape[0], 4 * lst_g.shape[1]), dtype=torch.float32, device=x.device)
    return pipeline.tiles_to_granule(sr, out, tiles, window, stats)


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


class GraphedPredictor:
    """BASELINE.json config 4: eval-mode forward of a fixed batch shape captured once into a HIP graph
    (``torch.cuda.CUDAGraph`` == hipGraph on ROCm) and replayed per batch of tiles.

    The whole forward is one C-ABI call that only enqueues kernels on the current stream (no allocation,
    no host sync inside the library), so stream capture records the ~60 launches as graph nodes; replay
    removes the per-launch host cost, which matters at small batch (predict.py runs batch 1 per tile).
    The de-normalisation ``* std + mean`` (predict.py:101) is part of the captured region.
    """

    def __init__(self, model, batch, stats, hr=256, device=None):
        self.model = model.eval()
        dev = device or next(model.parameters()).device
        self.batch, self.stats = int(batch), stats
        self.x = torch.zeros((self.batch, 2, hr, hr), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.inference_mode():
            for _ in range(2):                      # warm-up: flat-buffer setup, allocator pools
                self.out = self.model(self.x) * stats["std_lst"] + stats["mean_lst"]
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.inference_mode(), torch.cuda.graph(self.graph):
            self.out = self.model(self.x) * stats["std_lst"] + stats["mean_lst"]

    @torch.inference_mode()
    def __call__(self, lst_up, ndvi):
        n = lst_up.shape[0]
        if n > self.batch:
            raise ValueError(f"batch {n} exceeds the captured batch {self.batch}")
        self.x[:n, 0:1].copy_(lst_up)
        self.x[:n, 1:2].copy_(ndvi)
        self.graph.replay()
        return self.out[:n].clone()
