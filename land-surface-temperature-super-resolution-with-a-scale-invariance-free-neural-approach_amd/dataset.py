"""Synthetic drop-in for the reference's ``dataset.ModisDatasetB`` (dataset.py:29-142).

The reference reads GeoTIFF pairs listed in ``data/ModisDatasetB.csv`` and ``data/statistics.json``
(neither is shipped, and GDAL/OpenCV are not installed here -- SURVEY.md §0), so this class keeps
the constructor signature, ``__len__``, the ``.stats`` dict and the ``__getitem__`` contract
  (lst (1,64,64), lst_up (1,256,256), ndvi (1,256,256)) float32 numpy, z-scored ('norm')
and fills them with seeded synthetic data of BASELINE.md §3: lst ~ N(0,1), ndvi ~ N(0,1) clipped to
+-3, lst_up = bicubic x4 of lst.  (The reference uses cv2.INTER_CUBIC for lst_up; OpenCV is absent,
so that resampler is 'parity unpinned' -- here lst_up is F.interpolate(bicubic, align_corners=False).)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

# statistics of the 83 shipped MODIS/ASTER test pairs (SURVEY.md §8 c), used as synthetic constants
DEFAULT_STATS = {"mean_lst": 307.2378, "std_lst": 5.5698, "mean_ndvi": 0.6452, "std_ndvi": 0.1683}


class ModisDatasetB(Dataset):
    def __init__(self, csv_path=None, transf="norm", split="Train", time="Both", length=16, seed=1234, hr=256):
        if transf != "norm":
            raise NotImplementedError("only transf='norm' (z-score, dataset.py:134-139) is provided")
        self.csv_path, self.transf, self.split, self.time = csv_path, transf, split, time
        self.stats = dict(DEFAULT_STATS)
        self.length, self.seed, self.hr = int(length), int(seed), int(hr)

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        if idx < 0 or idx >= self.length:
            raise IndexError(idx)
        rs = np.random.RandomState(self.seed + 7919 * int(idx) + (0 if self.split == "Train" else 104729))
        lr = self.hr // 4
        lst = rs.standard_normal((lr, lr)).astype(np.float32)
        ndvi = np.clip(rs.standard_normal((self.hr, self.hr)), -3, 3).astype(np.float32)
        lst_up = F.interpolate(torch.from_numpy(lst)[None, None], scale_factor=4, mode="bicubic",
                               align_corners=False)[0, 0].numpy()
        return np.expand_dims(lst, 0), np.expand_dims(lst_up, 0), np.expand_dims(ndvi, 0)


class ModisDatasetB_scale_invariance(ModisDatasetB):
    """Synthetic drop-in for ``dataset.ModisDatasetB_scale_invariance`` (dataset.py:145-263): same constructor
    and ``.stats``; ``__getitem__`` returns (lst_4km_up (1,64,64), ndvi_1km (1,64,64), lst (1,64,64)) computed from
    the synthetic (lst, ndvi) pair exactly as dataset.py:256-263 does (bicubic /4 of NDVI without blur, norm-L4
    decimation of the de-normalised LST, bicubic x4 back, re-normalisation), with torch CPU ops."""

    def __getitem__(self, idx):
        lst, _, ndvi = super().__getitem__(idx)
        lst_t, ndvi_t = torch.from_numpy(lst)[None], torch.from_numpy(ndvi)[None]
        nd = F.interpolate(F.pad(ndvi_t, (4, 4, 4, 4), mode="reflect"), scale_factor=0.25, mode="bicubic")[:, :, 1:-1, 1:-1]
        k = lst_t * self.stats["std_lst"] + self.stats["mean_lst"]
        k = k.unfold(3, 4, 4).unfold(2, 4, 4).pow(4).sum((-1, -2)).div(16).pow(0.25)
        up = F.interpolate(k, scale_factor=4, mode="bicubic", align_corners=False)
        up = (up - self.stats["mean_lst"]) / self.stats["std_lst"]
        return up[0].numpy(), nd[0].numpy(), lst


def synthetic_device_batch(batch, device, seed=1234, hr=256):
    """BASELINE.md §3 bench inputs, generated once on the device: (lst, lst_up, ndvi)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    lr = hr // 4
    lst = torch.randn((batch, 1, lr, lr), generator=g)
    ndvi = torch.randn((batch, 1, hr, hr), generator=g).clamp_(-3, 3)
    lst_up = F.interpolate(lst, scale_factor=4, mode="bicubic", align_corners=False)
    return lst.to(device), lst_up.contiguous().to(device), ndvi.to(device)
