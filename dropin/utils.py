"""``import utils as us`` (train_model_B_gradFTM.py:29, predict.py:16) -- an overlay with every ``us.*`` name the three
training scripts use, the hot-path ones on MI355X:

  us.downscale_LST_SR_to_LR, us.get_output_ftm     utils.py:1671, :1833   -> gfx950 kernels (autograd-capable)
  us.generate_psf_kernel                           utils.py:1615          -> host, float64 -> fp32 (as the reference)
  us.psnr_skimage, us.ssim_skimage                 utils.py:548-578       -> device kernel; accept the numpy arrays the
                                                                             scripts pass (.detach().cpu().numpy()) or tensors
  us.model_checkpoint                              utils.py:667-714       -> sifsr.train.ModelCheckpoint
  us.read_JsonA/B/C, us.save_model, us.load_model  utils.py:718-826       -> plain host code
  us.upsampling                                    utils.py:163-180       -> bicubic x scale, OpenCV INTER_CUBIC semantics
                                                                             (A = -0.75, half-pixel centres, edge clamp)

GeoTIFF / HDF I/O, the classical sharpening baselines and the plotting helpers (GDAL, OpenCV, rasterio: not installed,
out of scope -- SURVEY.md §2) raise ``NotImplementedError`` naming what was asked for.
"""
import json
import os
import sys

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import sifsr  # noqa: E402
from sifsr import metrics as _metrics  # noqa: E402
from sifsr.sif_ops import downscale_LST_SR_to_LR, get_output_ftm, psf_taps_1d  # noqa: E402,F401
from sifsr.train import ModelCheckpoint as model_checkpoint  # noqa: E402,F401

json_load = json.load          # model_perf_aster_formatds.py uses us.json_load


def generate_psf_kernel(res, mtf_res, mtf_fc, half_kernel_width=None):
    """utils.py:1615-1639: the (2h+1)^2 Gaussian PSF as float32 (outer product of the normalised 1-D taps)."""
    t = psf_taps_1d(mtf_fc, mtf_res / res, half_kernel_width)
    return np.outer(t, t).astype(np.float32)


def _dev_pair(predictions, targets):
    dev = torch.device("cuda", torch.cuda.current_device())
    as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, torch.float32).contiguous()
    return as_t(predictions), as_t(targets)


def psnr_skimage(predictions, targets):
    """utils.py:548-552 -> float (batch mean)."""
    return float(_metrics.psnr_ssim(*_dev_pair(predictions, targets))[0])


def ssim_skimage(predictions, targets):
    """utils.py:554-578 -> float (batch mean)."""
    return float(_metrics.psnr_ssim(*_dev_pair(predictions, targets))[1])


def upsampling(img, scale):
    """utils.py:163-180 (``cv2.resize(..., INTER_CUBIC)``): numpy (h,w) -> numpy (h*scale[0], w*scale[1]), same dtype."""
    from sifsr import pipeline
    if scale[0] != 4 or scale[1] != 4:
        raise NotImplementedError("only the x4 bicubic of the hot path (dataset.py:140, predict.py:97) is implemented")
    a = np.asarray(img)
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    return pipeline.bicubic_up4(t[None, None])[0, 0].cpu().numpy().astype(a.dtype, copy=False)


def _read(file, keys):
    with open(file) as f:
        data = json.load(f)
    return tuple(data[k] for k in keys)


def read_JsonA(file):
    """utils.py:718-739."""
    return _read(file, ("dataset_parameter", "modelA_parameters", "hyperparameters", "save_parameters", "device"))


def read_JsonC(file):
    """utils.py:767-787."""
    return _read(file, ("dataset_parameter", "modelC_parameters", "hyperparameters", "save_parameters", "device"))


def read_JsonB(file):
    """utils.py:741-764: (dataset_parameter, modelA_parameters, modelB_parameters, hyperparameters, save_parameters, device)."""
    return _read(file, ("dataset_parameter", "modelA_parameters", "modelB_parameters", "hyperparameters", "save_parameters", "device"))


def save_model(model, path, model_name):
    """utils.py:802-826: the state_dict AND the whole module, as the reference does."""
    torch.save(model.state_dict(), os.path.join(path, model_name + "_state_dict.pt"))
    torch.save(model, os.path.join(path, model_name + ".pt"))


def load_model(model, state_dict_file, device="cpu"):
    """utils.py:791-800."""
    model.load_state_dict(torch.load(state_dict_file, map_location=torch.device(device), weights_only=True))


class OutOfScopeAttribute(AttributeError, NotImplementedError):
    """Raised for names of the reference's utils.py that the MI355X build does not provide.  An AttributeError, so that
    ``hasattr``, ``from utils import *`` (which probes ``__all__``) and ``inspect`` see an ordinary missing attribute; also a
    NotImplementedError, for callers that catch that."""


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    raise OutOfScopeAttribute(
        f"utils.{name}: not part of the SIF-CNN-SR hot path (GDAL / OpenCV / rasterio I/O, classical baselines and plots "
        "are out of scope of the MI355X build, SURVEY.md §2); use the reference's own utils.py for it")
