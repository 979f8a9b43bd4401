"""MI355X-native (gfx950) implementation of the SIF-CNN-SR hot path.

Drop-in surface (SURVEY.md §8 b):
  model.ModelB_2                      <- reference model.py:533
  sif_ops.downscale_LST_SR_to_LR      <- reference utils.py:1671
  sif_ops.get_output_ftm              <- reference utils.py:1833
  sif_ops.sobel_bank / huber_loss / sif_loss  (fused forms of the train scripts' loss block)
  optim.FlatAdam                      <- torch.optim.Adam, train_model_B_gradFTM.py:453
  dataset.ModisDatasetB               <- reference dataset.py:29 (synthetic drop-in, same __getitem__)
  train.train_step / predict.predict_tiles  <- train_model_B_gradFTM.py:86-121 / predict.py:84-103
  pipeline.prepare_tiles / granule_to_tiles / tiles_to_granule, predict.predict_granule  <- dataset.py:134-142, predict.py:84-103
  metrics.psnr_skimage / ssim_skimage  <- utils.py:548-578 (on device)
  fourier.fft2_magnitude / attenuation_spectra / get_FRR / get_FRO / get_FRU  <- compare_methods.py:312-324, utils.py:598-662

The directory name is the repository's mandated package name (it contains '-', so it is imported
through ``importlib`` or the ``sifsr`` alias: ``import sifsr`` at the repo root loads this package
and registers ``sifsr`` / ``sifsr.<submodule>`` as aliases of the same module objects).
"""
import importlib
import sys

_SUBMODULES = ("_lib", "model", "sif_ops", "optim", "dataset", "distributed", "train", "pipeline", "metrics", "fourier", "predict")
for _m in _SUBMODULES:
    importlib.import_module(__name__ + "." + _m)

sys.modules["sifsr"] = sys.modules[__name__]
for _m in _SUBMODULES:
    sys.modules.setdefault("sifsr." + _m, sys.modules[__name__ + "." + _m])

from .model import ModelB_2  # noqa: E402,F401
from .sif_ops import downscale_LST_SR_to_LR, get_output_ftm, sobel_bank, huber_loss, sif_loss  # noqa: E402,F401
from .optim import FlatAdam  # noqa: E402,F401
from .dataset import ModisDatasetB  # noqa: E402,F401
from ._lib import SifsrError  # noqa: E402,F401
