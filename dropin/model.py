"""``from model import ModelB_2`` (train_model_B_gradFTM.py:27, predict.py:9, model_perf_aster_formatds.py:46) resolved to
the MI355X implementation: put this directory on ``PYTHONPATH`` ahead of the reference's own ``model.py`` and the
reference's scripts construct the gfx950 model with zero edits.

``ModelB_2`` here is a subclass whose ``__module__`` is ``model``, so ``torch.save(model)`` (utils.py:826) pickles it as
``model.ModelB_2`` -- the name the reference's own full-module pickles carry -- and ``torch.load`` of such a file
resolves through this shim as long as it is importable as ``model``.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import sifsr  # noqa: E402  (the repository-root alias of the hyphenated package directory)
from sifsr.model import ModelB_2 as _ModelB_2  # noqa: E402


class ModelB_2(_ModelB_2):
    """model.py:533 -- same constructor signature, attributes and 104-key state_dict; forward on MI355X."""


ModelB_2.__module__ = __name__
SifsrError = sifsr.SifsrError
