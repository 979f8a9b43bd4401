"""``from dataset import ModisDatasetB`` (train_model_B_gradFTM.py:28) / ``ModisDatasetB_scale_invariance``
(train_model_B_scale_invariance.py:31) resolved to the synthetic drop-ins with the reference's constructor signature,
``.stats`` and ``__getitem__`` shapes (dataset.py:50,81,101-142,217-263).  The reference's own class reads GeoTIFF
pairs through GDAL / OpenCV from ``data/ModisDatasetB.csv``, none of which ships."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import sifsr  # noqa: E402,F401
from sifsr.dataset import DEFAULT_STATS, ModisDatasetB, ModisDatasetB_scale_invariance  # noqa: E402,F401
