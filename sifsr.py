"""``import sifsr`` -> the MI355X package whose directory name is not a Python identifier."""
import importlib
import sys

_pkg = importlib.import_module(
    "land-surface-temperature-super-resolution-with-a-scale-invariance-free-neural-approach_amd")
sys.modules[__name__] = _pkg
