"""Drop-in for the part of the reference's `utils` package the models' callers import
(`from utils import CombinedLoss, SSIMLoss`, reference train.py:24,153).  The dataset / download
helpers (reference utils/__init__.py:5-6) are host file I/O and out of scope: when this package
is placed over the reference's `utils/` (INTEGRATION.md option A keeps the reference's own `__init__.py`, `dataset.py`,
`download_data.py` and replaces `losses.py` only) they keep coming from the reference's files; when this file is used
instead, it forwards to those modules if they sit next to it, so `from utils import MVTecDataset` (evaluate.py:23,
train.py:24) works in both layouts."""
import importlib
from pathlib import Path

from .losses import CombinedLoss, SSIMLoss

__all__ = ["SSIMLoss", "CombinedLoss"]

for _mod, _names in (("dataset", ("MVTecDataset", "get_dataloaders")),
                     ("download_data", ("create_synthetic_test_data", "download_with_kagglehub"))):
    if (Path(__file__).with_name(_mod + ".py")).exists():
        _m = importlib.import_module("." + _mod, __name__)
        for _n in _names:
            if hasattr(_m, _n):
                globals()[_n] = getattr(_m, _n)
                __all__.append(_n)
