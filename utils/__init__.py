"""Drop-in for the part of the reference's `utils` package the models' callers import
(`from utils import CombinedLoss, SSIMLoss`, reference train.py:24,153).  The dataset / download
helpers (reference utils/__init__.py:5-6) are host file I/O and out of scope."""
from .losses import CombinedLoss, SSIMLoss

__all__ = ["SSIMLoss", "CombinedLoss"]
