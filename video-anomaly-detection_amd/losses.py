"""Training criteria kept for API compatibility (reference utils/losses.py:14-121).

`SSIMLoss` / `CombinedLoss` are used only by the reference's train.py (:149-158), never by an
evaluate script, so they are not part of the HIP scoring path.
They are stated here with plain torch ops so `from utils import CombinedLoss, SSIMLoss` keeps working.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

_C1, _C2 = 0.01 ** 2, 0.03 ** 2   # reference utils/losses.py:82-83
_SIGMA = 1.5                      # reference utils/losses.py:37


def _gaussian_window(size: int, channels: int) -> torch.Tensor:
    offs = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-offs ** 2 / (2 * _SIGMA ** 2))
    g = g / g.sum()
    return torch.outer(g, g).expand(channels, 1, size, size).contiguous()


class SSIMLoss(nn.Module):
    """1 - mean SSIM with an 11x11 sigma-1.5 Gaussian window (reference utils/losses.py:14-93)."""

    def __init__(self, window_size: int = 11, channels: int = 3):
        super().__init__()
        self.window_size = window_size
        self.channels = channels
        self.window = _gaussian_window(window_size, channels)

    def forward(self, pred: torch.Tensor, target: torch.Tensor):
        if self.window.device != pred.device:
            self.window = self.window.to(pred.device)
        groups, pad, win = pred.shape[1], self.window_size // 2, self.window

        def blur(t):
            return F.conv2d(t, win, padding=pad, groups=groups)

        mu_p, mu_t = blur(pred), blur(target)
        var_p = blur(pred * pred) - mu_p * mu_p
        var_t = blur(target * target) - mu_t * mu_t
        cov = blur(pred * target) - mu_p * mu_t
        ssim = ((2 * mu_p * mu_t + _C1) * (2 * cov + _C2)) / ((mu_p * mu_p + mu_t * mu_t + _C1) * (var_p + var_t + _C2))
        return 1 - ssim.mean()


class CombinedLoss(nn.Module):
    """(1 - alpha) * MSE + alpha * (1 - SSIM) (reference utils/losses.py:96-121)."""

    def __init__(self, alpha: float = 0.5, window_size: int = 11):
        super().__init__()
        self.alpha = alpha
        self.mse = nn.MSELoss()
        self.ssim = SSIMLoss(window_size=window_size)

    def forward(self, pred: torch.Tensor, target: torch.Tensor):
        return (1 - self.alpha) * self.mse(pred, target) + self.alpha * self.ssim(pred, target)
