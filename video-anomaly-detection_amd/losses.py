"""`SSIMLoss` / `CombinedLoss` (reference utils/losses.py:14-121).

On GPU fp32 tensors both directions are hand-written HIP (csrc/ssim.hip), wrapped in one `torch.autograd.Function`:
* forward `vad_ssim_mse`: both inputs are read once, the five blurred maps and the SSIM map never reach memory, one launch
  returns 1-SSIM, MSE and their combination;
* backward `vad_ssim_mse_backward` (what reference train.py:41-46 needs when it back-propagates through the criterion):
  the analytic adjoint in two fused passes, gradient with respect to the prediction.
The stock torch composition remains for CPU tensors and for the unusual case where the TARGET needs a gradient.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip

_C1, _C2 = 0.01 ** 2, 0.03 ** 2   # reference utils/losses.py:82-83
_SIGMA = 1.5                      # reference utils/losses.py:37


def _gaussian_window(size: int, channels: int) -> torch.Tensor:
    offs = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-offs ** 2 / (2 * _SIGMA ** 2))
    g = g / g.sum()
    return torch.outer(g, g).expand(channels, 1, size, size).contiguous()


def _hip_eligible(pred: torch.Tensor, target: torch.Tensor) -> bool:
    """GPU tensors, fp32 [B,C,H,W], same shape, and no gradient wanted for the target: the fused HIP passes apply."""
    target_grad = torch.is_grad_enabled() and target.requires_grad
    return (pred.is_cuda and target.is_cuda and not target_grad and pred.dim() == 4 and pred.shape == target.shape
            and pred.dtype == torch.float32 and target.dtype == torch.float32)


def _hip_criteria(pred: torch.Tensor, target: torch.Tensor, window_size: int, alpha: float) -> torch.Tensor:
    """-> device tensor [3] = (1 - mean SSIM, MSE, (1-alpha)*MSE + alpha*(1 - mean SSIM)); raises VadError if the
    library is missing or the window is unsupported (there is no silent fallback on this path)."""
    l = hip.lib()
    pred, target = pred.contiguous(), target.contiguous()
    b, c, h, w = pred.shape
    n = l.vad_ssim_workspace_floats(b * c, h, w)
    if n == 0:
        raise hip.VadError(f"SSIM: unsupported shape {tuple(pred.shape)}")
    ws = torch.empty(n, dtype=torch.float32, device=pred.device)
    out = torch.empty(3, dtype=torch.float32, device=pred.device)
    with torch.cuda.device(pred.device):
        hip.check(l.vad_ssim_mse(pred.data_ptr(), target.data_ptr(), b * c, h, w, int(window_size), float(alpha),
                                 ws.data_ptr(), out.data_ptr(), hip.current_stream()), "vad_ssim_mse")
    hip.calls["ssim"] = hip.calls.get("ssim", 0) + 1
    return out


class _HipCriterion(torch.autograd.Function):
    """(pred, target) -> out3[which]; backward = vad_ssim_mse_backward with the matching alpha (which 0: SSIM only -> 1,
    which 2: the combination -> alpha)."""

    @staticmethod
    def forward(ctx, pred, target, window_size, alpha, which):
        out = _hip_criteria(pred.detach(), target.detach(), window_size, alpha)
        ctx.save_for_backward(pred.detach(), target.detach())
        ctx.window_size, ctx.alpha = int(window_size), (1.0 if which == 0 else float(alpha))
        return out[which].clone()

    @staticmethod
    def backward(ctx, grad_out):
        pred, target = ctx.saved_tensors
        pred, target = pred.contiguous(), target.contiguous()
        l = hip.lib()
        b, c, h, w = pred.shape
        ws = torch.empty(l.vad_ssim_grad_workspace_floats(b * c, h, w), dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred)
        go = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        with torch.cuda.device(pred.device):
            hip.check(l.vad_ssim_mse_backward(pred.data_ptr(), target.data_ptr(), b * c, h, w, ctx.window_size, ctx.alpha,
                                              go.data_ptr(), ws.data_ptr(), grad.data_ptr(), hip.current_stream()),
                      "vad_ssim_mse_backward")
        hip.calls["ssim_backward"] = hip.calls.get("ssim_backward", 0) + 1
        return grad, None, None, None, None


def _hip_loss(pred, target, window_size, alpha, which):
    if torch.is_grad_enabled() and pred.requires_grad:
        return _HipCriterion.apply(pred, target, window_size, alpha, which)
    return _hip_criteria(pred, target, window_size, alpha)[which]


class SSIMLoss(nn.Module):
    """1 - mean SSIM with an 11x11 sigma-1.5 Gaussian window (reference utils/losses.py:14-93)."""

    def __init__(self, window_size: int = 11, channels: int = 3):
        super().__init__()
        self.window_size = window_size
        self.channels = channels
        self.window = _gaussian_window(window_size, channels)

    def forward(self, pred: torch.Tensor, target: torch.Tensor):
        if _hip_eligible(pred, target):
            return _hip_loss(pred, target, self.window_size, 1.0, 0)
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
        if _hip_eligible(pred, target):
            return _hip_loss(pred, target, self.ssim.window_size, float(self.alpha), 2)
        return (1 - self.alpha) * self.mse(pred, target) + self.alpha * self.ssim(pred, target)
