"""Dice loss (drop-in for the reference's loss.py:6-14) on libn3d reduction kernels."""
import torch
import torch.nn as nn

from . import kernels as K
from ._lib import N3DError


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, t, smooth):
        if not p.is_cuda:
            raise N3DError("WeightedDiceLoss: predictions are on %s; no CPU fallback" % p.device)
        if p.dtype != torch.float32 or t.dtype != torch.float32:
            raise N3DError("WeightedDiceLoss: fp32 tensors expected")
        if K._bcv_strides(p) is None:
            p = p.contiguous()
        if K._bcv_strides(t) is None:
            t = t.contiguous()
        loss, sums = K.dice_fwd(p, t, smooth)
        ctx.save_for_backward(p, t, sums)
        ctx.smooth = smooth
        return loss

    @staticmethod
    def backward(ctx, dloss):
        p, t, sums = ctx.saved_tensors
        dp = torch.empty_like(p)  # preserves p's (NDHWC) strides
        if K._bcv_strides(dp) is None:
            dp = torch.empty(p.shape, device=p.device, dtype=p.dtype)
        K.dice_bwd(p, t, ctx.smooth, sums, dloss.contiguous(), dp)
        return dp, None, None


class WeightedDiceLoss(nn.Module):
    """1 - mean_{b,c} (2*sum(p*t)+eps) / (sum(p)+sum(t)+eps), sums over the last three dims."""

    def __init__(self, axis=(-1, -2, -3), smooth=1e-6):
        super().__init__()
        if tuple(sorted(a % 5 for a in axis)) != (2, 3, 4):
            raise NotImplementedError("WeightedDiceLoss: only the spatial axes (-1,-2,-3) are built")
        self.axis = axis
        self.smooth = smooth

    def forward(self, y_pred, y_truth):
        return _DiceFn.apply(y_pred, y_truth, float(self.smooth))
