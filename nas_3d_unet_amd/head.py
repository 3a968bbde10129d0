"""Network head as fused launches (nas.py:50-52 / searched.py:91-93: `nn.Sequential(ConvOps(c, n_out, kernel_size=1,
dropout_rate=p, ops_order='weight'), nn.Sigmoid())`, and loss.py:12-14 behind it).

`run(head, x)` = Dropout3d -> 1x1x1 conv -> sigmoid in ONE kernel (n3d_head_fwd) with a one-pass backward (n3d_head_bwd);
`run_loss(head, x, t, smooth)` additionally forms the Dice sums in the forward pass and the Dice gradient inside the
backward pass, so the trainers' step has no separate sigmoid / Dice passes over (B, n_out, D, H, W).
The modules (and their state-dict names `last_conv.0.conv.*`) stay the reference's; only what runs underneath differs.
Shapes the fused kernels do not take (see include/n3d.h) fall back to the op-by-op path.
"""
from __future__ import annotations

import torch

from . import kernels as K
from . import programs as P

KEEP_LOGITS = False   # tests: keep the pre-sigmoid activations of the last run() / run_loss() call in `last_logits`
last_logits = None


def feat_shape(x):
    """(B, C) of a feature map: a 5-D (B, C, D, H, W) tensor, or a node-planar 6-D (nn, B, cn, D, H, W) one (kernels.Planar)"""
    return (x.shape[1], x.shape[0] * x.shape[2]) if x.dim() == 6 else (x.shape[0], x.shape[1])


def fusable(head, x):
    op = head[0]
    return (len(head) == 2 and isinstance(head[1], torch.nn.Sigmoid) and getattr(op, "ops_list", None) == ["weight"]
            and not getattr(op, "depthwised", True) and op._k == 1 and op._stride == 1 and not op._transposed
            and feat_shape(x)[1] in (4, 8, 12, 16, 24, 32) and op.conv.weight.shape[0] <= 4)


def _gate(op, x):
    B, Cc = feat_shape(x)
    return P.draw_gate(op.dropout, op.training, B, Cc, x.device)


def _feat_view(x):
    return K.as_planar(x, "head input") if x.dim() == 6 else K.as_view(x, "head input", bf16_ok=True)


def _feat_like(xv):
    """gradient buffer of the head input's layout"""
    if isinstance(xv, K.Planar):
        return K.empty_planar(xv.nn, xv.B, xv.cn, xv.D, xv.H, xv.W, xv.t.device, xv.t.dtype)
    return K.as_view(K.empty_ndhwc(xv.B, xv.C, xv.D, xv.H, xv.W, xv.t.device, xv.t.dtype), bf16_ok=True)


class HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate, x, w, b):
        global last_logits
        xv = _feat_view(x)
        p, logits, _, _ = K.head_fwd(xv, w, b, gate, want_logits=KEEP_LOGITS)
        if KEEP_LOGITS:
            last_logits = logits
        ctx.xv, ctx.gate, ctx.w, ctx.b = xv, gate, w, b
        return p

    @staticmethod
    def backward(ctx, dp):
        xv, w, b = ctx.xv, ctx.w, ctx.b
        if K._bcv_strides(dp) is None:
            dp = dp.contiguous()
        dx = _feat_like(xv)
        dw, db = K.grad_target(w), K.grad_target(b)
        K.head_bwd(xv, w, b, ctx.gate, dx, dw, db, dp=dp)
        inplace = getattr(w, "_n3d_grad", None) is not None
        ctx.xv = None
        return None, dx.t, None if inplace else dw, None if inplace else db


class HeadDiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate, smooth, x, t, w, b):
        global last_logits
        xv = _feat_view(x)
        if K._bcv_strides(t) is None:
            t = t.contiguous()
        # (the trainers' autograd-free pipeline sets `skip_p`: a step needs the loss, and the backward pass recomputes p from x)
        want_p = KEEP_LOGITS or not getattr(ctx, "skip_p", False)
        p, logits, sums, loss = K.head_fwd(xv, w, b, gate, t, smooth, want_logits=KEEP_LOGITS, want_p=want_p)
        if KEEP_LOGITS:
            last_logits = logits
        ctx.xv, ctx.gate, ctx.w, ctx.b, ctx.t, ctx.sums, ctx.smooth = xv, gate, w, b, t, sums, smooth
        if p is not None:
            ctx.mark_non_differentiable(p)
        return loss, p

    @staticmethod
    def backward(ctx, dloss, _dp):
        xv, w, b = ctx.xv, ctx.w, ctx.b
        dx = _feat_like(xv)
        dw, db = K.grad_target(w), K.grad_target(b)
        K.head_bwd(xv, w, b, ctx.gate, dx, dw, db, t=ctx.t, sums=ctx.sums, dloss=dloss.contiguous(), smooth=ctx.smooth)
        inplace = getattr(w, "_n3d_grad", None) is not None
        ctx.xv = ctx.t = None
        return None, None, dx.t, None, None if inplace else dw, None if inplace else db


def run(head, x):
    """probabilities of the head on features x"""
    if not fusable(head, x):
        return head(x)
    op = head[0]
    return HeadFn.apply(_gate(op, x), x, op.conv.weight, op.conv.bias)


def run_loss(head, x, t, smooth=1e-6):
    """(Dice loss, probabilities) of the head on features x against the target t.  The probabilities are returned for
    inspection only and carry NO gradient on either path (the fused kernels form the Dice gradient inside the head's backward
    pass): a caller who wants another loss on them uses run() and differentiates through that."""
    fused = fusable(head, x)
    byte_ok = fused and t.dtype == torch.uint8 and head[0].conv.weight.shape[0] == 3      # {0, 1} bytes: the three-channel head kernels read them as they are
    if not fused or not (t.dtype == torch.float32 or byte_ok):
        from .loss import WeightedDiceLoss
        p = head(x)
        return WeightedDiceLoss(smooth=smooth)(p, t if t.dtype == torch.float32 else t.to(torch.float32)), p.detach()
    op = head[0]
    return HeadDiceFn.apply(_gate(op, x), float(smooth), x, t, op.conv.weight, op.conv.bias)
