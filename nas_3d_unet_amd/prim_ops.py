"""MI355X-native drop-in for the reference's ``prim_ops`` module (prim_ops.py:1-174).

Same public surface -- ``OPS``, ``DownOps``, ``UpOps``, ``NormOps``, ``BaseOp``, ``ConvOps``,
``SEConvOp``, ``PoolingOp``, ``IdentityOp`` -- same constructor signatures, same parameter
attribute names and torch-native weight shapes (so reference state_dicts load), same errors.
What is underneath is different: each module owns *launch programs* (programs.py) over the
libn3d HIP kernels; ``forward`` runs them through one autograd node per fused segment.
There is no CPU path: calling an op on a CPU tensor raises.
"""
from __future__ import annotations

from math import ceil

import torch
import torch.nn as nn

from . import programs as P

# registry ------------------------------------------------------------------------------------
# name -> (class tag, kwargs); order of the three lists = alpha column order (prim_ops.py:23-45)
_TABLE = [
    ("identity", "id", {}),
    ("se_conv", "se", {}),
    ("dil_conv", "conv", dict(dilation=2)),
    ("dep_conv", "conv", dict(depthwised=True)),
    ("conv", "conv", {}),
    ("avg_pool", "pool", dict(pool_type="avg")),
    ("max_pool", "pool", dict(pool_type="max")),
    ("down_se_conv", "se", dict(stride=2)),
    ("down_dil_conv", "conv", dict(stride=2, dilation=2)),
    ("down_dep_conv", "conv", dict(stride=2, depthwised=True)),
    ("down_conv", "conv", dict(stride=2)),
    ("up_se_conv", "se", dict(stride=2, transposed=True)),
    ("up_dep_conv", "conv", dict(stride=2, depthwised=True, transposed=True)),
    ("up_conv", "conv", dict(stride=2, transposed=True)),
    ("up_dil_conv", "conv", dict(stride=2, dilation=2, transposed=True)),
]


def _factory(tag, kw):
    def make(c):
        cls = {"id": IdentityOp, "se": SEConvOp, "conv": ConvOps, "pool": PoolingOp}[tag]
        return cls(c, c, **kw)
    return make


OPS = {name: _factory(tag, kw) for name, tag, kw in _TABLE}

DownOps = ["avg_pool", "max_pool", "down_se_conv", "down_dil_conv", "down_dep_conv", "down_conv"]
UpOps = ["up_se_conv", "up_dep_conv", "up_conv", "up_dil_conv"]
NormOps = ["identity", "se_conv", "dil_conv", "dep_conv", "conv"]


def _padding(kernel_size, stride, dilation):
    return max(0, ceil((dilation * (kernel_size - 1) - stride + 1) / 2))


def _pad4(c):
    return (c + 3) // 4 * 4


class _OpTwin:
    """Zero-padded twin of ONE op whose channel counts are not multiples of 4 -- the per-op form of unet.PaddedTwin, for callers that
    compose the registry ops themselves (the reference's unchanged nas.py / searched.py with any `init_n_kernels`, nas.py:13-49,
    searched.py:55-90).  The kernels move channels four at a time: the twin is the same op built on pad4(channels), its padded
    parameter entries are zero, so padded channels hold exactly 0 in every forward tensor; its GroupNorm counts the REAL channels
    (programs.gn_groups -> a negative group count at the C ABI).  Before every forward the op's parameters are embedded into the twin's
    (leading slices), the input is copied into a zero-padded tensor, and the output / the gradients are cut back: index plumbing in
    torch, arithmetic in libn3d.  Slower than the build-side nets' whole-net twin (a copy and a slice per op); same results."""

    def __init__(self, op):
        cin, cout = op._n3d_io
        self.cin, self.cout = cin, cout
        twin = type(op)(_pad4(cin), _pad4(cout), **op._n3d_ctor)
        plist = list(op.parameters())
        self.twin = twin.to(plist[0].device) if plist else twin
        for q in self.twin.parameters():
            q.requires_grad_(True)
        if self.twin.norm is not None:
            self.twin.norm._n3d_real_c = int(op.norm.num_channels)

    def embed(self, op):
        tp = dict(self.twin.named_parameters())
        with torch.no_grad():
            for n, r in op.named_parameters():
                t = tp[n]
                if t.shape == r.shape:
                    t.copy_(r)
                else:
                    t.zero_()
                    t[tuple(slice(0, k) for k in r.shape)].copy_(r)
        self.twin.train(op.training)
        if (op.dropout is None) != (self.twin.dropout is None):
            self.twin.dropout = None if op.dropout is None else nn.Dropout3d(op.dropout.p)


class _PaddedOpFn(torch.autograd.Function):
    """forward / backward of ONE op through its zero-padded twin (see _OpTwin); inputs: (op, twin, need_grad, x, *the op's parameters)"""

    @staticmethod
    def forward(ctx, op, tw, need_grad, xin, *_params):
        from .train import _padded_flags
        tw.embed(op)
        cin, cout = tw.cin, tw.cout
        B, _c, D, H, W = xin.shape
        xp = torch.zeros((B, D, H, W, _pad4(cin)), dtype=xin.dtype, device=xin.device).permute(0, 4, 1, 2, 3)
        xp[:, :cin].copy_(xin)
        with _padded_flags(), torch.set_grad_enabled(need_grad):
            xi = xp.requires_grad_(need_grad and xin.requires_grad)
            out = BaseOp._run_segments(tw.twin, xi)
        if need_grad:
            ctx.saved = (op, tw, xi, out)
        return out.detach()[:, :cout]

    @staticmethod
    def backward(ctx, dout):
        from .train import _padded_flags
        op, tw, xi, out = ctx.saved
        tp = dict(tw.twin.named_parameters())
        reals = list(op.named_parameters())
        dp = torch.zeros_like(out)
        dp[:, :tw.cout].copy_(dout)
        wanted = ([xi] if xi.requires_grad else []) + [tp[n] for n, _ in reals]
        with _padded_flags():
            gs = list(torch.autograd.grad([out], wanted, [dp], allow_unused=True))
        gx = gs.pop(0)[:, :tw.cin] if xi.requires_grad else None
        gpar = [None if g is None else g[tuple(slice(0, k) for k in r.shape)] for g, (_, r) in zip(gs, reals)]
        return (None, None, None, gx, *gpar)


def _run_padded_op(op, x):
    tw = op.__dict__.get("_n3d_optwin")
    plist = list(op.parameters())
    if tw is None or (plist and next(tw.twin.parameters()).device != plist[0].device):
        tw = op.__dict__["_n3d_optwin"] = _OpTwin(op)      # (kept out of the module tree: the state dict stays the reference's)
    need_grad = torch.is_grad_enabled() and (x.requires_grad or any(q.requires_grad for q in plist))
    return _PaddedOpFn.apply(op, tw, need_grad, x, *plist)


class BaseOp(nn.Module):
    """Sequences weight / norm / act by the ``ops_order`` string (prim_ops.py:48-83)."""

    def __init__(self, in_channels, out_channels, dropout_rate=0, ops_order="weight_norm_act"):
        super().__init__()
        self.ops_list = ops_order.split("_")
        if "norm" in self.ops_list:
            self.norm = nn.GroupNorm(P.group_count(out_channels), out_channels)
        else:
            self.norm = None
        self.activation = nn.ReLU() if "act" in self.ops_list else None
        self.dropout = nn.Dropout3d(dropout_rate) if dropout_rate > 0 else None
        self._segments = None
        # channel counts (the kernels move channels four at a time; odd counts run through a zero-padded twin of the op: _OpTwin).  A
        # dense conv takes any number of INPUT channels as it is (copies), and any output count when no norm follows (the head)
        self._n3d_io = (in_channels, out_channels)
        self._n3d_ctor = {}

    def __setattr__(self, name, value):
        # the launch programs are built once from (ops_list, norm, dropout, weight modules): re-assigning one of them
        # (e.g. `op.dropout = None`, as the reference's users do to switch the head's Dropout3d off) must rebuild them
        if name in ("norm", "dropout", "activation", "conv", "depth_conv", "point_conv", "fc", "ops_list") and "_segments" in self.__dict__:
            self.__dict__["_segments"] = None
            P.PLAN_VERSION[0] += 1    # cached cell / net plans (fused.searched_plan / supernet_plan / net_plan) hold the old segments
        super().__setattr__(name, value)

    # subclasses describe their weight op ------------------------------------------------------
    def _weight_program(self):
        raise NotImplementedError

    def _se_epilogue(self):
        return None

    def _build_segments(self):
        """Split ops_list into canonical [act] weight [norm] [act] segments."""
        for tok in self.ops_list:
            if tok not in ("weight", "norm", "act"):
                raise Warning("Unrecognized op: %s" % tok)
        segs = []
        cur = None

        def flush():
            nonlocal cur
            if cur is not None:
                segs.append(P.Segment(cur.get("weight"), cur.get("norm"), cur.get("relu_in", False),
                                      cur.get("relu_out", False), cur.get("se"), cur.get("dropout")))
            cur = None

        toks = self.ops_list
        for i, tok in enumerate(toks):
            if tok == "act":
                nxt = toks[i + 1] if i + 1 < len(toks) else None
                if cur is None and nxt == "weight":
                    cur = {"relu_in": True, "stage": 0}
                else:
                    if cur is None:
                        cur = {"stage": 0}
                    if cur["stage"] >= 3:
                        flush()
                        cur = {"stage": 0}
                    cur["relu_out"] = True
                    cur["stage"] = 3
                    flush()
            elif tok == "weight":
                if cur is not None and cur["stage"] >= 1:
                    flush()
                if cur is None:
                    cur = {"stage": 0}
                cur["weight"] = self._weight_program()
                cur["se"] = self._se_epilogue()
                cur["dropout"] = self.dropout
                cur["stage"] = 1
            elif tok == "norm":
                if self.norm is None:
                    continue
                if cur is not None and cur["stage"] >= 2:
                    flush()
                if cur is None:
                    cur = {"stage": 0}
                if cur.get("se") is not None:
                    flush()  # an SE scale epilogue cannot also carry a norm
                    cur = {"stage": 0}
                cur["norm"] = self.norm
                cur["stage"] = 2
        flush()
        return segs

    def _odd_channels(self):
        cin, cout = self._n3d_io
        dense = isinstance(self, ConvOps) and not self.depthwised
        return (cout % 4 != 0 and not (dense and self.norm is None)) or (cin % 4 != 0 and not dense)

    def _run_segments(self, x):
        if self._segments is None:
            self._segments = self._build_segments()
        for seg in self._segments:
            x = P.run_segment(seg, x, self.training)
        return x

    def forward(self, x):
        if self._odd_channels():
            return _run_padded_op(self, x)
        return self._run_segments(x)


class ConvOps(BaseOp):
    """Conv3d / ConvTranspose3d, optionally depthwise + pointwise (prim_ops.py:85-117)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, transposed=False,
                 depthwised=False, dropout_rate=0, ops_order="weight_norm_act"):
        super().__init__(in_channels, out_channels, dropout_rate, ops_order)
        self._n3d_ctor = dict(kernel_size=kernel_size, stride=stride, dilation=dilation, transposed=transposed, depthwised=depthwised,
                              dropout_rate=dropout_rate, ops_order=ops_order)
        self.depthwised = depthwised
        self._k, self._stride, self._transposed = kernel_size, stride, transposed
        self._pad = _padding(kernel_size, stride, dilation)
        opad = 0 if stride == 1 else 1
        if depthwised:
            # the depthwise stage never receives `dilation` (prim_ops.py:95-97,105-106)
            self._dil = 1
            if transposed:
                self.depth_conv = nn.ConvTranspose3d(in_channels, in_channels, kernel_size, stride=stride,
                                                     padding=self._pad, groups=in_channels, output_padding=opad)
            else:
                self.depth_conv = nn.Conv3d(in_channels, in_channels, kernel_size, stride=stride, padding=self._pad,
                                            groups=in_channels)
            self.point_conv = nn.Conv3d(in_channels, out_channels, kernel_size=1)
        else:
            self._dil = dilation
            if transposed:
                self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, padding=self._pad,
                                               dilation=dilation, output_padding=opad)
            else:
                self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=self._pad,
                                      dilation=dilation)

    def _weight_program(self):
        if self.depthwised:
            if self._k != 3:
                raise NotImplementedError("depthwise ConvOps: only kernel_size=3 is built")
            return P.DepthSepW(self.depth_conv, self.point_conv, self._stride, self._pad, self._transposed)
        return P.DenseConvW(self.conv, self._k, self._stride, self._dil, self._pad, self._transposed)

    def weight_call(self, x):
        seg = P.Segment(self._weight_program())
        return P.run_segment(seg, x, False)


class SEConvOp(BaseOp):
    """Squeeze-and-excitation gate, then (stride 2 only) a 3x3x3 conv (prim_ops.py:119-153)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, transposed=False,
                 dropout_rate=0, ops_order="weight_norm"):
        super().__init__(in_channels, out_channels, dropout_rate, ops_order=ops_order if stride > 1 else "weight")
        self._n3d_ctor = dict(kernel_size=kernel_size, stride=stride, dilation=dilation, transposed=transposed, dropout_rate=dropout_rate,
                              ops_order=ops_order)
        self.stride = stride
        self._transposed = transposed
        self._k = kernel_size
        self._pad = _padding(kernel_size, stride, dilation)
        self.avg_pool = nn.AdaptiveAvgPool3d(1)
        self.fc = nn.Sequential(nn.Linear(in_channels, 1), nn.ReLU(), nn.Linear(1, out_channels), nn.Sigmoid())
        if stride > 1:
            if transposed:
                self.conv = nn.ConvTranspose3d(in_channels, out_channels, kernel_size, stride=stride, padding=self._pad,
                                               output_padding=0 if stride == 1 else 1)
            else:
                self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=self._pad)

    def _weight_program(self):
        if self.stride >= 2:
            if self._k != 3:
                raise NotImplementedError("SEConvOp: only kernel_size=3 is built")
            return P.SEConvW(self.fc, self.conv, self.stride, self._pad, self._transposed)
        return P.IdentityW()

    def _se_epilogue(self):
        return P.SEGate(self.fc) if self.stride < 2 else None

    def weight_call(self, x):
        seg = P.Segment(self._weight_program(), se_gate=self._se_epilogue())
        return P.run_segment(seg, x, False)


class PoolingOp(BaseOp):
    """2x2x2 average / max pooling (prim_ops.py:155-168)."""

    def __init__(self, in_channels, out_channels, pool_type, kernel_size=2, stride=2, ops_order="weight"):
        super().__init__(in_channels, out_channels, ops_order=ops_order)
        self._n3d_ctor = dict(pool_type=pool_type, kernel_size=kernel_size, stride=stride, ops_order=ops_order)
        if pool_type == "avg":
            self.pool = nn.AvgPool3d(kernel_size, stride=stride)
        elif pool_type == "max":
            self.pool = nn.MaxPool3d(kernel_size, stride=stride)
        else:
            raise NotImplementedError
        if kernel_size != 2 or stride != 2:
            raise NotImplementedError("PoolingOp: only kernel_size=2, stride=2 is built")
        self._is_max = pool_type == "max"

    def _weight_program(self):
        return P.PoolW(self._is_max)

    def weight_call(self, x):
        return P.run_segment(P.Segment(self._weight_program()), x, False)


class IdentityOp(BaseOp):
    """Not a pure identity: GroupNorm -> ReLU with its own affine (prim_ops.py:170-174)."""

    def __init__(self, in_channels, out_channels, ops_order="weight_norm_act"):
        super().__init__(in_channels, out_channels, ops_order=ops_order)
        self._n3d_ctor = dict(ops_order=ops_order)

    def _weight_program(self):
        return P.IdentityW()

    def weight_call(self, x):
        return x
