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

    def forward(self, x):
        if self._segments is None:
            self._segments = self._build_segments()
        for seg in self._segments:
            x = P.run_segment(seg, x, self.training)
        return x


class ConvOps(BaseOp):
    """Conv3d / ConvTranspose3d, optionally depthwise + pointwise (prim_ops.py:85-117)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, transposed=False,
                 depthwised=False, dropout_rate=0, ops_order="weight_norm_act"):
        super().__init__(in_channels, out_channels, dropout_rate, ops_order)
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

    def _weight_program(self):
        return P.IdentityW()

    def weight_call(self, x):
        return x
