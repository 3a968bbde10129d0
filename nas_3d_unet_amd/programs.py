"""Launch programs: how one reference op (prim_ops.BaseOp subclasses) maps onto libn3d kernels.

A *segment* is the canonical fused sequence  [ReLU-on-load] -> [weight op] -> [GroupNorm] -> [ReLU]
followed by an optional weighted accumulation into a destination (MixedOp weight / node sum).
BaseOp.forward (prim_ops.py:68-83) walks `ops_order`; every order used by the reference
('weight_norm_act', 'act_weight_norm', 'weight_norm', 'weight') is a single segment, any other
order is split into several.  Forward and backward of a segment are plain kernel launch
sequences (`seg_forward` / `seg_backward`) so that the op-level, MixedOp-level and cell-level
autograd Functions can share them.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import kernels as K
from ._lib import ACCUMULATE, RELU, RELU_IN, N3DError
from ._lib import F32 as _F32


# bumped whenever an op's launch program is invalidated (prim_ops.BaseOp.__setattr__): cached cell / net plans compare it
PLAN_VERSION = [0]


def group_count(c):
    """GroupNorm group rule of the reference (prim_ops.py:57)."""
    return 1 if c % 16 != 0 else c // 16


def gn_groups(norm, C):
    """group-count argument of the GroupNorm kernels for a tensor of C channels behind `norm`.  A norm of a zero-padded twin net
    (unet.PaddedTwin: channel counts that are not multiples of 4) carries the REAL channel count: -real = one group of `real` channels
    inside the C stored ones (include/n3d.h, "padded channels"; real counts that need padding are never multiples of 16: one group)"""
    real = getattr(norm, "_n3d_real_c", None)
    if real is not None and real != C:
        return -int(real)
    return group_count(C)


# conv-bias gradients of a conv that feeds a GroupNorm come analytically out of the GroupNorm backward sums (N * Bc + ...); the padded
# twin switches that off (its GroupNorm kernels run with a scaled element count, the bias gradient needs the true one)
ANALYTIC_CONV_BIAS = True


class Saved:
    """bag of tensors / scalars kept between forward and backward of one segment"""
    pass


# =================================================================================================
# weight programs
# =================================================================================================
class WeightProgram:
    produces_stats = False

    def params(self):
        return []

    def norm_fed_bias(self):
        """bias Parameter of the conv whose output is the segment's raw tensor (its gradient can then be
        derived from the GroupNorm backward sums instead of a reduction over the tensor), or None"""
        return None

    def out_shape(self, x):
        raise NotImplementedError


class IdentityW(WeightProgram):
    """IdentityOp.weight_call (prim_ops.py:173-174)"""

    def out_shape(self, x):
        return (x.B, x.C, x.D, x.H, x.W)

    def fwd(self, x, relu_in, gate, want_stats):
        if relu_in or gate is not None:
            raise N3DError("identity weight op with act-before-weight / dropout is not supported")
        return x, None, 0, None

    def bwd(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias=False):
        if not need_dx:
            return None, []
        if dx_out is None:
            return draw.t, []
        K.affine_act(draw, None, None, None, dx_out, ACCUMULATE if dx_acc else 0)
        return dx_out.t, []


class PoolW(WeightProgram):
    """PoolingOp.weight_call: AvgPool3d(2,2) / MaxPool3d(2,2) (prim_ops.py:160-168)"""

    def __init__(self, is_max):
        self.is_max = is_max

    def out_shape(self, x):
        return (x.B, x.C, x.D // 2, x.H // 2, x.W // 2)

    def fwd(self, x, relu_in, gate, want_stats):
        if relu_in or gate is not None:
            raise N3DError("pooling with act-before-weight / dropout is not supported")
        if x.D % 2 or x.H % 2 or x.W % 2:
            raise N3DError("pool2: spatial dims must be even, got %s" % ((x.D, x.H, x.W),))
        y = K.as_view(K.empty_ndhwc(x.B, x.C, x.D // 2, x.H // 2, x.W // 2, x.t.device, x.t.dtype))
        K.pool2_fwd(x, y, self.is_max)
        s = Saved()
        s.x = x
        return y, None, 0, s

    def bwd(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias=False, scale=None):
        """scale: device scalar pointer multiplying the gradient (the primitive's MixedOp weight) or None"""
        if not need_dx:
            return None, []
        x = saved.x
        if dx_out is None:
            dx_out = K.like(x)
            dx_acc = False
        K.pool2_bwd(draw, x, dx_out, self.is_max, dx_acc, scale)
        return dx_out.t, []


def _materialise_pre(x, relu_in, gate):
    """u = relu?(x) * gate  as a real tensor (rare orders only)."""
    u = K.like(x)
    K.affine_act(x, gate, None, None, u, RELU if (relu_in and gate is None) else 0)
    if relu_in and gate is not None:
        # relu(x)*gate with a possibly negative gate: do it in two passes
        K.affine_act(x, None, None, None, u, RELU)
        K.affine_act(u, gate, None, None, u, 0)
    return u


def _pre_backward(x, relu_in, gate, du, dx_out, dx_acc):
    """dx (+)= du * gate * [x > 0]"""
    if dx_out is None:
        dx_out = K.like(x)
        dx_acc = False
    fl = (RELU if relu_in else 0) | (ACCUMULATE if dx_acc else 0)
    K.affine_act_bwd_apply(du, x, None, None, gate, None, None, dx_out, fl)
    return dx_out


class DenseConvW(WeightProgram):
    """nn.Conv3d / nn.ConvTranspose3d of ConvOps / SEConvOp (prim_ops.py:100-102,109-110,142-147)."""
    produces_stats = True

    def __init__(self, conv_module, k, stride, dil, pad, transposed):
        self.m = conv_module
        self.k, self.stride, self.dil, self.pad, self.transposed = k, stride, dil, pad, transposed

    def params(self):
        return [self.m.weight, self.m.bias]

    def norm_fed_bias(self):
        return self.m.bias

    def geom(self, x, w=None):
        w = self.m.weight if w is None else w
        if self.transposed:
            cin_t, cout_t = w.shape[0], w.shape[1]
            if x.C != cin_t:
                raise N3DError("conv_transpose: input has %d channels, weight expects %d" % (x.C, cin_t))
            opad = 0 if self.stride == 1 else 1
            def od(i):
                return (i - 1) * self.stride - 2 * self.pad + self.dil * (self.k - 1) + opad + 1
            return K.conv_geom(x.B, od(x.D), od(x.H), od(x.W), cout_t, cin_t, self.k, self.stride, self.dil, self.pad)
        cout, cin = w.shape[0], w.shape[1]
        if x.C != cin:
            raise N3DError("conv: input has %d channels, weight expects %d" % (x.C, cin))
        return K.conv_geom(x.B, x.D, x.H, x.W, cin, cout, self.k, self.stride, self.dil, self.pad)

    def out_shape(self, x):
        g = self.geom(x)
        return (g.B, g.Ci, g.Di, g.Hi, g.Wi) if self.transposed else (g.B, g.Co, g.Do, g.Ho, g.Wo)

    def fwd_prepare(self, x, relu_in, gate, want_stats):
        """everything of fwd() except the conv launch: returns (call tuple for K.conv_fwd2, (y, stats, rows, saved))"""
        s = Saved()
        s.pre = None
        if self.transposed and (relu_in or gate is not None):
            s.pre = (x, relu_in, gate)
            x = _materialise_pre(x, relu_in, gate)
            relu_in, gate = False, None
        w = self.m.weight
        s.wpad = None
        if x.C % 4 != 0 and not self.transposed:
            # 1-3 input modalities at the stems (or any conv input whose channel count is not a multiple of 4): the kernels
            # read channels four at a time, so input and weight get zero channels up to the next multiple (copies only)
            s.cin = x.C
            x, w = self._pad_input_channels(x)
            s.wpad = w
        g = self.geom(x, w)
        shp = (g.B, g.Ci, g.Di, g.Hi, g.Wi) if self.transposed else (g.B, g.Co, g.Do, g.Ho, g.Wo)
        y = K.as_view(K.empty_ndhwc(*shp, x.t.device))
        stats, rows = None, 0
        if want_stats:
            rows = K.conv_stats_rows(g, self.transposed, 0, x, y)
            if rows > 0:  # 0: this shape's kernel cannot emit statistics -> seg_forward runs n3d_channel_stats
                stats = torch.empty((x.B, rows, shp[1], 2), dtype=torch.float64, device=x.t.device)
        s.x, s.g, s.relu_in, s.gate = x, g, relu_in, gate
        call = (g, x, w, self.m.bias, y, RELU_IN if relu_in else 0, gate, stats, self.transposed)
        return call, (y, stats, rows, s)

    def _pad_input_channels(self, x):
        c4 = (x.C + 3) // 4 * 4
        xp = K.as_view(K.zeros_ndhwc(x.B, c4, x.D, x.H, x.W, x.t.device))
        xp.t[:, :x.C].copy_(x.t)
        w = self.m.weight
        wp = torch.zeros((w.shape[0], c4) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
        wp[:, :x.C].copy_(w.detach())
        wp._n3d_nopack = True
        return xp, wp

    def fwd(self, x, relu_in, gate, want_stats):
        call, res = self.fwd_prepare(x, relu_in, gate, want_stats)
        g, xx, w, b, y, fl, gt, stats, tr = call
        K.conv_fwd(g, xx, w, b, y, fl, gt, stats, tr)
        return res

    def bwd_call(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias):
        """call tuple for K.conv_bwd_both2 if this backward is a plain (data + weight gradient) pair candidate, else None"""
        x, g = saved.x, saved.g
        dw = K.grad_target(self.m.weight)
        db = None if skip_bias else K.grad_target(self.m.bias)
        if not (need_dx and saved.pre is None and saved.wpad is None and dw is not None and g.Ci % 16 == 0 and g.Co % 16 == 0):
            return None
        if self.transposed and (db is not None or saved.relu_in or saved.gate is not None):
            return None
        if dx_out is None:
            dx_out = K.like(x)
            dx_acc = False
        call = (g, x, draw, self.m.weight, dx_out, dw, db, ACCUMULATE if dx_acc else 0, x if saved.relu_in else None, saved.gate,
                RELU_IN if saved.relu_in else 0, saved.gate, self.transposed)
        return call, dx_out, [dw, db]

    def bwd_data_call(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias):
        """call tuple for K.conv_bwd_data2 if this backward is a data gradient only (frozen weights: the architecture pass of
        the search step), else None"""
        x, g = saved.x, saved.g
        if not need_dx or saved.pre is not None or saved.wpad is not None or self.m.weight.requires_grad or self.m.bias.requires_grad:
            return None
        if self.transposed and (saved.relu_in or saved.gate is not None):
            return None
        if dx_out is None:
            dx_out = K.like(x)
            dx_acc = False
        call = (g, draw, self.m.weight, dx_out, ACCUMULATE if dx_acc else 0, x if saved.relu_in else None, saved.gate, self.transposed)
        return call, dx_out, [None, None]

    def _bwd_padded(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias):
        """backward of a conv whose input channels were zero-padded (fwd_prepare): gradients of the padded tensors, real slices out"""
        x, g, wp, cin = saved.x, saved.g, saved.wpad, saved.cin
        dw = K.grad_target(self.m.weight)
        db = None if skip_bias else K.grad_target(self.m.bias)
        if dw is not None or db is not None:
            dwp = torch.empty_like(wp) if dw is not None else None
            K.conv_bwd_weight(g, x, draw, dwp, db, RELU_IN if saved.relu_in else 0, saved.gate, False, defer=False)
            if dw is not None:
                dw.copy_(dwp[:, :cin])
        dx = None
        if need_dx:
            dxp = K.like(x)
            K.conv_bwd_data(g, draw, wp, dxp, 0, x if saved.relu_in else None, saved.gate, False)
            if dx_out is None:
                dx_out = K.as_view(K.empty_ndhwc(x.B, cin, x.D, x.H, x.W, x.t.device, x.t.dtype))
                dx_acc = False
            if dx_acc:
                dx_out.t.add_(dxp.t[:, :cin])
            else:
                dx_out.t.copy_(dxp.t[:, :cin])
            dx = dx_out.t
        return dx, [dw, db]

    def bwd(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias=False):
        x, g = saved.x, saved.g
        if saved.wpad is not None:
            return self._bwd_padded(saved, draw, need_dx, dx_out, dx_acc, skip_bias)
        w = self.m.weight
        dw = K.grad_target(w)
        db = None if skip_bias else K.grad_target(self.m.bias)
        if (need_dx and saved.pre is None and self.transposed and dw is not None and db is None
                and g.Ci % 16 == 0 and g.Co % 16 == 0):
            if dx_out is None:
                dx_out = K.like(x)
                dx_acc = False
            K.conv_bwd_both(g, x, draw, w, dx_out, dw, None, ACCUMULATE if dx_acc else 0, None, None, 0, None, True)
            return dx_out.t, [dw, db]
        if (need_dx and saved.pre is None and not self.transposed and (dw is not None or db is not None)
                and g.Ci % 16 == 0 and g.Co % 16 == 0):
            # deep levels: data gradient and weight gradient in one launch where libn3d can fold them
            if dx_out is None:
                dx_out = K.like(x)
                dx_acc = False
            K.conv_bwd_both(g, x, draw, w, dx_out, dw, db, ACCUMULATE if dx_acc else 0, x if saved.relu_in else None, saved.gate,
                            RELU_IN if saved.relu_in else 0, saved.gate)
            return dx_out.t, [dw, db]
        if dw is not None or db is not None:
            K.conv_bwd_weight(g, x, draw, dw, db, RELU_IN if saved.relu_in else 0, saved.gate, self.transposed)
        dx = None
        if need_dx:
            if saved.pre is not None:
                du = K.like(x)
                K.conv_bwd_data(g, draw, w, du, 0, None, None, self.transposed)
                x0, r0, g0 = saved.pre
                dx = _pre_backward(x0, r0, g0, du, dx_out, dx_acc).t
            else:
                if dx_out is None:
                    dx_out = K.like(x)
                    dx_acc = False
                K.conv_bwd_data(g, draw, w, dx_out, ACCUMULATE if dx_acc else 0, x if saved.relu_in else None, saved.gate,
                                self.transposed)
                dx = dx_out.t
        return dx, [dw, db]


class DepthSepW(WeightProgram):
    """depth_conv -> point_conv of a depthwised ConvOps (prim_ops.py:95-98,105-107,111-114)."""
    produces_stats = True

    def __init__(self, depth_module, point_module, stride, pad, transposed):
        self.dm, self.pm = depth_module, point_module
        self.stride, self.pad, self.transposed = stride, pad, transposed

    def params(self):
        return [self.dm.weight, self.dm.bias, self.pm.weight, self.pm.bias]

    def norm_fed_bias(self):
        return self.pm.bias

    def dgeom(self, x):
        c = x.C
        if self.transposed:
            opad = 0 if self.stride == 1 else 1
            def od(i):
                return (i - 1) * self.stride - 2 * self.pad + 2 + opad + 1
            return K.conv_geom(x.B, od(x.D), od(x.H), od(x.W), c, c, 3, self.stride, 1, self.pad, True)
        return K.conv_geom(x.B, x.D, x.H, x.W, c, c, 3, self.stride, 1, self.pad, True)

    def out_shape(self, x):
        g = self.dgeom(x)
        co = self.pm.weight.shape[0]
        return (g.B, co, g.Di, g.Hi, g.Wi) if self.transposed else (g.B, co, g.Do, g.Ho, g.Wo)

    def fwd_depth(self, x, relu_in, gate, launch=True):
        """first stage: the depthwise conv.  Returns the saved state (s.mid = its output); launch=False leaves the launch to the
        caller (s.job is the K.dwconv_batch job)"""
        s = Saved()
        s.pre = None
        if relu_in or gate is not None:
            s.pre = (x, relu_in, gate)
            x = _materialise_pre(x, relu_in, gate)
        gd = self.dgeom(x)
        mid_shape = (gd.B, gd.Ci, gd.Di, gd.Hi, gd.Wi) if self.transposed else (gd.B, gd.Co, gd.Do, gd.Ho, gd.Wo)
        mid = K.as_view(K.empty_ndhwc(*mid_shape, x.t.device))
        s.job = (gd, self.transposed, x, self.dm.weight, self.dm.bias, mid, 0)
        if launch:
            K.conv_fwd(gd, x, self.dm.weight, self.dm.bias, mid, 0, None, None, self.transposed)
        co = self.pm.weight.shape[0]
        s.x, s.mid, s.gd = x, mid, gd
        s.gp = K.conv_geom(mid.B, mid.D, mid.H, mid.W, mid.C, co, 1, 1, 1, 0)
        return s

    def point_call(self, s, want_stats):
        """second stage, the 1x1x1 conv, as a call tuple for K.conv_fwd / K.conv_fwd2 (two of them fold into one launch) and
        its (y, stats, rows)"""
        mid, gp = s.mid, s.gp
        co = self.pm.weight.shape[0]
        y = K.as_view(K.empty_ndhwc(mid.B, co, mid.D, mid.H, mid.W, mid.t.device))
        stats, rows = None, 0
        if want_stats:
            rows = K.conv_stats_rows(gp, False, 0, mid, y)
            if rows > 0:
                stats = torch.empty((mid.B, rows, co, 2), dtype=torch.float64, device=mid.t.device)
        return (gp, mid, self.pm.weight, self.pm.bias, y, 0, None, stats, False), (y, stats, rows)

    def fwd(self, x, relu_in, gate, want_stats):
        s = self.fwd_depth(x, relu_in, gate)
        call, (y, stats, rows) = self.point_call(s, want_stats)
        K.conv_fwd(*call)
        return y, stats, rows, s

    def point_bwd_call(self, saved, draw, skip_bias):
        """backward of the 1x1x1 conv as a call tuple for K.conv_bwd_both2 (weights trainable) or K.conv_bwd_data2 (frozen), or
        None when its shape has no folded kernel.  Returns (kind, call, dmid, [g_pw, g_pb])"""
        mid, gp = saved.mid, saved.gp
        if gp.Ci % 16 != 0 or gp.Co % 16 != 0:
            return None
        pw, pb = self.pm.weight, self.pm.bias
        g_pw = K.grad_target(pw)
        g_pb = None if skip_bias else K.grad_target(pb)
        dmid = K.like(mid)
        if g_pw is not None:
            return "both", (gp, mid, draw, pw, dmid, g_pw, g_pb, 0, None, None, 0, None, False), dmid, [g_pw, g_pb]
        if g_pb is None:
            return "data", (gp, draw, pw, dmid, 0, None, None, False), dmid, [None, None]
        return None

    def bwd_point(self, saved, draw, skip_bias):
        """backward of the 1x1x1 conv alone.  Returns (dmid, [g_pw, g_pb])"""
        mid, gp = saved.mid, saved.gp
        c = self.point_bwd_call(saved, draw, skip_bias)
        if c is not None and c[0] == "both":
            K.conv_bwd_both(*c[1][:12])          # data + weight gradient of the 1x1x1 conv in one launch
            return c[2], c[3]
        pw, pb = self.pm.weight, self.pm.bias
        g_pw = K.grad_target(pw)
        g_pb = None if skip_bias else K.grad_target(pb)
        if g_pw is not None or g_pb is not None:
            K.conv_bwd_weight(gp, mid, draw, g_pw, g_pb, 0, None, False)
        dmid = K.like(mid)
        K.conv_bwd_data(gp, draw, pw, dmid, 0, None, None, False)
        return dmid, [g_pw, g_pb]

    def depth_data_job(self, saved, dmid, dx_out, dx_acc):
        """the depthwise data gradient as a K.dwconv_batch job (None: this op needs the materialised-input path).
        Returns (job, dx_out)"""
        if saved.pre is not None:
            return None
        x = saved.x
        if dx_out is None:
            dx_out = K.like(x)
            dx_acc = False
        return (saved.gd, not self.transposed, dmid, self.dm.weight, None, dx_out, ACCUMULATE if dx_acc else 0), dx_out

    def bwd_depth(self, saved, dmid, need_dx, dx_out, dx_acc, data=True):
        """backward of the depthwise conv given d(mid).  Returns (dx, [g_dw, g_db]); data=False: weight gradient only"""
        x, gd = saved.x, saved.gd
        dwt, dbs = self.dm.weight, self.dm.bias
        g_dw = K.grad_target(dwt)
        g_db = K.grad_target(dbs)
        if g_dw is not None or g_db is not None:
            K.conv_bwd_weight(gd, x, dmid, g_dw, g_db, 0, None, self.transposed)
        dx = None
        if need_dx and data:
            if saved.pre is not None:
                du = K.like(x)
                K.conv_bwd_data(gd, dmid, dwt, du, 0, None, None, self.transposed)
                x0, r0, g0 = saved.pre
                dx = _pre_backward(x0, r0, g0, du, dx_out, dx_acc).t
            else:
                if dx_out is None:
                    dx_out = K.like(x)
                    dx_acc = False
                K.conv_bwd_data(gd, dmid, dwt, dx_out, ACCUMULATE if dx_acc else 0, None, None, self.transposed)
                dx = dx_out.t
        return dx, [g_dw, g_db]

    def bwd(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias=False):
        dmid, (g_pw, g_pb) = self.bwd_point(saved, draw, skip_bias)
        dx, (g_dw, g_db) = self.bwd_depth(saved, dmid, need_dx, dx_out, dx_acc)
        return dx, [g_dw, g_db, g_pw, g_pb]


class SEGate:
    """avg_pool -> fc of SEConvOp (prim_ops.py:133-139,148-151); shared by the two SE programs."""

    def __init__(self, fc):
        self.fc = fc  # nn.Sequential(Linear(C,1), ReLU, Linear(1,C), Sigmoid)

    def params(self):
        return [self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias]

    def fwd(self, x):
        st, rows = K.channel_stats(x)
        mean, hidden, gate = K.se_gate_fwd(st, rows, x.N, self.fc[0].weight, self.fc[0].bias, self.fc[2].weight,
                                           self.fc[2].bias, x.B, x.C)
        return mean, hidden, gate


class SEConvW(WeightProgram):
    """SEConvOp.weight_call with stride 2: conv(x * gate(x)) (prim_ops.py:148-153)."""
    produces_stats = True

    def __init__(self, fc, conv_module, stride, pad, transposed):
        self.gate = SEGate(fc)
        self.conv = DenseConvW(conv_module, 3, stride, 1, pad, transposed)

    def params(self):
        return self.gate.params() + self.conv.params()

    def norm_fed_bias(self):
        return self.conv.m.bias

    def out_shape(self, x):
        return self.conv.out_shape(x)

    def fwd_prepare(self, x, gate3, want_stats):
        """everything of fwd() but the gate and the conv launch: gate3 = (mean, hidden, gate) of SEGate.fwd(x) (or of a batched
        K.se_gate_fwdN).  Returns (call tuple for K.conv_fwd / conv_fwdN, [y, stats, rows, saved])"""
        mean, hidden, g = gate3
        u = K.like(x)
        K.affine_act(x, g, None, None, u, 0)
        call, (y, stats, rows, cs) = self.conv.fwd_prepare(u, False, None, want_stats)
        s = Saved()
        s.x, s.mean, s.hidden, s.gate, s.cs = x, mean, hidden, g, cs
        return call, [y, stats, rows, s]

    def fwd(self, x, relu_in, gate, want_stats):
        if relu_in or gate is not None:
            raise N3DError("SE conv with act-before-weight / dropout is not supported")
        call, r = self.fwd_prepare(x, self.gate.fwd(x), want_stats)
        g, xx, w, b, y, fl, gt, stats, tr = call
        K.conv_fwd(g, xx, w, b, y, fl, gt, stats, tr)
        return tuple(r)

    def bwd_tail(self, saved, du, cg, need_dx, dx_out, dx_acc, pre=None):
        """backward behind the conv: du = d(x * gate).  pre = ((sums, rows) | None, se_gate_bwd outputs | None) when those passes
        already ran batched.  Returns (dx, grads)"""
        x = saved.x
        fc = self.gate.fc
        if pre is not None and pre[1] is not None:
            dw1, db1, dw2, db2, A, Bc = pre[1]
        else:
            sums, rows = pre[0] if pre is not None and pre[0] is not None else K.affine_act_bwd_reduce(du, x, None, None, 0)
            dw1, db1, dw2, db2, A, Bc = K.se_gate_bwd(sums, rows, None, saved.mean, saved.hidden, saved.gate, fc[0].weight,
                                                      fc[2].weight, x.B, x.C, x.N, None, fc)
        dx = None
        if need_dx:
            if dx_out is None:
                dx_out = K.like(x)
                dx_acc = False
            K.affine_act_bwd_apply(du, x, None, None, A, Bc, None, dx_out, ACCUMULATE if dx_acc else 0)
            dx = dx_out.t
        return dx, [dw1, db1, dw2, db2] + cg

    def bwd(self, saved, draw, need_dx, dx_out, dx_acc, skip_bias=False):
        du_t, cg = self.conv.bwd(saved.cs, draw, True, None, False, skip_bias)
        return self.bwd_tail(saved, K.as_view(du_t), cg, need_dx, dx_out, dx_acc)


# =================================================================================================
# segments
# =================================================================================================
class Segment:
    """[relu_in] -> weight -> [norm] -> [relu_out]; `se_gate` marks the stride-1 SE op whose whole
    effect is the epilogue scale x * gate (prim_ops.py:127-128,152)."""

    def __init__(self, weight, norm=None, relu_in=False, relu_out=False, se_gate=None, dropout=None):
        self.weight = weight if weight is not None else IdentityW()
        self.norm = norm
        self.relu_in = relu_in
        self.relu_out = relu_out
        self.se_gate = se_gate
        self.dropout = dropout  # nn.Dropout3d applied before the weight op (prim_ops.py:72-73)

    def params(self):
        p = list(self.weight.params())
        if self.se_gate is not None:
            p += self.se_gate.params()
        if self.norm is not None:
            p += [self.norm.weight, self.norm.bias]
        return p


def _wptr(alpha_row, k):
    if alpha_row is None:
        return None
    return C.c_void_p(alpha_row.data_ptr() + 4 * k)


RECOMPUTE_K1 = True     # (tests switch it off to compare with the stored form)


def recompute_ok(seg, x):
    """can this segment run WITHOUT storing its raw conv output (include/n3d.h, n3d_conv_k1_norm_*)?  A 'weight_norm' 1x1x1 conv with
    4 (8) input channels on a large volume: stem0.  The caller must not need the op's input gradient."""
    w = seg.weight
    if not (RECOMPUTE_K1 and isinstance(w, DenseConvW) and w.k == 1 and w.stride == 1 and not w.transposed and seg.norm is not None
            and not seg.relu_in and not seg.relu_out and seg.se_gate is None and seg.dropout is None and x.C % 4 == 0):
        return False
    g = w.geom(x)
    # the recompute form takes the conv-bias gradient analytically out of the GroupNorm sums (N * Bc + ...): not with a padded norm
    # (G < 0: the kernels run with the scaled element count of include/n3d.h, "padded channels"), nor with the analytic form off
    if not ANALYTIC_CONV_BIAS or gn_groups(seg.norm, g.Co) < 0:
        return False
    return K.conv_k1_norm_ok(g)


def _seg_forward_recompute(seg, x, out=None):
    """the segment as statistics pass -> coefficients -> conv + normalise pass: the raw tensor is never written"""
    w = seg.weight
    g = w.geom(x)
    Co = g.Co
    rows = K.conv_stats_rows(g, False, _lib_flags_src(x))
    stats = torch.empty((x.B, rows, Co, 2), dtype=torch.float64, device=x.t.device)
    K.conv_k1_norm_fwd(g, x, w.m.weight, w.m.bias, None, None, None, stats)
    G = gn_groups(seg.norm, Co)
    s = Saved()
    s.kind, s.G, s.raw, s.ws = "gn_rc", G, None, None
    s.x, s.g = x, g
    s.a, s.b, s.mr, s.sumraw = K.gn_coeffs(stats, rows, seg.norm.weight, seg.norm.bias, x.B, Co, G, g.Do * g.Ho * g.Wo, seg.norm.eps)
    if out is None:
        out = K.as_view(K.empty_ndhwc(g.B, Co, g.Do, g.Ho, g.Wo, x.t.device))
    K.conv_k1_norm_fwd(g, x, w.m.weight, w.m.bias, out, s.a, s.b, None)
    return out, s


def _lib_flags_src(x):
    from ._lib import SRC_BF16, BF16
    return SRC_BF16 if x.dt == BF16 else 0


def _seg_backward_recompute(seg, s, dout, need_dx):
    if need_dx:
        raise N3DError("a segment run in recompute form has no input gradient (programs.recompute_ok: the caller decides at forward time)")
    w = seg.weight
    x, g = s.x, s.g
    cbias = w.m.bias if w.m.bias.requires_grad else None
    sums, rows = K.conv_k1_norm_bwd_reduce(g, x, w.m.weight, w.m.bias, dout, s.a, s.b)
    dgamma, dbeta, A, Bc, Cc, dcb = K.gn_bwd_coeffs(sums, rows, seg.norm.weight, s.mr, None, x.B, g.Co, s.G, g.Do * g.Ho * g.Wo, None,
                                                    seg.norm.bias, s.sumraw, cbias)
    dw = K.grad_target(w.m.weight)
    if dw is not None:
        K.conv_k1_norm_bwd_apply_wgrad(g, x, w.m.weight, w.m.bias, dout, s.a, s.b, A, Bc, Cc, dw)
    return None, [dw, dcb, dgamma, dbeta]


def seg_forward(seg, x, drop_gate=None, out=None, accumulate=False, alpha_row=None, alpha_k=0, recompute=False):
    """Run one segment.  x/out: kernels.View.  Returns (out_view, saved).  recompute: the caller does not need the op's input gradient
    and the raw conv output may stay unwritten where the op qualifies (recompute_ok)."""
    if recompute and drop_gate is None and not accumulate and alpha_row is None and recompute_ok(seg, x):
        return _seg_forward_recompute(seg, x, out)
    want_stats = seg.norm is not None and seg.weight.produces_stats
    raw, stats, rows, ws = seg.weight.fwd(x, seg.relu_in, drop_gate, want_stats)
    return _seg_epilogue_forward(seg, raw, stats, rows, ws, out, accumulate, alpha_row, alpha_k)


def _seg_epilogue_forward(seg, raw, stats, rows, ws, out, accumulate, alpha_row, alpha_k):
    """[norm] -> [ReLU] -> weighted accumulation of a segment whose weight op has produced `raw`"""
    s = Saved()
    wp = _wptr(alpha_row, alpha_k)
    s.ws, s.raw = ws, raw
    s.a = s.b = s.mr = None
    s.kind = "plain"
    if seg.norm is not None:
        if stats is None:
            stats, rows = K.channel_stats(raw)
        cn = raw.C
        G = gn_groups(seg.norm, cn)
        s.kind, s.G = "gn", G
        if rows <= K.fused_max_rows():
            # small tensor: coefficients are computed in the epilogue kernel's prologue (one launch less)
            if out is None:
                out = K.like(raw)
                accumulate = False
            fl = (RELU if seg.relu_out else 0) | (ACCUMULATE if accumulate else 0)
            s.a, s.b, s.mr, s.sumraw = K.affine_act_gn(raw, stats, rows, seg.norm.weight, seg.norm.bias, G, seg.norm.eps, wp, out, fl)
            return out, s
        s.a, s.b, s.mr, s.sumraw = K.gn_coeffs(stats, rows, seg.norm.weight, seg.norm.bias, raw.B, cn, G, raw.N, seg.norm.eps)
    elif seg.se_gate is not None:
        s.mean, s.hidden, s.a = seg.se_gate.fwd(raw)
        s.kind = "se"
    need_pass = s.kind != "plain" or seg.relu_out or wp is not None or out is not None
    if not need_pass:
        return raw, s
    if out is None:
        out = K.like(raw)
        accumulate = False
    fl = (RELU if seg.relu_out else 0) | (ACCUMULATE if accumulate else 0)
    K.affine_act(raw, s.a, s.b, wp, out, fl)
    return out, s


def _pairable_fwd(seg):
    return (seg.norm is not None and seg.weight.produces_stats and seg.se_gate is None)


def gn_pairable(seg):
    """segment whose epilogue can share a launch with another one of the same output shape (see pair_forward)"""
    return _pairable_fwd(seg) and not isinstance(seg.weight, IdentityW) and seg.dropout is None


def pair_forward(segA, xA, segB, xB, out, outB=None, accumulate=False, alphaA=None, alphaB=None):
    """Node of a searched cell: out = segA(xA) + segB(xB) (searched.py:45-50), `out` a View that is overwritten.
    With `outB` the two ops are independent (the two preprocess ops of a cell, cell.py:47-50): segA -> out, segB -> outB.
    Supernet nodes (cell.py:76-81) use it too: alphaX = (alpha row tensor, column) is the MixedOp weight of the term and
    `accumulate` adds the pair to what the node buffer already holds.
    Both weight ops run first; if both epilogues are small GroupNorm epilogues of one shape they share ONE launch
    (n3d_affine_act_gn2), otherwise the two ordinary epilogues run one after the other.  Returns (savedA, savedB)."""
    res = []
    if isinstance(segA.weight, DenseConvW) and isinstance(segB.weight, DenseConvW):
        # both weight ops are plain convs: one launch where libn3d can fold them (n3d_conv_fwd2)
        calls = []
        for seg, x in ((segA, xA), (segB, xB)):
            call, r = seg.weight.fwd_prepare(x, seg.relu_in, None, seg.norm is not None)
            calls.append(call)
            res.append(list(r))
        K.conv_fwd2(calls)
        for seg, r in zip((segA, segB), res):
            if seg.norm is not None and r[1] is None:
                r[1], r[2] = K.channel_stats(r[0])
        res = [tuple(r) for r in res]
    else:
        res = [pair_weight_phase(seg, x) for seg, x in ((segA, xA), (segB, xB))]
    return pair_epilogue_phase(segA, res[0], segB, res[1], out, outB, accumulate, alphaA, alphaB)


def pair_weight_phase(seg, x):
    """the weight op of ONE op of a pair and the channel statistics its epilogue needs: (raw, stats, rows, weight-op state).  The
    side-stream schedule launches the op of a searched-cell node whose input is not the node computed last early, on the side
    stream (fused._run_forward_impl), and hands the result to pair_epilogue_phase."""
    want_stats = seg.norm is not None and seg.weight.produces_stats
    raw, stats, rows, ws = seg.weight.fwd(x, seg.relu_in, None, want_stats)
    if seg.norm is not None and stats is None:
        stats, rows = K.channel_stats(raw)
    return (raw, stats, rows, ws)


def pair_epilogue_phase(segA, resA, segB, resB, out, outB=None, accumulate=False, alphaA=None, alphaB=None):
    """second half of pair_forward: the two epilogues (one launch where they pair) of weight ops that are complete on the
    launching stream."""
    res = [resA, resB]
    (rawA, stA, rowsA, wsA), (rawB, stB, rowsB, wsB) = res
    G = gn_groups(segA.norm, rawA.C)
    if (_pairable_fwd(segA) and _pairable_fwd(segB) and rawA.C == rawB.C and rawA.N == rawB.N
            and segA.norm.eps == segB.norm.eps and K.pair_shape_ok(rawA.C)):
        wpA = _wptr(*alphaA) if alphaA is not None else None
        wpB = _wptr(*alphaB) if alphaB is not None else None
        terms = [(rawA, stA, rowsA, segA.norm.weight, segA.norm.bias, wpA, segA.relu_out),
                 (rawB, stB, rowsB, segB.norm.weight, segB.norm.bias, wpB, segB.relu_out)]
        sv = K.affine_act_gn2(terms, G, segA.norm.eps, out, ACCUMULATE if accumulate else 0, outB)
        saved = []
        for (raw, _, _, ws), (a, b, mr, sr) in zip(res, sv):
            s = Saved()
            s.ws, s.raw, s.kind, s.G = ws, raw, "gn", G
            s.a, s.b, s.mr, s.sumraw = a, b, mr, sr
            saved.append(s)
        return saved[0], saved[1]
    aA, kA = alphaA if alphaA is not None else (None, 0)
    aB, kB = alphaB if alphaB is not None else (None, 0)
    _, sA = _seg_epilogue_forward(segA, rawA, stA, rowsA, wsA, out, accumulate, aA, kA)
    if outB is not None:
        _, sB = _seg_epilogue_forward(segB, rawB, stB, rowsB, wsB, outB, False, aB, kB)
    else:
        _, sB = _seg_epilogue_forward(segB, rawB, stB, rowsB, wsB, out, True, aB, kB)
    return sA, sB


def pair_backward(segA, sA, segB, sB, dout, argsA, argsB, doutB=None, alphaA=None, alphaB=None):
    """Backward of pair_forward.  argsX = (need_dx, dx_out, dx_acc).  Returns ((dxA, gradsA), (dxB, gradsB)) with
    grads ordered like segX.params().  doutB: gradient of segB's own output (independent-outputs mode).
    alphaX = (alpha row, column, dalpha row | None): MixedOp weight of the term and where its gradient <dout, z> goes."""
    aA = alphaA if alphaA is not None else (None, 0, None)
    aB = alphaB if alphaB is not None else (None, 0, None)
    pair = (sA.kind == "gn" and sB.kind == "gn" and not isinstance(segA.weight, IdentityW) and not isinstance(segB.weight, IdentityW)
            and sA.raw.C == sB.raw.C and sA.raw.N == sB.raw.N)
    if pair:
        pair = K.pair_shape_ok(sA.raw.C)
    if not pair:
        rb = seg_backward(segB, sB, doutB if doutB is not None else dout, *argsB, aB[0], aB[1], aB[2])
        ra = seg_backward(segA, sA, dout, *argsA, aA[0], aA[1], aA[2])
        return ra, rb
    terms = [_gn_bwd_term(segA, sA, aA), _gn_bwd_term(segB, sB, aB)]
    outs = K.affine_act_bwd_gn2(dout, terms, sA.G, doutB)
    # weight-op backward in reverse forward order (B then A), as the unpaired path does
    order = ((segB, sB, terms[1], outs[1], argsB), (segA, sA, terms[0], outs[0], argsA))
    results = _weight_backward(order)
    return results[1], results[0]


def pair_backward_epilogue(segA, sA, segB, sB, dout, argsA, argsB):
    """First half of pair_backward for a searched-cell node whose two terms pair: the epilogue backward (both d(raw) and the norms'
    parameter gradients).  Returns the two `_weight_backward` items (A, B), or None when the pair form does not apply.  The caller
    runs the weight-op backwards itself -- fused._run_backward_impl puts the one whose input gradient goes to a preprocess output
    on the side stream."""
    if not (sA.kind == "gn" and sB.kind == "gn" and not isinstance(segA.weight, IdentityW) and not isinstance(segB.weight, IdentityW)
            and sA.raw.C == sB.raw.C and sA.raw.N == sB.raw.N and K.pair_shape_ok(sA.raw.C)):
        return None
    none = (None, 0, None)
    terms = [_gn_bwd_term(segA, sA, none), _gn_bwd_term(segB, sB, none)]
    outs = K.affine_act_bwd_gn2(dout, terms, sA.G, None)
    return (segA, sA, terms[0], outs[0], argsA), (segB, sB, terms[1], outs[1], argsB)


def _gn_bwd_term(seg, s, alpha):
    """descriptor of one GroupNorm-type term for K.affine_act_bwd_gn2 / affine_act_bwd_gnN; alpha = (row, column, dalpha row | None)"""
    cbias = seg.weight.norm_fed_bias() if ANALYTIC_CONV_BIAS else None
    if cbias is not None and not cbias.requires_grad:
        cbias = None
    raw = s.raw
    return dict(raw=raw, a=s.a, b=s.b, mr=s.mr, sumraw=s.sumraw, gamma=seg.norm.weight, beta=seg.norm.bias,
                wptr=_wptr(alpha[0], alpha[1]), dalpha_ptr=(C.c_void_p(alpha[2].data_ptr() + 4 * alpha[1]) if alpha[2] is not None else None),
                relu=seg.relu_out, conv_bias=cbias,
                draw=K.like(raw))


def _weight_backward(order):
    """Weight-op backwards of GroupNorm-type terms whose d(raw) is ready.  order = [(seg, saved, term dict, (dgamma, dbeta,
    dconv_bias), (need_dx, dx_out, dx_acc))] in execution order.  Two neighbouring plain convs with distinct input-gradient
    targets share one launch (data + weight gradients: n3d_conv_bwd_both2; frozen weights: n3d_conv_bwd_data2).
    Returns [(dx, grads ordered like seg.params())] in the same order."""
    results = []
    i = 0
    while i < len(order):
        pre = None
        if i + 1 < len(order) and isinstance(order[i][0].weight, DenseConvW) and isinstance(order[i + 1][0].weight, DenseConvW):
            two = order[i:i + 2]
            cands = [o[0].weight.bwd_call(o[1].ws, o[2]["draw"], o[4][0], o[4][1], o[4][2], o[3][2] is not None) for o in two]
            if all(c is not None for c in cands) and cands[0][1].p.value != cands[1][1].p.value:
                K.conv_bwd_both2([c[0] for c in cands])
                pre = [(c[1].t, c[2]) for c in cands]
            else:
                cands = [o[0].weight.bwd_data_call(o[1].ws, o[2]["draw"], o[4][0], o[4][1], o[4][2], o[3][2] is not None) for o in two]
                if all(c is not None for c in cands) and cands[0][1].p.value != cands[1][1].p.value:
                    K.conv_bwd_data2([c[0] for c in cands])
                    pre = [(c[1].t, c[2]) for c in cands]
        npre = 2
        if pre is None and i + 1 < len(order) and isinstance(order[i][0].weight, SEConvW) and isinstance(order[i + 1][0].weight, SEConvW):
            # a run of stride-2 SE convs: conv backwards two per launch, the gate backwards in one launch
            j = i
            while j < len(order) and isinstance(order[j][0].weight, SEConvW):
                j += 1
            run = order[i:j]
            dus, cgs = [None] * len(run), [None] * len(run)
            k = 0
            while k < len(run):
                done = False
                if k + 1 < len(run):
                    two = run[k:k + 2]
                    for kind, fn in (("bwd_call", K.conv_bwd_both2), ("bwd_data_call", K.conv_bwd_data2)):
                        cands = [getattr(o[0].weight.conv, kind)(o[1].ws.cs, o[2]["draw"], True, None, False, o[3][2] is not None) for o in two]
                        if all(c is not None for c in cands):
                            fn([c[0] for c in cands])
                            for q, c in zip((k, k + 1), cands):
                                dus[q], cgs[q] = c[1], list(c[2])
                            done = True
                            break
                if done:
                    k += 2
                    continue
                o = run[k]
                du_t, cg = o[0].weight.conv.bwd(o[1].ws.cs, o[2]["draw"], True, None, False, o[3][2] is not None)
                dus[k], cgs[k] = K.as_view(du_t), list(cg)
                k += 1
            red = [K.affine_act_bwd_reduce(du, o[1].ws.x, None, None, 0) for o, du in zip(run, dus)]
            x0 = run[0][1].ws.x
            gpre = [None] * len(run)
            if len({(o[1].ws.x.B, o[1].ws.x.C, o[1].ws.x.N) for o in run}) == 1:
                tds = [dict(sums=r[0], rows=r[1], wptr=None, mean=o[1].ws.mean, hidden=o[1].ws.hidden, gate=o[1].ws.gate,
                            fc=o[0].weight.gate.fc, dalpha_ptr=None) for o, r in zip(run, red)]
                gpre = K.se_gate_bwdN(tds, x0.N, x0.B, x0.C)
            pre = []
            for o, du, cg, r, gp_ in zip(run, dus, cgs, red, gpre):
                pre.append(o[0].weight.bwd_tail(o[1].ws, du, cg, o[4][0], o[4][1], o[4][2], (r, gp_)))
            npre = len(run)
        if (pre is None and i + 1 < len(order) and isinstance(order[i][0].weight, DepthSepW)
                and isinstance(order[i + 1][0].weight, DepthSepW)):
            # a run of depthwise-separable primitives: the 1x1x1 convs' backward two per launch, the depthwise weight gradients
            # one by one, the depthwise data gradients (distinct targets) in one launch
            j = i
            while j < len(order) and isinstance(order[j][0].weight, DepthSepW):
                j += 1
            run = order[i:j]
            cands = [o[0].weight.point_bwd_call(o[1].ws, o[2]["draw"], o[3][2] is not None) for o in run]
            dmids, gpw = [None] * len(run), [None] * len(run)
            k = 0
            while k < len(run):
                if k + 1 < len(run) and cands[k] is not None and cands[k + 1] is not None and cands[k][0] == cands[k + 1][0]:
                    (K.conv_bwd_both2 if cands[k][0] == "both" else K.conv_bwd_data2)([cands[k][1], cands[k + 1][1]])
                    for q in (k, k + 1):
                        dmids[q], gpw[q] = cands[q][2], cands[q][3]
                    k += 2
                else:
                    dmids[k], gpw[k] = run[k][0].weight.bwd_point(run[k][1].ws, run[k][2]["draw"], run[k][3][2] is not None)
                    k += 1
            jobs = [o[0].weight.depth_data_job(o[1].ws, dm, o[4][1], o[4][2]) if o[4][0] else None for o, dm in zip(run, dmids)]
            # batches: jobs of one destination shape (a down / up cell node mixes stride-2 and stride-1 edges), distinct targets
            shapes = {}
            if len({jb[1].p.value for jb in jobs if jb is not None}) == sum(jb is not None for jb in jobs):
                for q, jb in enumerate(jobs):
                    if jb is not None:
                        shapes.setdefault((jb[1].B, jb[1].C, jb[1].D, jb[1].H, jb[1].W), []).append(q)
            batched = {q for qs in shapes.values() if len(qs) >= 2 for q in qs}
            pre = []
            for q, (o, dm, gp_, jb) in enumerate(zip(run, dmids, gpw, jobs)):
                dx, gdw = o[0].weight.bwd_depth(o[1].ws, dm, o[4][0], o[4][1], o[4][2], data=q not in batched)
                pre.append((jb[1].t if q in batched else dx, gdw + gp_))
            for qs in shapes.values():
                if len(qs) >= 2:
                    K.dwconv_batch([jobs[q][0] for q in qs])
            npre = len(run)
        for k in range(npre if pre is not None else 1):
            seg, s, t, (dgamma, dbeta, dcb), (need_dx, dx_out, dx_acc) = order[i + k]
            if pre is not None:
                dx, wg = pre[k]
            else:
                dx, wg = seg.weight.bwd(s.ws, t["draw"], need_dx, dx_out, dx_acc, dcb is not None)
            wg = list(wg)
            if dcb is not None:
                for j, p in enumerate(seg.weight.params()):
                    if p is t["conv_bias"]:
                        wg[j] = dcb
            results.append((dx, wg + [dgamma, dbeta]))
        i += npre if pre is not None else 1
    return results


def group_ok(segs):
    """terms whose BACKWARD epilogues can run as one N-term group (K.affine_act_bwd_gnN): GroupNorm-type, one eps, 3..8 of them"""
    return (3 <= len(segs) <= K.MAX_GROUP_TERMS and all(gn_pairable(g) for g in segs)
            and all(g.norm.eps == segs[0].norm.eps for g in segs))


NODE_FWD_COEFFS = True   # GroupNorm coefficients + SE gates of a group in one launch (tests compare with the per-term launches)


def group_forward(terms, out, accumulate):
    """Supernet node (cell.py:76-81): out (+)= sum_k alpha_k * op_k(x_k) for up to 8 primitives of any kind (GroupNorm-type
    convs, identity-with-norm, SE gates, pooling).  terms = [(segment, input View, alpha row | None, alpha column)].
    The weight ops run first (two neighbouring plain convs in one launch), then the coefficients -- all GroupNorm ones in one
    launch, one per SE gate -- and ONE pass over `out` for everything.  Returns the saved states, in term order.
    The two halves are callable on their own: the weight phase of a term only reads the term's input, so the trainers' side-stream
    schedule runs it early on a second stream for the terms whose input is not the node computed last (fused._run_forward_side)."""
    return group_epilogue_phase(terms, group_weight_phase(terms), out, accumulate)


def group_weight_phase(terms):
    """first half of group_forward: every weight op of `terms` (any number) and the channel statistics their epilogues need.
    Returns [[raw View, stats | None, rows, weight-op state]] in term order."""
    res = []
    # the gates of the stride-2 SE convs of this chunk: their input statistics in one launch, the gates in one launch
    sec = [k for k, t in enumerate(terms) if isinstance(t[0].weight, SEConvW)]
    gates = {}
    if len(sec) >= 2:
        sts = K.channel_statsN([terms[k][1] for k in sec])
        x0 = terms[sec[0]][1]
        if len({(terms[k][1].B, terms[k][1].C, terms[k][1].N) for k in sec}) == 1:
            for k, g3 in zip(sec, K.se_gate_fwdN([(st, rows, terms[k][0].weight.gate.fc) for k, (st, rows) in zip(sec, sts)], x0.N, x0.B, x0.C)):
                gates[k] = g3
    i = 0
    while i < len(terms):
        if isinstance(terms[i][0].weight, SEConvW) and (i in gates or (i + 1 < len(terms) and isinstance(terms[i + 1][0].weight, SEConvW))):
            # a run of stride-2 SE convs: x * gate each, then the convs up to four per launch
            j = i
            while j < len(terms) and j < i + 4 and isinstance(terms[j][0].weight, SEConvW):
                j += 1
            calls = []
            for k in range(i, j):
                seg, x, _, _ = terms[k]
                if seg.relu_in:
                    raise N3DError("SE conv with act-before-weight is not supported")
                g3 = gates[k] if k in gates else seg.weight.gate.fwd(x)
                call, r = seg.weight.fwd_prepare(x, g3, seg.norm is not None)
                calls.append(call)
                res.append(r)
            K.conv_fwdN(calls)
            i = j
        elif (i + 1 < len(terms) and isinstance(terms[i][0].weight, PoolW) and isinstance(terms[i + 1][0].weight, PoolW)
              and terms[i][0].weight.is_max != terms[i + 1][0].weight.is_max and terms[i][1].p.value == terms[i + 1][1].p.value
              and not (terms[i][0].relu_in or terms[i + 1][0].relu_in)):
            # the average and the max pooling of one edge: one pass over the input
            x = terms[i][1]
            if x.D % 2 or x.H % 2 or x.W % 2:
                raise N3DError("pool2: spatial dims must be even, got %s" % ((x.D, x.H, x.W),))
            ys = [K.as_view(K.empty_ndhwc(x.B, x.C, x.D // 2, x.H // 2, x.W // 2, x.t.device)) for _ in range(2)]
            a_first = not terms[i][0].weight.is_max
            K.pool2_fwd_both(x, ys[0] if a_first else ys[1], ys[1] if a_first else ys[0])
            for y in ys:
                ws = Saved()
                ws.x = x
                res.append([y, None, 0, ws])
            i += 2
        elif i + 1 < len(terms) and isinstance(terms[i][0].weight, DenseConvW) and isinstance(terms[i + 1][0].weight, DenseConvW):
            # a run of plain convs: up to four per launch
            j = i
            while j < len(terms) and j < i + 4 and isinstance(terms[j][0].weight, DenseConvW):
                j += 1
            calls, rr = [], []
            for seg, x, _, _ in terms[i:j]:
                call, r = seg.weight.fwd_prepare(x, seg.relu_in, None, seg.norm is not None)
                calls.append(call)
                rr.append(list(r))
            K.conv_fwdN(calls)
            res.extend(rr)
            i = j
        elif i + 1 < len(terms) and isinstance(terms[i][0].weight, DepthSepW) and isinstance(terms[i + 1][0].weight, DepthSepW):
            # a run of depthwise-separable primitives: all depthwise stages in one launch, the 1x1x1 convs two per launch
            j = i
            while j < len(terms) and isinstance(terms[j][0].weight, DepthSepW):
                j += 1
            run = terms[i:j]
            wss = [seg.weight.fwd_depth(x, seg.relu_in, None, launch=False) for seg, x, _, _ in run]
            K.dwconv_batch([ws.job for ws in wss])
            pcs = [seg.weight.point_call(ws, seg.norm is not None) for (seg, _, _, _), ws in zip(run, wss)]
            for k in range(0, len(run) - 1, 2):
                K.conv_fwd2([pcs[k][0], pcs[k + 1][0]])
            if len(run) % 2:
                K.conv_fwd(*pcs[-1][0])
            res.extend([y, stats, rows, ws] for (_, (y, stats, rows)), ws in zip(pcs, wss))
            i = j
        else:
            seg, x, _, _ = terms[i]
            res.append(list(seg.weight.fwd(x, seg.relu_in, None, seg.norm is not None and seg.weight.produces_stats)))
            i += 1
    # the channel statistics still missing (conv outputs whose kernel emitted none, GroupNorm / SE-gate inputs) in one launch
    need = [r[0] for r, (seg, _, _, _) in zip(res, terms) if (seg.norm is not None and r[1] is None) or (seg.norm is None and seg.se_gate is not None)]
    if len(need) >= 2:
        K.channel_statsN(need)   # fills the statistics cache; the per-term calls below hit it
    for r, (seg, _, _, _) in zip(res, terms):
        if (seg.norm is not None or seg.se_gate is not None) and r[1] is None:
            r[1], r[2] = K.channel_stats(r[0])      # (a hit in the statistics cache after the batched launch above)
    return [list(r) for r in res]


def group_epilogue_phase(terms, res, out, accumulate):
    """second half of group_forward: coefficients (all GroupNorm ones in one launch, SE gates) and ONE pass over `out` for up to 8
    terms whose weight phase (`res`, in term order) is complete on the launching stream.  Returns the saved states."""
    saved, gn, se = [], [], []
    for r, (seg, _, _, _) in zip(res, terms):
        s = Saved()
        s.ws, s.raw = r[3], r[0]
        s.a = s.b = s.mr = None
        s.kind = "plain"
        if seg.norm is not None:
            if r[1] is None:
                r[1], r[2] = K.channel_stats(r[0])
            s.kind, s.G = "gn", gn_groups(seg.norm, r[0].C)
            gn.append((s, (r[0], r[1], r[2], seg.norm.weight, seg.norm.bias)))
        elif seg.se_gate is not None:
            s.kind = "se"
            se.append((s, seg.se_gate.fc, r[0], (r[1], r[2]) if r[1] is not None else None))
        saved.append(s)
    merged = False
    if NODE_FWD_COEFFS and gn and se and len({(raw.B, raw.C, raw.N) for _, _, raw, _ in se} | {(g[1][0].B, g[1][0].C, g[1][0].N) for g in gn}) == 1:
        # GroupNorm coefficients and SE gates of the group in ONE launch (the same bodies as gn_coeffsN / se_gate_fwdN)
        eps = [seg.norm.eps for seg, _, _, _ in terms if seg.norm is not None]
        if any(e != eps[0] for e in eps):
            raise N3DError("group_forward: GroupNorm terms of one node with different eps")
        sts = [st if st is not None else K.channel_stats(raw) for _, _, raw, st in se]
        gout, sout = K.node_fwd_coeffs([g[1] for g in gn], gn[0][0].G, eps[0], [(st, rows, fc) for (st, rows), (_, fc, _, _) in zip(sts, se)])
        for (s, _), (a, b, mr, sr) in zip(gn, gout):
            s.a, s.b, s.mr, s.sumraw = a, b, mr, sr
        for (s, _, _, _), (mean, hidden, gate) in zip(se, sout):
            s.mean, s.hidden, s.a = mean, hidden, gate
        merged = True
    if merged:
        pass
    elif len(se) == 1:
        s, fc, raw, st = se[0]
        if st is None:
            st = K.channel_stats(raw)
        s.mean, s.hidden, s.a = K.se_gate_fwd(st[0], st[1], raw.N, fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, raw.B, raw.C)
    elif se:
        raw0 = se[0][2]
        sts = [st if st is not None else K.channel_stats(raw) for _, _, raw, st in se]
        for (s, _, _, _), (mean, hidden, gate) in zip(se, K.se_gate_fwdN([(st, rows, fc) for (st, rows), (_, fc, _, _) in zip(sts, se)], raw0.N, raw0.B, raw0.C)):
            s.mean, s.hidden, s.a = mean, hidden, gate
    if gn and not merged:
        eps = [seg.norm.eps for seg, _, _, _ in terms if seg.norm is not None]
        if any(e != eps[0] for e in eps):
            raise N3DError("group_forward: GroupNorm terms of one node with different eps")
        for (s, _), (a, b, mr, sr) in zip(gn, K.gn_coeffsN([g[1] for g in gn], gn[0][0].G, eps[0])):
            s.a, s.b, s.mr, s.sumraw = a, b, mr, sr
    tl = [(s.raw, s.a, s.b, _wptr(arow, col) if arow is not None else None, seg.relu_out) for s, (seg, _, arow, col) in zip(saved, terms)]
    K.affine_actN(tl, out, ACCUMULATE if accumulate else 0)
    return saved


def group_prepare(terms, dout):
    """the descriptors of an N-term group's backward, ahead of its phases: terms = [(segment, saved, (alpha row, column, dalpha row |
    None))] in forward order.  Returns a K.GnGroupBwd whose reduce / coeffs phases the caller runs (K.node_bwd_prologue: shared with
    the node's other primitives) before it hands it to group_backward."""
    tds = [_gn_bwd_term(seg, s, al if al is not None else (None, 0, None)) for seg, s, al in terms]
    return K.GnGroupBwd(dout, tds, terms[0][1].G)


def group_backward(terms, dout, prepared=None):
    """Backward of group_forward.  terms = [(segment, saved, (need_dx, dx_out, dx_acc), (alpha row, column, dalpha row | None))] in
    forward order; dout: View of the node gradient all of them consume.  prepared: group_prepare's object for the same terms, its
    reductions and coefficients done.  Returns [(dx, grads)] in forward order."""
    if prepared is not None:
        prepared.apply()
        tds, outs = prepared.terms, prepared.outs
    else:
        tds = [_gn_bwd_term(seg, s, al if al is not None else (None, 0, None)) for seg, s, _, al in terms]
        outs = K.affine_act_bwd_gnN(dout, tds, terms[0][1].G)
    order = [(seg, s, td, o, args) for (seg, s, args, _), td, o in zip(terms, tds, outs)][::-1]
    return _weight_backward(order)[::-1]


def needs_reduce(seg, s, dalpha):
    """does seg_backward start with the reduction pass over (dout, raw)?  (callers may batch those passes: pre_sums)"""
    return s.kind in ("gn", "se") or dalpha is not None


def _finish_gn_backward(seg, s, draw, need_dx, dx_out, dx_acc, cbias, dcb, dgamma, dbeta):
    """weight-op backward behind a GroupNorm epilogue backward that has produced d(raw) and the norm's parameter gradients"""
    dx, wg = seg.weight.bwd(s.ws, draw, need_dx, dx_out, dx_acc, dcb is not None)
    wg = list(wg)
    if dcb is not None:
        # the conv bias gradient came out of the GroupNorm backward sums: put it in the bias slot
        plist = seg.weight.params()
        for i, p in enumerate(plist):
            if p is cbias:
                wg[i] = dcb
    return dx, wg + [dgamma, dbeta]


def seg_backward(seg, s, dout, need_dx=True, dx_out=None, dx_acc=False, alpha_row=None, alpha_k=0, dalpha=None, pre_sums=None, pre_se=None,
                 pre_dalpha=False):
    """Backward of one segment.  dout: View of d(out).  Returns (dx tensor | None, [param grads])
    with param grads ordered like seg.params().  If `dalpha` (a float tensor) is given,
    dalpha[alpha_k] = <dout, z> is written (MixedOp architecture gradient, cell.py:29-32).
    pre_sums = (sums, rows): the reduction pass was already done (K.affine_act_bwd_reduceN over several terms of a node);
    pre_se = (dw1, db1, dw2, db2, A, Bc): so was the SE gate backward (K.se_gate_bwdN); pre_dalpha: so was dalpha of an
    un-normalised primitive (K.plain_dalphaN)."""
    if s.kind == "gn_rc":
        return _seg_backward_recompute(seg, s, dout, need_dx)
    wp = _wptr(alpha_row, alpha_k)
    dap = C.c_void_p(dalpha.data_ptr() + 4 * alpha_k) if dalpha is not None else None
    raw = s.raw
    fl = RELU if seg.relu_out else 0
    extra = []
    if s.kind == "gn":
        cbias = seg.weight.norm_fed_bias() if ANALYTIC_CONV_BIAS else None
        if cbias is not None and not cbias.requires_grad:
            cbias = None
        ident = isinstance(seg.weight, IdentityW)
        if ident and not need_dx:
            ident = False
        sums, rows = pre_sums if pre_sums is not None else K.affine_act_bwd_reduce(dout, raw, s.a, s.b, fl)
        if ident:
            # the raw tensor is the input itself: the apply pass writes dx directly
            if dx_out is None:
                dx_out = K.like(raw)
                dx_acc = False
            target, tfl = dx_out, fl | (ACCUMULATE if dx_acc else 0)
        else:
            target, tfl = K.like(raw), fl
        if rows <= K.fused_max_rows() and raw.B <= 4:
            dgamma, dbeta, dcb = K.affine_act_bwd_apply_gn(dout, raw, s.a, s.b, sums, rows, seg.norm.weight, seg.norm.bias, s.mr, wp,
                                                           s.sumraw, cbias, target, s.G, tfl, dap)
        else:
            dgamma, dbeta, A, Bc, Cc, dcb = K.gn_bwd_coeffs(sums, rows, seg.norm.weight, s.mr, wp, raw.B, raw.C, s.G, raw.N, dap,
                                                            seg.norm.bias, s.sumraw, cbias)
            K.affine_act_bwd_apply(dout, raw, s.a, s.b, A, Bc, Cc, target, tfl)
        if ident:
            return dx_out.t, [dgamma, dbeta]
        return _finish_gn_backward(seg, s, target, need_dx, dx_out, dx_acc, cbias, dcb, dgamma, dbeta)
    elif s.kind == "se":
        sums, rows = pre_sums if pre_sums is not None else K.affine_act_bwd_reduce(dout, raw, s.a, None, 0)
        fc = seg.se_gate.fc
        if pre_se is not None:
            dw1, db1, dw2, db2, A, Bc = pre_se
        else:
            dw1, db1, dw2, db2, A, Bc = K.se_gate_bwd(sums, rows, wp, s.mean, s.hidden, s.a, fc[0].weight, fc[2].weight,
                                                      raw.B, raw.C, raw.N, dap, fc)
        dx = None
        if need_dx:
            if dx_out is None:
                dx_out = K.like(raw)
                dx_acc = False
            K.affine_act_bwd_apply(dout, raw, None, None, A, Bc, None, dx_out, ACCUMULATE if dx_acc else 0)
            dx = dx_out.t
        return dx, [dw1, db1, dw2, db2]
    else:
        if isinstance(seg.weight, PoolW) and not seg.relu_out:
            # pooling: dx (+)= w * pool^T(dout) in the pooling backward itself; dalpha = <dout, pooled> from the reduction rows
            if dap is not None and not pre_dalpha:
                sums, rows = pre_sums if pre_sums is not None else K.affine_act_bwd_reduce(dout, raw, None, None, 0)
                K.plain_bwd_coeffs(sums, rows, wp, raw.B, raw.C, raw.t.device, dap, want_A=False)
            dx, wg = seg.weight.bwd(s.ws, dout, need_dx, dx_out, dx_acc, scale=wp)
            return dx, list(wg) + extra
        if seg.relu_out or wp is not None or dap is not None:
            sums, rows = (None, 0)
            if dap is not None:
                sums, rows = pre_sums if pre_sums is not None else K.affine_act_bwd_reduce(dout, raw, None, None, fl)
            A = K.plain_bwd_coeffs(sums, rows, wp, raw.B, raw.C, raw.t.device, dap, want_A=True)
            draw = K.like(raw)
            K.affine_act_bwd_apply(dout, raw, None, None, A, None, None, draw, fl)
        else:
            # d(raw) IS the incoming gradient buffer, which the backward chain may go on accumulating into: no queued launch on it
            with K.no_defer():
                dx, wg = seg.weight.bwd(s.ws, dout, need_dx, dx_out, dx_acc)
            return dx, list(wg) + extra
    dx, wg = seg.weight.bwd(s.ws, draw, need_dx, dx_out, dx_acc)
    return dx, list(wg) + extra


# =================================================================================================
# op-level autograd Function (one reference BaseOp call = one autograd node per segment)
# =================================================================================================
class SegmentFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, drop_gate, x, *params):
        xv = K.as_view(x, "input")
        out, s = seg_forward(seg, xv, drop_gate)
        ctx.seg, ctx.s = seg, s
        ctx.nparams = len(params)
        return out.t

    @staticmethod
    def backward(ctx, dout):
        seg, s = ctx.seg, ctx.s
        dv = K.as_view(dout, "grad_output")
        dx, grads = seg_backward(seg, s, dv, need_dx=ctx.needs_input_grad[2])
        plist = seg.params()
        out = []
        for i, p in enumerate(plist):
            g = grads[i] if i < len(grads) else None
            if getattr(p, "_n3d_grad", None) is not None:
                g = None  # already written in place into the trainer's flat gradient buffer
            out.append(g if ctx.needs_input_grad[3 + i] else None)
        ctx.s = None
        return (None, None, dx) + tuple(out)


# Dropout3d (prim_ops.py:66,72-73): a (B, C) gate of 0 / 1/(1-p) folded into the conv as an input scale.  The gate is drawn on
# the device by n3d_dropout3d_gate from a counter-based generator whose state (seed, step counter) lives in device memory next
# to the nn.Dropout3d module, so a captured HIP graph draws a fresh mask on every replay and no torch RNG kernel runs.
_forced_gate = None
_drop_serial = [0]
PASS_TAG = None      # "arch" / "weight" while a SearchTrainer pass runs (a forced gate may differ between the two)


class forced_dropout_gate:
    """with forced_dropout_gate(g): every Dropout3d inside uses the given (B, C) gate tensor (parity tests feed the mask
    the CPU oracle used).  g = None: dropout off.  g = {"arch": ga, "weight": gw}: the search step's two passes use one each."""

    def __init__(self, gate):
        self.gate = gate

    def __enter__(self):
        global _forced_gate
        self.prev, _forced_gate = _forced_gate, (self.gate,)
        return self

    def __exit__(self, *exc):
        global _forced_gate
        _forced_gate = self.prev
        return False


def dropout_state(drop, device, seed=None):
    """device int32[3] {seed_lo, seed_hi, counter} of an nn.Dropout3d, created on first use from torch's seed, the process
    rank (data-parallel ranks must not share masks) and a per-module serial number; seed != None re-seeds it"""
    st = getattr(drop, "_n3d_state", None)
    if st is None or st.device != device or seed is not None:
        if seed is None:
            import torch.distributed as dist
            rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
            _drop_serial[0] += 1
            seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * (rank * 4096 + _drop_serial[0])) & 0xFFFFFFFFFFFFFFFF
        words = [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, 0]
        st = torch.tensor([w - (1 << 32) if w >= (1 << 31) else w for w in words], dtype=torch.int32, device=device)
        drop._n3d_state = st
    return st


def draw_gate(drop, training, B, Cc, device):
    """the Dropout3d gate of this call: None (inactive), the forced one, or a fresh draw"""
    if _forced_gate is not None:
        g = _forced_gate[0]
        return g[PASS_TAG] if isinstance(g, dict) else g
    if drop is None or not training or drop.p <= 0:
        return None
    return K.dropout3d_gate(dropout_state(drop, device), drop.p, B, Cc)


def run_segment(seg, x, training):
    gate = None
    if seg.dropout is not None:
        gate = draw_gate(seg.dropout, training, x.shape[0], x.shape[1], x.device)
    return SegmentFn.apply(seg, gate, x, *seg.params())
