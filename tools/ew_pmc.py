#!/usr/bin/env python3
"""PMC / trace target: the three node-epilogue streaming kernels of a searched-cell node (n3d_affine_act2, n3d_affine_act_bwd_reduce2,
n3d_affine_act_bwd_apply2) at (B, N = S^3, C = 4), `iters` launches each, with the node (and its gradient) either a DENSE tensor
(node-planar last cell, round 4) or a 16-byte channel SLICE of a 48-byte-pitch concatenation buffer (round 3).
   usage: ew_pmc.py <dense|slice> <S> <B> <iters>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from nas_3d_unet_amd import kernels as K, _lib

layout, s, b, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda")
c = 4
mk = lambda: K.as_view(K.empty_ndhwc(b, c, s, s, s, dev).normal_())
raw0, raw1, draw0, draw1 = mk(), mk(), mk(), mk()
if layout == "dense":
    node, dnode = mk(), mk()
else:
    buf, dbuf = (K.as_view(K.empty_ndhwc(b, 3 * c, s, s, s, dev).normal_()) for _ in range(2))
    node, dnode = K.View(buf.t[:, c:2 * c], buf.ld), K.View(dbuf.t[:, c:2 * c], dbuf.ld)
N = s ** 3
lib = _lib.load()
a = [torch.randn(b, c, device=dev) for _ in range(10)]
rows = K.stats_rows(N, c)
sums = [torch.empty((b, rows, c, 3), dtype=torch.float64, device=dev) for _ in range(2)]


def fwd_terms():
    return [_lib.GnFwdTerm(r.p.value, r.ld, None, 0, 1, None, None, None, a[2 * i].data_ptr(), a[2 * i + 1].data_ptr(), None, None, 0, 0) for i, r in enumerate((raw0, raw1))]


def bwd_terms():
    out = []
    for i, (r, d) in enumerate(((raw0, draw0), (raw1, draw1))):
        out.append(_lib.GnBwdTerm(r.p.value, r.ld, a[2 * i].data_ptr(), a[2 * i + 1].data_ptr(), sums[i].data_ptr(), rows, 1, None, None, None, None,
                                  d.p.value, d.ld, None, None, None, None, a[4 + 3 * i].data_ptr(), a[5 + 3 * i].data_ptr(), a[6 + 3 * i].data_ptr(), 0, 0))
    return out


sp = K.stream_ptr()
for _ in range(iters):
    t = fwd_terms()
    _lib.check(lib.n3d_affine_act2(C.byref(t[0]), C.byref(t[1]), node.p, node.ld, None, 0, b, N, c, 0, sp), "act2")
    t = bwd_terms()
    _lib.check(lib.n3d_affine_act_bwd_reduce2(dnode.p, dnode.ld, None, 0, C.byref(t[0]), C.byref(t[1]), b, N, c, sp), "reduce2")
    _lib.check(lib.n3d_affine_act_bwd_apply2(dnode.p, dnode.ld, None, 0, C.byref(t[0]), C.byref(t[1]), b, N, c, sp), "apply2")
torch.cuda.synchronize()
print("done", layout, s, b, iters)
