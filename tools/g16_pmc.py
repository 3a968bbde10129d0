#!/usr/bin/env python3
"""PMC / trace target for the C >= 16 conv family (gemm16 pair / quad / dual kernels, the tile16 kernels they fall back to):
one process, every case `iters` times through the C ABI, exactly as a searched-cell node issues them (forward pair =
n3d_conv_fwd2, backward = n3d_conv_bwd_both2 of the node's two convs).  The cases are told apart afterwards by (kernel
name, grid size): tools/g16_pmc_summary.py.
   usage: g16_pmc.py <iters> [C size ...]      default cases: 16@16 16@32 32@16 32@8 64@8 64@4 64@2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nas_3d_unet_amd import kernels as K

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
args = [int(a) for a in sys.argv[2:]]
cases = [tuple(args[i:i + 2]) for i in range(0, len(args), 2)] or [(16, 16), (16, 32), (32, 16), (32, 8), (64, 8), (64, 4), (64, 2)]
dev = torch.device("cuda")
B = 2


def build(c, size):
    """the two stride-1 convs of a node (dilation 1 and 2) on one level"""
    out = []
    for dil in (1, 2):
        x = K.as_view(K.empty_ndhwc(B, c, size, size, size, dev).normal_())
        y = K.as_view(K.empty_ndhwc(B, c, size, size, size, dev).normal_())
        dx = K.as_view(K.empty_ndhwc(B, c, size, size, size, dev).normal_())
        w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
        b = torch.randn(c, device=dev) * 0.1
        g = K.conv_geom(B, size, size, size, c, c, 3, 1, dil, dil)
        rows = K.conv_stats_rows(g, False, 0, x, y)
        stats = torch.empty((B, max(rows, 1), c, 2), dtype=torch.float64, device=dev) if rows > 0 else None
        out.append(dict(g=g, x=x, y=y, dx=dx, w=w, b=b, dw=torch.zeros_like(w), db=torch.zeros_like(b), stats=stats))
    return out


ctx = K.StepContext(dev)
with K.step_context(ctx):
    built = {cs: build(*cs) for cs in cases}

    def fwd(p):
        K.conv_fwd2([(q["g"], q["x"], q["w"], q["b"], q["y"], 0, None, q["stats"], False) for q in p])

    def bwd(p):
        K.conv_bwd_both2([(q["g"], q["x"], q["y"], q["w"], q["dx"], q["dw"], q["db"], 0, None, None, 0, None, False) for q in p])
        ctx.final.clear(); ctx.keep.clear()

    for p in built.values():
        fwd(p); bwd(p)
    ctx.freeze(); ctx.pack_all()
    torch.cuda.synchronize()
    for cs, p in built.items():
        for _ in range(iters):
            fwd(p)
        for _ in range(iters):
            bwd(p)
        torch.cuda.synchronize()
print("done", cases, iters)
