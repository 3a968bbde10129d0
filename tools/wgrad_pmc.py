#!/usr/bin/env python3
"""PMC / trace target: the weight gradient of the 3x3x3 stride-1 conv at C channels on (B, C, S^3), fp32 or bf16 storage, `iters` launches.
   usage: wgrad_pmc.py <f32|bf16> <C> <S> <B> <dil> <iters>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nas_3d_unet_amd import kernels as K

dt, c, s, b, dil, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
dev = torch.device("cuda")
tdt = torch.bfloat16 if dt == "bf16" else torch.float32
x = K.as_view(K.empty_ndhwc(b, c, s, s, s, dev, tdt).normal_())
dy = K.as_view(K.empty_ndhwc(b, c, s, s, s, dev, tdt).normal_())
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
dw = torch.empty_like(w)
g = K.conv_geom(b, s, s, s, c, c, 3, 1, dil, dil)
for _ in range(iters):
    K.conv_bwd_weight(g, x, dy, dw, None, 0, None, False)
torch.cuda.synchronize()
print("done", dt, c, s, b, dil, iters)
