import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import bench, kernel_table as kt
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda")
torch.manual_seed(1)
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
tr = Trainer(net, graph=False, storage="bf16")
xn, tn = bench.synthetic_batch(2, 128, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
for _ in range(3): tr.step(x, t)
with kt.Recorder() as rec:
    tr._eager(x, t)
torch.cuda.synchronize()
for name, args in rec.calls:
    if name == "n3d_conv_fwd2":
        cs = [kt._struct(args[0]), kt._struct(args[1])]
        if any(c.g.contents.Ci == 16 and c.g.contents.Di == 32 for c in cs):
            for c in cs:
                g = c.g.contents
                print("  C%d->%d %d^3 s%d d%d flags %d xld %d yld %d x%%16 %d y%%16 %d gate %s stats %s bias %s ws_bytes %d" % (g.Ci, g.Co, g.Di, g.stride, g.dil, c.flags, c.xld, c.yld,
                      (c.x or 0) % 16, (c.y or 0) % 16, bool(c.in_gate), bool(c.stats), bool(c.bias), c.ws_bytes))
            us = kt._time_call(name, args, dev)
            print("  -> %.1f us" % us)
import ctypes as C
from nas_3d_unet_amd import _lib, kernels as K
lib = _lib.load()
done = False
for name, args in rec.calls:
    if name == "n3d_conv_fwd2" and not done:
        cs = [kt._struct(args[0]), kt._struct(args[1])]
        if cs[0].g.contents.Ci == 16 and cs[0].g.contents.Di == 32 and cs[0].g.contents.stride == 1:
            done = True
            for c in cs:
                g = c.g.contents
                a = [C.byref(g), C.c_void_p(c.x), c.xld, C.c_void_p(c.w), C.c_void_p(c.bias), C.c_void_p(c.y), c.yld, c.flags, C.c_void_p(c.in_gate), C.c_void_p(c.stats),
                     C.c_void_p(c.ws), c.ws_bytes, None]
                print("  single s%d d%d: %.1f us" % (g.stride, g.dil, kt._time_call("n3d_conv_fwd", a, dev)))
                a[9] = C.c_void_p(None)
                print("  single s%d d%d without stats: %.1f us" % (g.stride, g.dil, kt._time_call("n3d_conv_fwd", a, dev)))
                a[4] = C.c_void_p(None)
                print("  single s%d d%d without stats, bias: %.1f us" % (g.stride, g.dil, kt._time_call("n3d_conv_fwd", a, dev)))
                # same geometry on fresh random tensors
                xx = K.as_view(K.empty_ndhwc(2, 16, g.Di, g.Hi, g.Wi, dev, torch.float32).normal_())
                yy = K.as_view(K.empty_ndhwc(2, 16, g.Do, g.Ho, g.Wo, dev, torch.float32))
                a2 = list(a); a2[1] = xx.p; a2[5] = yy.p
                print("  ... fresh x / y: %.1f us" % kt._time_call("n3d_conv_fwd", a2, dev))
                xr = torch.empty(2 * g.Di * g.Hi * g.Wi * 16, device=dev); 
                src = (C.c_float * 1)
                import numpy as np
                xt = torch.frombuffer((C.c_char * 0)(), dtype=torch.uint8) if False else None
