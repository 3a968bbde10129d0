#!/usr/bin/env python3
"""Lists the conv geometries one searched-net train step issues (eager, one step) with call counts."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from nas_3d_unet_amd import kernels as K, searched
from nas_3d_unet_amd.train import Trainer

dev = torch.device("cuda")
cnt = collections.Counter()


def wrap(name):
    f = getattr(K, name)
    def g(geom, *a, **k):
        tr = k.get("transposed", a[-1] if a and isinstance(a[-1], bool) else False)
        cnt[(name + ("_T" if tr else ""), geom.B, geom.Di, geom.Ci, geom.Do, geom.Co, geom.k, geom.stride, geom.dil, geom.depthwise)] += 1
        return f(geom, *a, **k)
    setattr(K, name, g)


for n in ("conv_fwd", "conv_bwd_data", "conv_bwd_weight"):
    wrap(n)
import nas_3d_unet_amd.programs as P
torch.manual_seed(0)
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**Bn.G_CONV)).to(dev)
net.train()
tr = Trainer(net, graph=False)
xn, tn = Bn.synthetic_batch(2, 64, 1)
x, t = torch.from_numpy(xn).to(dev), torch.from_numpy(tn).to(dev)
tr.step(x, t)
cnt.clear()
tr.step(x, t)
torch.cuda.synchronize()
print("name B Di Ci Do Co k s d dw : count")
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print(k, v)
