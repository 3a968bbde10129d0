#!/usr/bin/env python3
"""Lists the GroupNorm epilogue calls of one searched-net train step: shape, statistics rows, fused-prologue or not."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from nas_3d_unet_amd import kernels as K, searched
from nas_3d_unet_amd.train import Trainer

dev = torch.device("cuda")
cnt = collections.Counter()
f2, b2, f1 = K.affine_act_gn2, K.affine_act_bwd_gn2, K.affine_act_gn


def fwd2(terms, G, eps, out, flags=0, out1=None):
    r = terms[0][0]
    cnt[("fwd2", r.B, r.C, r.N, terms[0][2], terms[1][2], K.pair_ok(r.C, G, terms[0][2], terms[1][2], r.B))] += 1
    return f2(terms, G, eps, out, flags, out1)


def bwd2(dout, terms, G, dout1=None):
    r = terms[0]["raw"]
    rows = K.stats_rows(r.N, r.C)
    cnt[("bwd2", r.B, r.C, r.N, rows, rows, K.pair_ok(r.C, G, rows, rows, r.B))] += 1
    return b2(dout, terms, G, dout1)


def fwd1(raw, stats, rows, *a, **k):
    cnt[("fwd1", raw.B, raw.C, raw.N, rows, 0, True)] += 1
    return f1(raw, stats, rows, *a, **k)


K.affine_act_gn2, K.affine_act_bwd_gn2, K.affine_act_gn = fwd2, bwd2, fwd1
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
print("kind B C N rows0 rows1 fused : count")
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[0][3])):
    print(k, v)
