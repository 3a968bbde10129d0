#!/usr/bin/env python3
"""Sweep of batch sizes / patch shapes / genotypes through the trainers (eager and graph) on the GPU: no exceptions,
finite losses, eager == graph.  Not a parity test (tests/ does that) -- a crash / limit finder."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as Bn
from nas_3d_unet_amd import searched, nas
from nas_3d_unet_amd.train import Trainer, SearchTrainer
from oracle import ref_path as orc

dev = torch.device("cuda")
G_ALL = dict(down=list(orc.G_ALL.down), up=list(orc.G_ALL.up))


def batch(B, shape, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, 4) + shape).astype(np.float32)
    t = (rng.uniform(0, 1, (B, 3) + shape) < 0.3).astype(np.float32)
    return torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)


def run_train(gname, gene, B, shape, depth=4):
    out = []
    for graph in (False, True):
        torch.manual_seed(7)
        net = searched.SearchedNet(4, 4, 3, depth, 3, True, searched.Genotype(**gene)).to(dev)
        net.train()
        net.last_conv[0].dropout = None  # deterministic comparison
        tr = Trainer(net, graph=graph)
        x, t = batch(B, shape, 3)
        ls = [float(tr.step(x, t)) for _ in range(3)]
        out.append(ls)
    ok = np.allclose(out[0], out[1], atol=2e-5) and np.all(np.isfinite(out[0]))
    print("%-7s B=%d %-14s eager %s graph %s %s" % (gname, B, shape, ["%.5f" % v for v in out[0]], ["%.5f" % v for v in out[1]], "OK" if ok else "MISMATCH"))
    return ok


ok = True
for gname, gene in (("G_conv", Bn.G_CONV), ("G_all", G_ALL)):
    for B, shape in ((1, (64, 64, 64)), (3, (64, 64, 64)), (4, (32, 32, 32)), (5, (32, 32, 32)), (2, (64, 32, 96)), (1, (32, 64, 32)), (7, (32, 32, 32))):
        ok &= run_train(gname, gene, B, shape)
# supernet search step, odd batch
for B, shape in ((1, (32, 32, 32)), (3, (32, 32, 64))):
    res = []
    for graph in (False, True):
        torch.manual_seed(9)
        net = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev)
        net.kernel.last_conv[0].dropout = None
        tr = SearchTrainer(net, graph=graph)
        x, t = batch(B, shape, 5); vx, vt = batch(B, shape, 6)
        res.append([tuple(float(v) for v in tr.step(x, t, vx, vt)) for _ in range(2)])
    good = np.allclose(np.array(res[0]), np.array(res[1]), atol=5e-5) and np.all(np.isfinite(np.array(res[0])))
    print("search  B=%d %-14s %s %s" % (B, shape, res[0], "OK" if good else "MISMATCH"))
    ok &= good
print("ALL OK" if ok else "FAILURES")
