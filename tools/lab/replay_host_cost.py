#!/usr/bin/env python3
"""Host-side cost of one captured-step replay (hipGraphLaunch of the ~320-kernel step graph) vs its GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as Bn
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer

dev = torch.device("cuda")
if "RANK" in os.environ:  # torchrun + N3D_FORCE_DP=1: the multi-GPU step path (eager all-reduce + Adam after the graph) on one rank
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev if False else None)
torch.manual_seed(0)
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**Bn.G_CONV)).to(dev)
net.train()
tr = Trainer(net, graph=True)
xn, tn = Bn.synthetic_batch(2, 64, 1)
x, t = Bn.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
for _ in range(5):
    tr.step(x, t)
torch.cuda.synchronize()
for n in (1, 20, 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr._graph.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("replays %3d: host enqueue %.3f ms each, until GPU done %.3f ms each" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
for n in (20, 100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps   %3d: host enqueue %.3f ms each, until GPU done %.3f ms each" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))

if tr.dp_path:
    import torch.distributed as dist
    for what in ("allreduce", "update", "both"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            tr._graph.replay()
            if what in ("allreduce", "both"):
                tr._allreduce()
            if what in ("update", "both"):
                tr._update()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("replay + %-9s: host %.3f ms, done %.3f ms per step" % (what, (t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
    dist.destroy_process_group()
