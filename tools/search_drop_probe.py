"""probe: the search step with and without its deferred weight-gradient launches (upper bound of what a faster weight-gradient stream can buy)"""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from nas_3d_unet_amd import nas, kernels as K
from nas_3d_unet_amd.train import SearchTrainer
dev = torch.device("cuda")
def run(drop):
    K._DROP_SIDE = drop
    torch.manual_seed(1)
    shell = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); shell.train()
    tr = SearchTrainer(shell, graph=True, side_wgrad="force")
    bs = []
    for s in range(2):
        xn, tn = bench.synthetic_batch(2, 64, 10 + s)
        bs.append((bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)))
    for _ in range(4): tr.step(bs[0][0], bs[0][1], bs[1][0], bs[1][1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): tr.step(bs[0][0], bs[0][1], bs[1][0], bs[1][1])
    torch.cuda.synchronize()
    print("drop wgrad launches" if drop else "full step", "%.3f ms per search step" % ((time.perf_counter() - t0) / n * 1e3), flush=True)
run(False); run(True)
