"""64^3 train step with the searched cells' conv pairs split over two streams (fused.SIDE_PAIRS*, the shipped form) or left to the
pair entry points (which fold two one-plane-tile convs into one launch since round 5), interleaved in one process / one box."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from nas_3d_unet_amd import searched, fused
from nas_3d_unet_amd.train import Trainer

dev = torch.device("cuda", 0)
CFG = bench.CFG


def run(flags, size=64, steps=40, warmup=5):
    fused.SIDE_PAIRS, fused.SIDE_PAIRS_BOTH, fused.SIDE_PAIRS_BWD = flags
    torch.manual_seed(1234)
    net = searched.SearchedNet(CFG["in_channels"], CFG["init_n_kernels"], CFG["out_channels"], CFG["depth"], CFG["n_nodes"], CFG["channel_change"],
                               searched.Genotype(**bench.G_CONV)).to(dev)
    net.train()
    tr = Trainer(net, graph=True)
    xn, tn = bench.synthetic_batch(2, size, 1234)
    x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
    for _ in range(warmup):
        tr.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(x, t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("SIDE_PAIRS, _BOTH, _BWD =", flags, ": %.3f ms per step (%s)" % (dt * 1e3, "side" if tr._use_side else "plain"), flush=True)
    tr.close()


for _ in range(2):
    for fl in ((True, True, True), (False, False, True), (False, False, False), (True, False, True)):
        run(fl)
# round 5, one box: shipped (True, True, True) 1.636 / 1.640 ms; forward pairs left to the folding pair entry point (False, False, True) 1.641 / 1.641;
# no pair work on the side streams at all 1.694 / 1.691; (True, False, True) 1.645 / 1.644
