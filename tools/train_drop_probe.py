"""probe: the 64^3 train step with and without its deferred weight-gradient launches (what the side work still costs the chain), and on one stream"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nas_3d_unet_amd import searched, kernels as K
from nas_3d_unet_amd.train import Trainer, reserve_side_streams
dev = torch.device("cuda")
reserve_side_streams(dev)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def run(drop, side):
    K._DROP_SIDE = drop
    torch.manual_seed(1234)
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
    tr = Trainer(net, graph=True, side_wgrad="force" if side else False)
    xn, tn = bench.synthetic_batch(2, size, 1234)
    x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
    for _ in range(5): tr.step(x, t)
    x, t = tr.input_buffers()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): tr.step(x, t)
    torch.cuda.synchronize()
    print("%-28s %-34s %.3f ms per step" % ("side-stream schedule" if side else "single stream", "weight-gradient launches DROPPED" if drop else "full step", (time.perf_counter() - t0) / n * 1e3), flush=True)
    K._DROP_SIDE = False
for side in (True, False):
    for drop in (False, True):
        run(drop, side)
