"""side-stream schedule: step time and host time per step for several cut densities"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch, bench
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda")
xn, tn = bench.synthetic_batch(2, 64, 1234)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
def run(side, mq):
    torch.manual_seed(1234)
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
    tr = Trainer(net, graph=True, side_wgrad=side)
    tr.side.min_queue = mq
    for _ in range(5): tr.step(x, t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30): tr.step(x, t)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    ng = int((tr.side.sync[8:108] > 0).sum()) if tr._side_graphs else None
    tr.check_sync()
    print("side=%s min_queue=%d graphs=%s: %.3f ms/step, host %.3f ms/step" % (side, mq, ng, tt / 30 * 1e3, th / 30 * 1e3), flush=True)
from nas_3d_unet_amd import kernels as K
K._DROP_SIDE = True
run(True, 3)
K._DROP_SIDE = False
run(False, 0)
for mq in (3, 2):
    run(True, mq)
run(False, 0)
