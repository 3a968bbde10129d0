"""side-stream schedule: time of the main graph (under contention from the side stream) and of the tail (join wait + slab reduction + Adam)"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
from nas_3d_unet_amd import searched, kernels as K
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda")
xn, tn = bench.synthetic_batch(2, 64, 1234)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
def run(drop, mq=int(os.environ.get("MQ", "3"))):
    K._DROP_SIDE = drop
    torch.manual_seed(1234)
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
    tr = Trainer(net, graph=True, side_wgrad=True)
    tr.side.min_queue = mq
    for _ in range(5): tr.step(x, t)
    K._DROP_SIDE = False
    g_main, g_side, g_tail = tr._side_graphs
    n = 30
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        evs[i][0].record()
        tr.side.raw_replay(g_side)
        g_main.replay()
        evs[i][1].record()
        g_tail.replay()
        evs[i][2].record()
    torch.cuda.synchronize()
    a = sum(e[0].elapsed_time(e[1]) for e in evs[5:]) / (n - 5)
    b = sum(e[1].elapsed_time(e[2]) for e in evs[5:]) / (n - 5)
    mx = max(e[0].elapsed_time(e[2]) for e in evs[5:])
    print("drop_side=%s min_queue=%d prio=%s: main graph %.3f ms, tail %.3f ms, sum %.3f, worst step %.3f" % (drop, mq, os.environ.get("N3D_SIDE_PRIORITY", "low"), a, b, a + b, mx), flush=True)
    tr.check_sync()
if os.environ.get("N3D_MAIN_NB"):
    # everything on a non-blocking stream instead of the legacy default stream (a CU-masked side stream is a BLOCKING stream)
    nb = torch.cuda.Stream(device=dev, priority=-1 if os.environ.get("N3D_MAIN_NB") == "high" else 0)
    with torch.cuda.stream(nb):
        run(True)
        run(False)
else:
    run(True)
    run(False)
