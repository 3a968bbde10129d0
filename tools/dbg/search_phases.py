"""search step with the side schedule: GPU time of the architecture-pass graph, the weight pass' main graph and its tail"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch, bench
from nas_3d_unet_amd import nas, kernels as K
from nas_3d_unet_amd.train import SearchTrainer
dev = torch.device("cuda")
def run(drop):
    K._DROP_SIDE = drop
    torch.manual_seed(1234)
    net = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); net.train()
    tr = SearchTrainer(net, graph=True, side_wgrad="force")
    xn, tn = bench.synthetic_batch(2, 64, 1234); vxn, vtn = bench.synthetic_batch(2, 64, 4321)
    x, t, vx, vt = (torch.from_numpy(a).to(dev) for a in (xn, tn, vxn, vtn))
    x, vx = bench.to_patch_layout(x), bench.to_patch_layout(vx)
    for _ in range(4): tr.step(x, t, vx, vt)
    K._DROP_SIDE = False
    g_arch, g_main, g_side, g_tail = tr._side_graphs
    n = 12
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        evs[i][0].record(); g_arch.replay(); evs[i][1].record()
        with torch.cuda.stream(tr.side.stream): g_side.replay()
        g_main.replay(); evs[i][2].record(); g_tail.replay(); evs[i][3].record()
    torch.cuda.synchronize()
    f = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in evs[2:]) / (n - 2)
    print("drop_side=%s: arch pass %.3f ms, weight pass main %.3f ms, tail %.3f ms, sum %.3f; cuts %d" % (drop, f(0, 1), f(1, 2), f(2, 3), f(0, 3), int((tr.side.sync[8:108] > 0).sum())), flush=True)
    tr.check_sync()
run(True)
run(False)
