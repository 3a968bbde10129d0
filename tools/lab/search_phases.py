"""search step with the side schedule: GPU time of the architecture-pass graph, the weight pass' main graph and its tail"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    (ga_main, ga_side, ga_tail), (gw_main, gw_side, gw_tail) = tr._side_graphs
    n = 12
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        evs[i][0].record(); tr.side.raw_replay(ga_side); ga_main.replay(); evs[i][1].record(); ga_tail.replay(); evs[i][2].record()
        tr.side.raw_replay(gw_side); gw_main.replay(); evs[i][3].record(); gw_tail.replay(); evs[i][4].record()
    torch.cuda.synchronize()
    f = lambda a, b: sum(e[a].elapsed_time(e[b]) for e in evs[2:]) / (n - 2)
    print("drop_side=%s fwd_side=%s: arch pass main %.3f + tail %.3f ms, weight pass main %.3f + tail %.3f ms, sum %.3f; hand-offs %d" % (
        drop, (tr.side_forward, tr.side.split), f(0, 1), f(1, 2), f(2, 3), f(3, 4), f(0, 4), int((tr.side.sync[8:8 + tr.side.JOIN] > 0).sum())), flush=True)
    tr.check_sync()
run(False)
