"""Search step under different backward side-stream splits: which preprocess-fed edges of every supernet cell the inline side stream
takes in the architecture / weight pass (SearchTrainer.side_backward_inputs).   python tools/lab/search_side_inputs.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from nas_3d_unet_amd import nas
from nas_3d_unet_amd.train import SearchTrainer

dev = torch.device("cuda", 0)
CFG = bench.CFG


def run(inputs, steps=20, warmup=4):
    torch.manual_seed(1234)
    net = nas.ShellNet(CFG["in_channels"], CFG["init_n_kernels"], CFG["out_channels"], CFG["depth"], CFG["n_nodes"], False, CFG["channel_change"]).to(dev)
    net.train()
    tr = SearchTrainer(net, graph=True)
    tr.side_backward_inputs = inputs
    xn, tn = bench.synthetic_batch(2, 64, 1234)
    vxn, vtn = bench.synthetic_batch(2, 64, 4321)
    x, t, vx, vt = (torch.from_numpy(a).to(dev) for a in (xn, tn, vxn, vtn))
    x, vx = bench.to_patch_layout(x), bench.to_patch_layout(vx)
    for _ in range(warmup):
        tr.step(x, t, vx, vt)
    own = tr.input_buffers()
    for dst, src in zip(own, (x, t, vx, vt)):
        dst.copy_(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(*own)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("side_backward_inputs", inputs, ": %.3f ms per step, schedule %s, time-outs %d" % (dt * 1e3, "three streams" if tr._use_side else "single", tr.sync_timeouts()), flush=True)
    del tr


for inp in ((1,), (0, 1), (0,), (), (1,)):
    run(inp)
# round 5 (one box): (1,) 11.03 / (0, 1) 11.52 / (0,) 11.24 / () 12.03 / (1,) 11.06 ms; choosing the inputs per cell kind (down / up cells, via a
# plan -> inputs callable) moved nothing: down (0,) + up (1,) 11.01, every other mix 11.24-11.67
