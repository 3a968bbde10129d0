"""probe: which weight-gradient groups (counted from the end of the backward walk) go to the INLINE side stream instead of the weight-gradient
stream (SideSchedule.tail_inline), by patch size:   python tools/tail_inline_probe.py [size] [f32|bf16]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer, reserve_side_streams
dev = torch.device("cuda")
reserve_side_streams(dev)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
storage = "bf16" if len(sys.argv) > 2 and sys.argv[2] == "bf16" else None
def run(ti):
    torch.manual_seed(1234)
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
    tr = Trainer(net, graph=True, side_wgrad="force", storage=storage)
    tr.side.tail_inline = tuple(ti)
    xn, tn = bench.synthetic_batch(2, size, 1234)
    x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
    for _ in range(5): tr.step(x, t)
    x, t = tr.input_buffers()
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 30
        for _ in range(n): tr.step(x, t)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    print("tail_inline %-22s %.3f ms per step  (time-outs %d)" % (str(tuple(ti)), best, int(tr.side.sync[1].item())), flush=True)
for ti in [(2,), (), (1,), (2, 4), (1, 3), (2, 4, 6), (1, 3, 5, 7), (2, 4, 6, 8, 10), (3, 6, 9), (2, 5, 8, 11)]:
    run(ti)
