"""which earlier activity in the process pushes the search trainer's side schedule into the time-sliced mode?"""
import sys, os, time, gc
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch, bench
from nas_3d_unet_amd import nas, searched
from nas_3d_unet_amd.train import SearchTrainer, Trainer
dev = torch.device("cuda")
mode = sys.argv[1]
xn, tn = bench.synthetic_batch(2, 64, 1234); vxn, vtn = bench.synthetic_batch(2, 64, 4321)
x, t, vx, vt = (torch.from_numpy(a).to(dev) for a in (xn, tn, vxn, vtn))
x, vx = bench.to_patch_layout(x), bench.to_patch_layout(vx)
if "pre" in mode:
    from nas_3d_unet_amd.train import SideSchedule
    from nas_3d_unet_amd import kernels as K
    sd0 = SideSchedule(dev, K.StepContext(dev), wgrad_stream=True)      # both side streams exist before anything else
    print("pre-made side streams", sd0.stream is not None, sd0.split)
if "dummy" in mode:
    # only the capture streams and a trivial graph, no trainer
    from nas_3d_unet_amd.train import capture_stream
    s = capture_stream(dev)
    a = torch.zeros(1024, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        a.add_(1.0)
    g.replay(); torch.cuda.synchronize()
if mode.split("+")[0] in ("plain", "side", "force"):
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
    tr0 = Trainer(net, graph=True, side_wgrad={"plain": False, "side": None, "force": "force"}[mode.split("+")[0]])
    for _ in range(5): tr0.step(x, t)
    torch.cuda.synchronize()
    print("first trainer", tr0.schedule_times)
    if "+keep" not in mode:
        del tr0, net
        gc.collect(); torch.cuda.empty_cache()
net = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); net.train()
tr = SearchTrainer(net, graph=True)
for _ in range(3): tr.step(x, t, vx, vt)
torch.cuda.synchronize()
print(mode, "search schedule_times", tr.schedule_times, "split", tr.side.split if tr.side else None)
