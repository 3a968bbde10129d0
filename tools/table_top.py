"""Top entry points of one step by microseconds per step, each against its roofline (kernel_table.table with a long list).
usage: table_top.py train|search [size] [storage]"""
import os, sys, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import bench, kernel_table
from nas_3d_unet_amd import nas, searched
from nas_3d_unet_amd.train import SearchTrainer, Trainer
dev = torch.device("cuda")
torch.manual_seed(1)
which = sys.argv[1] if len(sys.argv) > 1 else "train"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
storage = sys.argv[3] if len(sys.argv) > 3 else None
xn, tn = bench.synthetic_batch(2, size, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
if which == "search":
    net = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); net.train()
    tr = SearchTrainer(net, graph=False, side_wgrad=False)
    step = lambda: tr._both(x, t, x, t)
else:
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
    tr = Trainer(net, graph=False, storage=storage, side_wgrad=False)
    step = lambda: tr._eager(x, t)
for _ in range(2): step()
rows, n = kernel_table.table(step, dev, top=45, candidates=60)
tot = 0.0
for r in rows:
    tot += r["us_per_step"]
    print("%-34s %-70s x%-3d %7.2f us/call %8.1f us/step  %s %s" % (r["entry"], r["shape"][:70], r["calls_per_step"], r["us_per_call"], r["us_per_step"],
                                                                   r.get("bound", ""), r.get("frac", "")))
print("launches per step", n, " listed us/step %.0f" % tot)
