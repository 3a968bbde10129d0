import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import bench, kernel_table
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda")
torch.manual_seed(1)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
storage = sys.argv[2] if len(sys.argv) > 2 else None
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
tr = Trainer(net, graph=False, storage=storage)
xn, tn = bench.synthetic_batch(2, size, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
for _ in range(3): tr.step(x, t)
rows, n = kernel_table.table(lambda: tr._eager(x, t), dev, top=400, candidates=400)
tot = sum(r["us_per_step"] for r in rows)
print("launches", n, "sum us", tot)
for r in rows:
    print("%7.1f us/step %3d x %6.2f  %-32s %s  frac=%s" % (r["us_per_step"], r["calls_per_step"], r["us_per_call"], r["entry"], r["shape"][:90], r.get("frac")))
