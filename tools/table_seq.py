"""Per-launch table of one train step IN EXECUTION ORDER (replay-timed per (entry, shape) group).  usage: table_seq.py [size] [batch] [storage]"""
import sys, os, json, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import bench, kernel_table
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda")
torch.manual_seed(1)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
storage = sys.argv[3] if len(sys.argv) > 3 else None
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
tr = Trainer(net, graph=False, storage=storage)
xn, tn = bench.synthetic_batch(batch, size, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
for _ in range(3): tr.step(x, t)
torch.cuda.synchronize()
with kernel_table.Recorder() as rec:
    tr._eager(x, t)
    torch.cuda.synchronize()
times = {}
tot = 0.0
for i, (name, args) in enumerate(rec.calls):
    sig, flop, byts = kernel_table.describe(name, args)
    if sig not in times:
        try:
            times[sig] = kernel_table._time_call(name, args, dev)
        except Exception as e:
            times[sig] = float("nan")
    us = times[sig]
    tot += us
    print("%3d %7.2f  %-30s %s" % (i, us, name, kernel_table._shape_text(sig)[:110]))
print("launches", len(rec.calls), "sum us %.1f" % tot)
