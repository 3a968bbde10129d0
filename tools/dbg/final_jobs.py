"""Dump the deferred weight-gradient reduction jobs of one train step (chunks x slab size per job)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from nas_3d_unet_amd import searched, kernels as K
from nas_3d_unet_amd.train import Trainer
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dev = torch.device("cuda:0")
C = bench.CFG
net = searched.SearchedNet(C["in_channels"], C["init_n_kernels"], C["out_channels"], C["depth"], C["n_nodes"], C["channel_change"],
                           searched.Genotype(**bench.G_CONV)).to(dev)
net.train()
tr = Trainer(net, graph=False, storage="bf16" if dt == "bf16" else None)
xn, tn = bench.synthetic_batch(2, size, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
tr.step(x, t)
orig = K.StepContext.flush_final
def dump(self):
    tot = 0
    for j in sorted(self.final, key=lambda j: -j.nchunks * j.ntiles * j.ci_t * j.co_t):
        el = j.ntiles * j.ci_t * j.co_t
        tot += j.nchunks * el
        print("chunks %6d slab %6d (tiles %d %dx%d) Ci %d Co %d taps %d  floats %d" % (j.nchunks, el, j.ntiles, j.ci_t, j.co_t, j.Ci, j.Co, j.taps, j.nchunks * el))
    print("total partial floats", tot, "jobs", len(self.final))
    return orig(self)
K.StepContext.flush_final = dump
tr.step(x, t)
torch.cuda.synchronize()
