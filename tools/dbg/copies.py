import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch, traceback, collections
import bench
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda")
torch.manual_seed(1)
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
tr = Trainer(net, graph=False)
xn, tn = bench.synthetic_batch(2, 64, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
tr.step(x, t); tr.step(x, t)
sites = collections.Counter()
orig = torch.Tensor.copy_
def spy(self, src, *a, **k):
    st = traceback.extract_stack(limit=6)
    sites[" <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-4:-1])] += 1
    return orig(self, src, *a, **k)
torch.Tensor.copy_ = spy
for name in ("clone", "contiguous", "fill_", "zero_", "add_", "mul_", "div_"):
    o = getattr(torch.Tensor, name)
    def mk(o, name):
        def f(self, *a, **k):
            st = traceback.extract_stack(limit=6)
            sites[name + " " + " <- ".join("%s:%d" % (os.path.basename(q.filename), q.lineno) for q in st[-4:-1])] += 1
            return o(self, *a, **k)
        return f
    setattr(torch.Tensor, name, mk(o, name))
tr.step(x, t)
for k, v in sites.most_common(30):
    print(v, k)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(x, t)
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if "copy" in e.key.lower() or "aten::" in e.key]
for e in sorted(ev, key=lambda e: -e.count)[:25]:
    print(e.count, e.key)
