"""Which torch ops (not n3d entries) does one train / search step issue?  usage: torch_ops.py [train|search]"""
import sys, os, collections, traceback
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from nas_3d_unet_amd import searched, nas
from nas_3d_unet_amd.train import Trainer, SearchTrainer
dev = torch.device("cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "train"
torch.manual_seed(1)
xn, tn = bench.synthetic_batch(2, 64, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
seen = collections.Counter()
where = {}
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        launches = not any(s in name for s in ("view", "empty", "as_strided", "slice", "select", "detach", "alias", "unsqueeze", "squeeze", "expand", "reshape", "permute", "transpose", "_unsafe", "t.default", "size", "stride", "is_", "_local_scalar", "unbind", "split", "narrow", "lift_fresh", "set_"))
        if launches:
            shp = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))
            key = (name, shp)
            seen[key] += 1
            if key not in where:
                st = traceback.extract_stack(limit=8)[:-1]
                where[key] = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st) if "nas_3d_unet_amd" in f.filename or "bench" in f.filename)[:200]
        return func(*args, **(kwargs or {}))
if what == "train":
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
    tr = Trainer(net, graph=False, side_wgrad=False)
    for _ in range(3): tr.step(x, t)
    torch.cuda.synchronize()
    with Log():
        tr._eager(x, t)
else:
    net = nas.NasShell(**bench.CFG, normal_w_share=False, channel_change=True).to(dev) if hasattr(nas, "NasShell") else None
    raise SystemExit("search: use bench.build_search")
torch.cuda.synchronize()
for (name, shp), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d  %-28s %-60s %s" % (n, name, str(shp)[:60], where[(name, shp)]))
print("total launching torch ops:", sum(seen.values()))
