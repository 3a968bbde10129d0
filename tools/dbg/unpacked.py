"""List the conv calls of one train step that pack their weights themselves (no N3D_PREPACKED)."""
import sys, os, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import bench, kernel_table as kt
from nas_3d_unet_amd import searched, _lib
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda:0")
Cf = bench.CFG
net = searched.SearchedNet(Cf["in_channels"], Cf["init_n_kernels"], Cf["out_channels"], Cf["depth"], Cf["n_nodes"], Cf["channel_change"],
                           searched.Genotype(**bench.G_CONV)).to(dev)
net.train()
tr = Trainer(net, graph=False)
xn, tn = bench.synthetic_batch(2, 64, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
tr.step(x, t); tr.step(x, t)
with kt.Recorder() as rec:
    tr.step(x, t)
torch.cuda.synchronize()
cnt = collections.Counter()
def g2t(g): return kt._gtuple(g)
for name, args in rec.calls:
    v = [kt._val(a) for a in args]
    if name in ("n3d_conv_fwd", "n3d_convT_fwd"):
        if not (v[7] & 32): cnt[(name, g2t(kt._geom(args[0])))] += 1
    elif name in ("n3d_conv_bwd_data", "n3d_convT_bwd_data"):
        if not (v[6] & 32): cnt[(name, g2t(kt._geom(args[0])))] += 1
    elif name in ("n3d_conv_bwd_both", "n3d_convT_bwd_both"):
        fd = v[8]
        if not (fd & 32): cnt[(name, g2t(kt._geom(args[0])))] += 1
    elif name in ("n3d_conv_fwd2", "n3d_conv_bwd_both2", "n3d_conv_bwd_data2"):
        for c in (kt._struct(args[0]), kt._struct(args[1])):
            fl = c.flags if name == "n3d_conv_fwd2" else c.flags_data
            if not (fl & 32): cnt[(name, g2t(c.g.contents))] += 1
for k, n in cnt.items(): print(n, k)
print("calls", len(rec.calls))
