import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
import bench, kernel_table
from nas_3d_unet_amd import nas
from nas_3d_unet_amd.train import SearchTrainer
dev = torch.device("cuda")
torch.manual_seed(1)
net = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); net.train()
tr = SearchTrainer(net, graph=False, side_wgrad=False)
xn, tn = bench.synthetic_batch(2, 64, 1); vxn, vtn = bench.synthetic_batch(2, 64, 2)
x, t, vx, vt = (torch.from_numpy(a).to(dev) for a in (xn, tn, vxn, vtn))
x, vx = bench.to_patch_layout(x), bench.to_patch_layout(vx)
for _ in range(2): tr.step(x, t, vx, vt)
# describe fwdN by the geometries inside
orig = kernel_table.describe
def describe(name, args):
    if name in ("n3d_conv_fwdN",):
        n = args[1]
        arr = args[0]
        sig = [name]
        f = b = 0
        for i in range(n):
            c = arr[i]; g = c.g.contents
            ff, bb = kernel_table._conv_cost(g, "fwdT" if c.transposed else "fwd", c.flags)
            sig.append(kernel_table._gtuple(g)); f += ff; b += bb
        return tuple(sig), f, b
    return orig(name, args)
kernel_table.describe = describe
rows, n = kernel_table.table(lambda: tr._both(x, t, vx, vt), dev, top=500, candidates=500)
print("launches", n)
for r in rows:
    print("%7.1f us/step %3d x %6.2f  %-30s %s  frac=%s" % (r["us_per_step"], r["calls_per_step"], r["us_per_call"], r["entry"], r["shape"][:150], r.get("frac")))
