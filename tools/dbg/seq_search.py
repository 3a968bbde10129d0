"""execution-order launch list of the supernet weight pass (names + coarse shapes)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch, bench, kernel_table
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
torch.cuda.synchronize()
with kernel_table.Recorder() as rec:
    tr._pass(x, t, False, update=False)
    torch.cuda.synchronize()
for i, (name, args) in enumerate(rec.calls):
    sig, _, _ = kernel_table.describe(name, args)
    print(i, name, kernel_table._shape_text(sig)[:100])
