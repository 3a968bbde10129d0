import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import numpy as np, torch
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer
from oracle import ref_path as orc
from _util import fill_module, dev
from test_gpu_nets import _genotype_for
cfg = orc.NetCfg(4, 6, 3, 2, 3, True)
gene = _genotype_for(cfg.n_nodes)
net = searched.SearchedNet(4, 6, 3, 2, 3, True, searched.Genotype(list(gene.down), list(gene.up)))
fill_module(net); net.last_conv[0].dropout = None; net = net.cuda()
rng = np.random.default_rng(23)
xn = rng.standard_normal((2, 4, 16, 16, 32)).astype(np.float32)
tn = (rng.uniform(0, 1, (2, 3, 16, 16, 32)) < 0.3).astype(np.float32)
P = orc.make_params(orc.searched_param_specs(cfg, gene), dtype=torch.float64, requires_grad=True)
w0 = {n: q.detach().clone() for n, q in P.items()}
opt = torch.optim.Adam(list(P.values()))
opt.zero_grad(); l = orc.dice_loss(orc.searched_forward(P, torch.from_numpy(xn).double(), gene, cfg), torch.from_numpy(tn).double()); l.backward(); opt.step()
g0 = {n: q.grad.clone() for n, q in P.items()}
tr = Trainer(net, graph=False, side_wgrad=False)
tp0 = {n: p.detach().clone() for n, p in tr.net.named_parameters()}
l1 = float(tr.step(dev(xn), dev(tn)))
torch.cuda.synchronize()
print("loss", l1, float(l))
tp = dict(tr.net.named_parameters())
bad = []
for n in tr._twin.names:
    moved = (tp[n].detach() - tp0[n])
    real_moved = tr._twin.extract(n, moved).cpu().double()
    pad_moved = float(moved.abs().sum()) - float(tr._twin.extract(n, moved).abs().sum())
    ref_moved = (P[n].detach() - w0[n])
    gmag = float(g0[n].abs().max())
    # compare only where the oracle's gradient is clearly non-noise
    sel = g0[n].abs() > 1e-9
    d = float((real_moved - ref_moved)[sel].abs().max()) if bool(sel.any()) else 0.0
    if pad_moved > 1e-9 or d > 2e-4:
        bad.append((n, pad_moved, d, gmag, tuple(tp[n].shape)))
for b in bad[:40]: print("  %-50s pad moved %.3e  real delta err %.3e  |g|max %.2e  %s" % b)
print(len(bad), "of", len(tr._twin.names))
