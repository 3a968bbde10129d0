import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import numpy as np, torch
from test_gpu_nets import build_net
from _util import dev
from oracle import ref_path as orc
from nas_3d_unet_amd import loss
for B in (2, 3, 4):
    rng = np.random.default_rng(31)
    xn = rng.standard_normal((B, 4, 32, 32, 32)).astype(np.float32)
    tn = (rng.uniform(0, 1, (B, 3, 32, 32, 32)) < 0.3).astype(np.float32)
    gene = orc.G_ALL
    P = orc.make_params(orc.searched_param_specs(orc.DEFAULT_CFG, gene), requires_grad=True)
    pr = orc.searched_forward(P, torch.from_numpy(xn), gene)
    lr = orc.dice_loss(pr, torch.from_numpy(tn)); lr.backward()
    net, head = build_net("searched", "G_ALL", 4)
    p = net(dev(xn)); l = loss.WeightedDiceLoss()(p, dev(tn)); l.backward()
    print("B", B, "loss diff", abs(float(l) - float(lr)), "p diff", float((p.detach().cpu() - pr.detach()).abs().max()))
    tot = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in P.values())))
    bad = []
    for n, q in net.named_parameters():
        d = float((q.grad.cpu() - P[n].grad).double().norm()) / tot
        bad.append((d, n))
    bad.sort(reverse=True)
    print("  worst:", bad[:6])
