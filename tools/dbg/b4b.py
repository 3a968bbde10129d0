import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_nets import build_net
from _util import dev
from nas_3d_unet_amd.train import Trainer
rng = np.random.default_rng(31)
xn = rng.standard_normal((4, 4, 32, 32, 32)).astype(np.float32)
tn = (rng.uniform(0, 1, (4, 3, 32, 32, 32)) < 0.3).astype(np.float32)
def fresh():
    net, _ = build_net("searched", "G_ALL", 4)
    return Trainer(net, graph=False)
def names(tr):
    return [n for n, _ in tr.model.named_parameters()]
def report(tag, a, b, tr):
    tot = float(a.double().norm())
    print(tag, "rel", float((a - b).double().norm()) / tot)
    bad = []
    for (n, p), o in zip(tr.model.named_parameters(), tr.fp.offsets):
        k = p.numel()
        bad.append((float((a[o:o + k] - b[o:o + k]).double().norm()) / tot, n))
    bad.sort(reverse=True)
    print("   ", bad[:5])
A = fresh(); A._fwd_bwd(dev(xn), dev(tn)); g4 = A.fp.grad.clone()
B = fresh(); B._fwd_bwd(dev(xn[:2]), dev(tn[:2])); ga = B.fp.grad.clone()
B._fwd_bwd(dev(xn[2:]), dev(tn[2:])); gb = B.fp.grad.clone()
C = fresh(); C._fwd_bwd(dev(xn[2:]), dev(tn[2:])); gc = C.fp.grad.clone()
C._fwd_bwd(dev(xn[2:]), dev(tn[2:])); gc2 = C.fp.grad.clone()
report("same input, second call vs fresh", gc, gb, B)
report("same input, same trainer twice", gc, gc2, B)
report("B4 vs mean of fresh halves", g4, 0.5 * (ga + gc), B)
A._fwd_bwd(dev(xn[:2]), dev(tn[:2])); ga2 = A.fp.grad.clone()
report("B2 after B4 on one trainer vs fresh", ga, ga2, B)
