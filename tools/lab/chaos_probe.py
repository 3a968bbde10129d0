"""How fast does a rounding-sized perturbation of the weights grow over Adam steps?  Two identical single-stream trainers; after the
first step a few weights of one are nudged by PERT (default 1e-7); the weight difference after each further step is printed."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np, torch
from test_gpu_nets import build_net
from _util import dev
from nas_3d_unet_amd.train import Trainer
rng = np.random.default_rng(41)
x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
pert = float(os.environ.get("PERT", "1e-7"))
trs = []
for k in range(2):
    net, _ = build_net("searched", "G_CONV", 4)
    trs.append(Trainer(net, graph=True, side_wgrad=False))
for tr in trs: tr.step(x, t)
torch.cuda.synchronize()
assert torch.equal(trs[0].fp.flat, trs[1].fp.flat)
g = torch.Generator(device="cuda").manual_seed(1)
mask = (torch.rand(trs[1].fp.flat.shape, device="cuda", generator=g) < float(os.environ.get("FRAC", "0.01"))).float()
trs[1].fp.flat.add_(mask * pert)
for s in range(2, 6):
    for tr in trs: tr.step(x, t)
    torch.cuda.synchronize()
    d = (trs[0].fp.flat - trs[1].fp.flat).abs()
    print("step %d: max %.3e norm ratio %.3e frac>5e-6 %.4f" % (s, float(d.max()), float(d.double().norm()) / float(trs[0].fp.flat.double().norm()), float((d > 5e-6).float().mean())))
