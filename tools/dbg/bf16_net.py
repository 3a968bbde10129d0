import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np, torch
from test_gpu_nets import build_net, keep_logits
from _util import dev
from nas_3d_unet_amd import loss, unet
size = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(7)
xn = rng.standard_normal((1, 4, size, size, size)).astype(np.float32)
tn = (rng.uniform(0, 1, (1, 3, size, size, size)) < 0.3).astype(np.float32)
res = {}
for st in ("fp32", "bf16", "bf16"):
    net, head = build_net("searched", "G_CONV", 4)
    unet.set_storage(net, st)
    with keep_logits() as k:
        l, p = net.forward_loss(dev(xn), dev(tn))
        logits = k.logits
    l.backward()
    g = torch.cat([q.grad.flatten() for q in net.parameters()])
    res.setdefault(st, []).append((float(l), logits.clone(), p.detach().clone(), g.clone(), {n: q.grad.clone() for n, q in net.named_parameters()}))
(l0, z0, p0, g0, d0), = res["fp32"]
(l1, z1, p1, g1, d1), (l2, z2, p2, g2, d2) = res["bf16"]
print("loss fp32 %.6f bf16 %.6f" % (l0, l1))
print("logits max err / range: %.3e" % (float((z1 - z0).abs().max()) / float(z0.abs().max())))
print("probs max abs err: %.3e" % float((p1 - p0).abs().max()))
print("grad rel err (whole vector): %.3e" % (float((g1 - g0).double().norm()) / float(g0.double().norm())))
worst = sorted(((float((d1[n] - d0[n]).double().norm()) / float(g0.double().norm()), n) for n in d0), reverse=True)[:5]
print("worst per-tensor / |g|:", worst)
print("bit-reproducible:", l1 == l2, torch.equal(z1, z2), torch.equal(g1, g2))
