"""Which parameters differ between the single-GPU trainer and the bucketed DP schedule (1-rank group)?"""
import os, sys, socket
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np, torch, torch.distributed as dist
from test_gpu_nets import build_net
from _util import dev
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1)
from nas_3d_unet_amd.train import Trainer
rng = np.random.default_rng(41)
x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
nsteps = int(os.environ.get("STEPS", "1"))
graph = os.environ.get("GRAPH", "1") == "1"
net, _ = build_net("searched", "G_CONV", 4)
ref = Trainer(net, graph=graph)
lr_ = [float(ref.step(x, t)) for _ in range(nsteps)]
wr = ref.fp.flat.clone(); gr = ref.fp.grad.clone() if hasattr(ref.fp, "grad") else None
net3, _ = build_net("searched", "G_CONV", 4)
plain = Trainer(net3, graph=graph, side_wgrad=False)
lp = [float(plain.step(x, t)) for _ in range(nsteps)]
wp = plain.fp.flat.clone()
net4, _ = build_net("searched", "G_CONV", 4)
eager = Trainer(net4, graph=False)
le = [float(eager.step(x, t)) for _ in range(nsteps)]
we = eager.fp.flat.clone()
os.environ["N3D_FORCE_DP"] = "1"
net2, _ = build_net("searched", "G_CONV", 4)
tr = Trainer(net2, graph=graph, n_buckets=int(os.environ.get("BUCKETS", "2")), comm="torch")
l = [float(tr.step(x, t)) for _ in range(nsteps)]
w = tr.fp.flat
print("losses", lr_, l)
names = [n for n, _ in net2.named_parameters()]
offs = tr.fp.offsets
ps = list(net2.parameters())
for i, (n, p) in enumerate(zip(names, ps)):
    a = wp[offs[i]:offs[i] + p.numel()]; b = wr[offs[i]:offs[i] + p.numel()]      # plain graph vs side graph
    d = float((a - b).abs().max())
    if d > float(os.environ.get("THR", "5e-6")): print("%-50s %s maxdiff %.3e" % (n, tuple(p.shape), d))
for nm, a, b in (("dp vs side", w, wr), ("dp vs plain", w, wp), ("plain vs side", wp, wr), ("plain vs eager", wp, we), ("side vs eager", wr, we)):
    dd = (a - b).abs()
    print("%s: max %.3e norm ratio %.3e frac>5e-6 %.4f" % (nm, float(dd.max()), float(dd.double().norm()) / float(b.double().norm()), float((dd > 5e-6).float().mean())))
d = (w - wr).abs()
print("weights: max %.3e mean %.3e frac>5e-6 %.4f" % (float(d.max()), float(d.mean()), float((d > 5e-6).float().mean())))
m, mr = tr.fp.exp_avg, ref.fp.exp_avg
print("exp_avg: max|ref| %.3e maxdiff %.3e" % (float(mr.abs().max()), float((m - mr).abs().max())))
worst = 0.0
for i, (n, p) in enumerate(zip(names, ps)):
    a = m[offs[i]:offs[i] + p.numel()]; b = mr[offs[i]:offs[i] + p.numel()]
    rel = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
    if rel > 1e-3: print("   exp_avg %-45s %s rel %.3e" % (n, tuple(p.shape), rel))
    worst = max(worst, rel)
print("exp_avg: worst per-parameter maxdiff / max|ref| %.3e" % worst)
dist.destroy_process_group()
