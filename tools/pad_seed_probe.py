"""round 6 probe: gradient error of a padded-twin net against the fp64 oracle at 32^3 under single switches (bisects the kernel form)"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
from oracle import ref_path as orc
import test_gpu_nets as T
from nas_3d_unet_amd import loss, searched, unet, fused, programs as P_, kernels as K, _lib
from _util import dev, fill_module
cfgt, shape = (4, 2, 3, 2, 3, True), (32, 32, 32)
cfg = orc.NetCfg(*cfgt)
gene = T._genotype_for(cfg.n_nodes)
P = orc.make_params(orc.searched_param_specs(cfg, gene), dtype=torch.float64, requires_grad=True)
rng = np.random.default_rng(17)
xn = rng.standard_normal((2, cfg.in_channels) + shape).astype(np.float32)
tn = (rng.uniform(0, 1, (2, cfg.out_channels) + shape) < 0.3).astype(np.float32)
pr = orc.searched_forward(P, torch.from_numpy(xn).double(), gene, cfg); lr = orc.dice_loss(pr, torch.from_numpy(tn).double()); lr.backward()
total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in P.values() if q.grad is not None)))

def run(tag):
    net = searched.SearchedNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, cfg.channel_change, searched.Genotype(list(gene.down), list(gene.up)))
    fill_module(net); net.last_conv[0].dropout = None; net.last_conv[0]._segments = None; net = net.cuda()
    p = net(dev(xn)); l = loss.WeightedDiceLoss()(p, dev(tn)); l.backward()
    res = []
    for n, q in net.named_parameters():
        ref = P[n].grad
        d = float((q.grad.cpu().double() - ref.double()).abs().max())
        res.append((d / (3e-4 * float(ref.abs().max()) + 2e-5 * total), n, d))
    res.sort(reverse=True)
    print("%-28s worst ratio %.3f %s (%.2e); #>0.01: %d" % (tag, res[0][0], res[0][1], res[0][2], sum(r[0] > 0.01 for r in res)), flush=True)
    return res

import contextlib
from nas_3d_unet_amd.train import _padded_flags
def setup(cfgt_, shape_, B=2, seed=17):
    global cfg, gene, P, xn, tn, total
    cfg = orc.NetCfg(*cfgt_); gene = T._genotype_for(cfg.n_nodes)
    P = orc.make_params(orc.searched_param_specs(cfg, gene), dtype=torch.float64, requires_grad=True)
    rng = np.random.default_rng(seed)
    xn = rng.standard_normal((B, cfg.in_channels) + shape_).astype(np.float32)
    tn = (rng.uniform(0, 1, (B, cfg.out_channels) + shape_) < 0.3).astype(np.float32)
    pr = orc.searched_forward(P, torch.from_numpy(xn).double(), gene, cfg); lr = orc.dice_loss(pr, torch.from_numpy(tn).double()); lr.backward()
    total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in P.values() if q.grad is not None)))
for seed in (17, 18, 19, 20):
    setup((4, 2, 3, 2, 3, True), (32, 32, 32), seed=seed); run("padded 2 32^3 seed %d" % seed)
for seed in (18, 19):
    setup((4, 2, 3, 2, 3, True), (16, 32, 64), seed=seed); run("padded 2 16x32x64 seed %d" % seed)
    setup((4, 4, 3, 2, 3, True), (32, 32, 32), seed=seed); run("unpadded 4 32^3 seed %d" % seed)
