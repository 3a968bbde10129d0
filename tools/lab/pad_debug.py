import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import numpy as np, torch
from nas_3d_unet_amd import loss, nas, searched, programs as P, fused
from oracle import ref_path as orc
from _util import fill_module, dev

def run(cfgt, analytic=True, phases=True):
    cfg = orc.NetCfg(*cfgt)
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    Pm = orc.make_params(orc.supernet_param_specs(cfg), requires_grad=True)
    fill_module(net); net.kernel.last_conv[0].dropout = None; net = net.cuda()
    rng = np.random.default_rng(17); size = 2 ** (cfg.depth + 1)
    xn = rng.standard_normal((2, cfg.in_channels, size, size, 2 * size)).astype(np.float32)
    tn = (rng.uniform(0, 1, (2, cfg.out_channels, size, size, 2 * size)) < 0.3).astype(np.float32)
    pr = orc.supernet_forward(Pm, torch.from_numpy(xn), cfg); lr = orc.dice_loss(pr, torch.from_numpy(tn)); lr.backward()
    prev = (P.ANALYTIC_CONV_BIAS, fused.NODE_PHASES, fused.NODE_APPLY, P.NODE_FWD_COEFFS)
    P.ANALYTIC_CONV_BIAS = analytic
    fused.NODE_PHASES = fused.NODE_APPLY = P.NODE_FWD_COEFFS = phases
    try:
        p = net(dev(xn)); l = loss.WeightedDiceLoss()(p, dev(tn)); l.backward()
    finally:
        P.ANALYTIC_CONV_BIAS, fused.NODE_PHASES, fused.NODE_APPLY, P.NODE_FWD_COEFFS = prev
    total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in Pm.values() if q.grad is not None)))
    rows = []
    for n, q in net.named_parameters():
        ref = Pm[n].grad if Pm[n].grad is not None else torch.zeros_like(Pm[n])
        d = float((q.grad.cpu() - ref).abs().max())
        rows.append((d / (3e-4 * float(ref.abs().max()) + 2e-5 * total), n, d, float(ref.abs().max())))
    rows.sort(reverse=True)
    print(cfgt, "analytic", analytic, "phases", phases, "loss err %.2e" % abs(float(l) - float(lr)), "total %.3f" % total)
    for r in rows[:6]: print("   x%.2f of tol  %-55s d=%.2e refmax=%.2e" % r)

run((4, 6, 3, 2, 3, True))
run((4, 4, 3, 2, 3, True), analytic=True)
run((4, 4, 3, 2, 3, True), analytic=False, phases=False)
run((4, 8, 3, 2, 3, True), analytic=False, phases=False)
run((4, 2, 3, 2, 2, False))


def vs_f64(cfgt):
    cfg = orc.NetCfg(*cfgt)
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net); net.kernel.last_conv[0].dropout = None; net = net.cuda()
    rng = np.random.default_rng(17); size = 2 ** (cfg.depth + 1)
    xn = rng.standard_normal((2, cfg.in_channels, size, size, 2 * size)).astype(np.float32)
    tn = (rng.uniform(0, 1, (2, cfg.out_channels, size, size, 2 * size)) < 0.3).astype(np.float32)
    res = {}
    for dt in (torch.float32, torch.float64):
        Pm = orc.make_params(orc.supernet_param_specs(cfg), dtype=dt, requires_grad=True)
        pr = orc.supernet_forward(Pm, torch.from_numpy(xn).to(dt), cfg); lr = orc.dice_loss(pr, torch.from_numpy(tn).to(dt)); lr.backward()
        res[dt] = {n: q.grad.double() for n, q in Pm.items() if q.grad is not None}
    p = net(dev(xn)); l = loss.WeightedDiceLoss()(p, dev(tn)); l.backward()
    print("vs fp64 oracle:", cfgt)
    for n in ("kernel.up_cells.2._ops.1._ops.2.conv.weight", "kernel.up_cells.2._ops.1._ops.2.conv.bias", "kernel.up_cells.2._ops.1._ops.2.norm.bias",
              "kernel.up_cells.2._ops.3._ops.2.conv.weight", "kernel.up_cells.1._ops.6._ops.0.conv.weight"):
        mine = dict(net.named_parameters())[n].grad.cpu().double()
        r64, r32 = res[torch.float64][n], res[torch.float32][n]
        print("   %-52s |hip - f64| %.2e   |oracle f32 - f64| %.2e   max|f64| %.2e" % (n, float((mine - r64).abs().max()), float((r32 - r64).abs().max()), float(r64.abs().max())))

vs_f64((4, 6, 3, 2, 3, True))
