"""GPU: op-level path used when the reference's unchanged searched.py drives this repo's registry ops --
torch's own `0 + a + b` and `torch.cat(dim=1)` between our autograd nodes (searched.py:45-51), channels-last
strides preserved -- must equal the fused SearchedCell path and the oracle."""
import numpy as np
import pytest
import torch

from _util import assert_close, fill_module
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


def test_unfused_cell_algebra_equals_fused():
    from nas_3d_unet_amd import searched
    gene = searched.Genotype(*orc.G_ALL)
    cell = fill_module(searched.SearchedCell(3, 12, 12, 8, gene, True), "c.").cuda()
    rng = np.random.default_rng(3)
    x0n = rng.standard_normal((2, 12, 8, 8, 8)).astype(np.float32)
    x1n = rng.standard_normal((2, 12, 4, 4, 4)).astype(np.float32)
    x0, x1 = torch.from_numpy(x0n).cuda().requires_grad_(True), torch.from_numpy(x1n).cuda().requires_grad_(True)
    y = cell(x0, x1)
    y.square().sum().backward()
    g_fused = {n: p.grad.clone() for n, p in cell.named_parameters()}
    dx0 = x0.grad.clone()
    for p in cell.parameters():
        p.grad = None
    # the reference's own forward body, verbatim semantics, on our ops
    a0, a1 = torch.from_numpy(x0n).cuda().requires_grad_(True), torch.from_numpy(x1n).cuda().requires_grad_(True)
    xs = [cell.preprocess0(a0), cell.preprocess1(a1)]
    i = 0
    for node in range(3):
        outs = []
        for _ in range(2):
            outs.append(cell._ops[i](xs[cell.genolist[i][1]]))
            i += 1
        xs.append(sum(outs))
    y2 = torch.cat(xs[-3:], dim=1)
    assert_close(y2, y, 1e-6, "forward")
    y2.square().sum().backward()
    assert_close(a0.grad, dx0, 2e-5, "dx0")
    for n, p in cell.named_parameters():
        assert_close(p.grad, g_fused[n], 2e-4, n)


@pytest.mark.parametrize("kind,cfg", [("searched", (4, 6, 3, 2, 3, True)), ("searched", (3, 5, 2, 2, 2, False)), ("supernet", (4, 6, 3, 2, 2, True)),
                                      ("supernet", (4, 2, 3, 2, 2, False))])
def test_reference_forward_bodies_on_odd_channel_counts(kind, cfg):
    """round 6 (missing 3): the reference builds any `init_n_kernels` (nas.py:13-49, searched.py:55-90).  When ITS unchanged
    nas.py / searched.py drive this repo's prim_ops / cell, every op and cell sees channel counts that are not multiples of 4 on its own:
    each op then runs through its zero-padded twin (prim_ops._OpTwin), MixedOp / Cell / SearchedCell fall back to the reference's own
    algebra over those ops.  Here the reference's forward bodies (KernelNet.forward nas.py:54-79, SearchedNet.forward searched.py:95-111
    = unet.route) run op by op on such a net: loss, probabilities and every gradient against the oracle in fp64."""
    import torch.nn.functional as F
    from nas_3d_unet_amd import loss, nas, searched, unet
    from test_gpu_nets import _genotype_for
    cfg = orc.NetCfg(*cfg)
    assert unet.needs_padding(cfg.init_n_kernels, cfg.depth, cfg.n_nodes, cfg.channel_change)
    odt = torch.float64
    rng = np.random.default_rng(18)
    size = 2 ** (cfg.depth + 1)
    xn = rng.standard_normal((2, cfg.in_channels, size, size, 2 * size)).astype(np.float32)
    tn = (rng.uniform(0, 1, (2, cfg.out_channels, size, size, 2 * size)) < 0.3).astype(np.float32)
    x = torch.from_numpy(xn).cuda()
    if kind == "searched":
        gene = _genotype_for(cfg.n_nodes)
        net = searched.SearchedNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, cfg.channel_change,
                                   searched.Genotype(list(gene.down), list(gene.up)))
        P = orc.make_params(orc.searched_param_specs(cfg, gene), dtype=odt, requires_grad=True)
        pr = orc.searched_forward(P, torch.from_numpy(xn).to(odt), gene, cfg)
        fill_module(net)
        net.last_conv[0].dropout = None
        net = net.cuda()
        plain = lambda cell, skip, cur: cell(skip, cur)
        p = unet.route(net, x, plain, plain)          # the reference's SearchedNet.forward body over the per-op modules
    else:
        net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
        P = orc.make_params(orc.supernet_param_specs(cfg), dtype=odt, requires_grad=True)
        pr = orc.supernet_forward(P, torch.from_numpy(xn).to(odt), cfg)
        fill_module(net)
        net.kernel.last_conv[0].dropout = None
        net = net.cuda()
        a1d, a1u, a2d, a2u = (F.softmax(a, dim=-1) for a in (net.alpha1_down, net.alpha1_up, net.alpha2_down, net.alpha2_up))
        p = unet.route(net.kernel, x, lambda cell, skip, cur: cell(skip, cur, a1d, a2d), lambda cell, skip, cur: cell(skip, cur, a1u, a2u))
    lr = orc.dice_loss(pr, torch.from_numpy(tn).to(odt))
    lr.backward()
    l = loss.WeightedDiceLoss()(p, torch.from_numpy(tn).cuda())
    l.backward()
    assert abs(float(l.detach()) - float(lr.detach())) < 5e-6
    assert float((p.detach().cpu().double() - pr.detach().double()).abs().max()) < 3e-5
    total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in P.values() if q.grad is not None)))
    for n, q in net.named_parameters():
        ref = P[n].grad if P[n].grad is not None else torch.zeros_like(P[n])
        assert q.grad is not None, n
        d = float((q.grad.cpu().double() - ref.double()).abs().max())
        assert d <= 3e-4 * float(ref.abs().max()) + 2e-5 * total, (n, d)
    # a twin never shows in the module tree: the state dict stays the reference's
    assert not any("twin" in k for k in net.state_dict())


def test_op_twin_follows_the_op_in_eval_mode_and_without_autograd():
    """prim_ops._OpTwin: inference of an odd-channel op under no_grad keeps no autograd state, eval / train and a dropout switched
    off after construction follow the op, and the twin never appears in the state dict"""
    from nas_3d_unet_amd import prim_ops as po
    torch.manual_seed(3)
    op = po.ConvOps(6, 6, kernel_size=3, dropout_rate=0.5).cuda()
    x = torch.randn(2, 6, 8, 8, 8, device="cuda")
    op.eval()
    with torch.no_grad():
        y0 = op(x)
    assert not y0.requires_grad and y0.shape == (2, 6, 8, 8, 8)
    tw = op.__dict__["_n3d_optwin"]
    assert not tw.twin.training and tw.twin.dropout is not None
    # eval-mode forward equals the oracle's functional conv + GroupNorm + ReLU on the same parameters
    ref = torch.nn.functional.conv3d(x.double().cpu(), op.conv.weight.detach().double().cpu(), op.conv.bias.detach().double().cpu(), padding=1)
    ref = torch.relu(torch.nn.functional.group_norm(ref, 1, op.norm.weight.detach().double().cpu(), op.norm.bias.detach().double().cpu(), 1e-5))
    assert_close(y0, ref.float(), 2e-5, "eval forward")
    op.dropout = None
    op.train()
    y1 = op(x)
    assert op.__dict__["_n3d_optwin"].twin.dropout is None and op.__dict__["_n3d_optwin"].twin.training
    assert_close(y1, ref.float(), 2e-5, "train forward without dropout")
    assert sorted(op.state_dict()) == ["conv.bias", "conv.weight", "norm.bias", "norm.weight"]
