"""Pin the CPU oracle against every golden vector produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import golden_common as gc
from oracle import ref_path as orc

torch.set_num_threads(8)
RTOL = 2e-5  # fp32 torch-CPU vs fp32 torch-CPU, different op grouping only


def close(a, b, rtol=RTOL, atol=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(1e-30, np.abs(b).max())
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "max err / max|ref| = %.3e" % err


def T(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(grad)


def test_registry_and_inventory(golden):
    g = golden("small")
    assert list(g["registry/OPS"]) == list(orc.PRIMS.keys())
    assert list(g["registry/DownOps"]) == orc.DOWN_NAMES
    assert list(g["registry/UpOps"]) == orc.UP_NAMES
    assert list(g["registry/NormOps"]) == orc.NORM_NAMES
    for gname in ("G_CONV", "G_ALL"):
        names = list(g["inventory/searched/%s/names" % gname])
        shapes = list(g["inventory/searched/%s/shapes" % gname])
        mine = dict((n, str(tuple(s))) for n, s in orc.searched_param_specs(orc.DEFAULT_CFG, getattr(orc, gname)))
        assert mine == dict(zip(names, shapes))
    names = list(g["inventory/supernet/names"])
    shapes = list(g["inventory/supernet/shapes"])
    mine = dict((n, str(tuple(s))) for n, s in orc.supernet_param_specs(orc.DEFAULT_CFG))
    assert mine == dict(zip(names, shapes))
    assert sum(int(np.prod(eval(s))) for s in shapes) == 6854184


@pytest.mark.parametrize("name,c,shape", gc.prim_cases())
def test_prims(golden, name, c, shape):
    g = golden("prims")
    key = gc.prim_key(name, c)
    P = orc.make_params(orc.prim_param_specs(key + ".", name, c), requires_grad=True)
    x = T(gc.case_input(key, (gc.B, c) + shape), True)
    y = orc.prim_forward(P, key + ".", name, x)
    close(y.detach(), g[key + "/y"])
    (y * T(gc.case_cotangent(key, tuple(y.shape)))).sum().backward()
    close(x.grad, g[key + "/dx"])
    for n, p in P.items():
        close(p.grad, g[key + "/grad/" + n], rtol=1e-4)


@pytest.mark.parametrize("key,kw,cin,shape", gc.convops_cases())
def test_convops(golden, key, kw, cin, shape):
    g = golden("convops")
    k = kw.get("kernel_size", 3)
    order = kw.get("ops_order", "weight_norm_act")
    P = orc.make_params(orc.convops_param_specs(key + ".", cin, kw["out_channels"], k, norm="norm" in order),
                        requires_grad=True)
    x = T(gc.case_input(key, (gc.B, cin) + shape), True)
    y = orc.convops_forward(P, key + ".", x, k, kw.get("stride", 1), 1, order=order)
    close(y.detach(), g[key + "/y"])
    (y * T(gc.case_cotangent(key, tuple(y.shape)))).sum().backward()
    close(x.grad, g[key + "/dx"])
    for n, p in P.items():
        close(p.grad, g[key + "/grad/" + n], rtol=1e-4)


@pytest.mark.parametrize("key,c,stride,transposed,shape", gc.mixed_cases())
def test_mixed(golden, key, c, stride, transposed, shape):
    g = golden("mixed")
    downward = not transposed
    names = orc.NORM_NAMES if stride == 1 else (orc.UP_NAMES if transposed else orc.DOWN_NAMES)
    specs = []
    for k, n in enumerate(names):
        specs += orc.prim_param_specs("%s._ops.%d." % (key, k), n, c)
    P = orc.make_params(specs, requires_grad=True)
    x = T(gc.case_input(key, (gc.B, c) + shape), True)
    a = T(gc.case_alpha(key + ("/a1" if stride == 1 else "/a2"), len(names)), True)
    y = orc.mixed_forward(P, key + ".", x, a, stride, downward)
    close(y.detach(), g[key + "/y"])
    (y * T(gc.case_cotangent(key, tuple(y.shape)))).sum().backward()
    close(x.grad, g[key + "/dx"])
    close(a.grad, g[key + "/dalpha"], rtol=1e-4)
    for n, p in P.items():
        close(p.grad.double().norm(), g[key + "/gnorm/" + n], rtol=1e-4)


def _cell_specs(prefix, c0, c1, cn, downward, gene=None):
    specs = orc.convops_param_specs(prefix + "preprocess0.", c0, cn, 1)
    specs += orc.convops_param_specs(prefix + "preprocess1.", c1, cn, 1)
    if gene is None:
        for ei, (_, _, stride) in enumerate(orc._cell_edges(3, downward)):
            for pi, n in enumerate(orc._edge_prims(stride, downward)):
                specs += orc.prim_param_specs(prefix + "_ops.%d._ops.%d." % (ei, pi), n, cn)
    else:
        for oi, (n, _) in enumerate(gene.down if downward else gene.up):
            specs += orc.prim_param_specs(prefix + "_ops.%d." % oi, n, cn)
    return specs


@pytest.mark.parametrize("key,c0,c1,cn,downward,s0,s1", gc.cell_cases())
def test_cells(golden, key, c0, c1, cn, downward, s0, s1):
    g = golden("cells")
    P = orc.make_params(_cell_specs(key + ".", c0, c1, cn, downward), requires_grad=True)
    x0 = T(gc.case_input(key + "/x0", (gc.B, c0) + s0), True)
    x1 = T(gc.case_input(key + "/x1", (gc.B, c1) + s1), True)
    a1 = T(gc.case_alpha_matrix(key + "/a1", 9, 5), True)
    a2 = T(gc.case_alpha_matrix(key + "/a2", 9, 6 if downward else 4), True)
    y = orc.cell_forward(P, key + ".", x0, x1, a1, a2, 3, downward)
    close(y.detach(), g[key + "/y"])
    (y * T(gc.case_cotangent(key, tuple(y.shape)))).sum().backward()
    close(x0.grad, g[key + "/dx0"], rtol=1e-4)
    close(x1.grad, g[key + "/dx1"], rtol=1e-4)
    close(a1.grad, g[key + "/da1"], rtol=1e-4)
    close(a2.grad, g[key + "/da2"], rtol=1e-4)
    for n, p in P.items():
        close(p.grad.double().norm(), g[key + "/gnorm/" + n], rtol=2e-4)
    for gname in ("G_CONV", "G_ALL"):
        gene = getattr(orc, gname)
        k2 = key + "/" + gname
        P = orc.make_params(_cell_specs(k2 + ".", c0, c1, cn, downward, gene), requires_grad=True)
        x0 = T(gc.case_input(key + "/x0", (gc.B, c0) + s0), True)
        x1 = T(gc.case_input(key + "/x1", (gc.B, c1) + s1), True)
        y = orc.searched_cell_forward(P, k2 + ".", x0, x1, gene.down if downward else gene.up, 3, downward)
        close(y.detach(), g[k2 + "/y"])
        (y * T(gc.case_cotangent(key, tuple(y.shape)))).sum().backward()
        close(x0.grad, g[k2 + "/dx0"], rtol=1e-4)
        close(x1.grad, g[k2 + "/dx1"], rtol=1e-4)
        for n, p in P.items():
            close(p.grad.double().norm(), g[k2 + "/gnorm/" + n], rtol=2e-4)


def _net_forward(kind, gname, depth, P, x, return_logits=True):
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    if kind == "searched":
        return orc.searched_forward(P, x, getattr(orc, gname), cfg, return_logits=return_logits)
    return orc.supernet_forward(P, x, cfg, return_logits=return_logits)


@pytest.mark.parametrize("key,kind,gname,depth,size,batch,adam",
                         [c for c in gc.net_cases() if c[4] <= 32])
def test_nets(golden, key, kind, gname, depth, size, batch, adam):
    g = golden("nets")
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    specs = orc.searched_param_specs(cfg, getattr(orc, gname)) if kind == "searched" else orc.supernet_param_specs(cfg)
    P = orc.make_params(specs, requires_grad=True)
    xn, tn = gc.net_batch(key, batch, size)
    x, t = T(xn), T(tn)
    p, logits = _net_forward(kind, gname, depth, P, x)
    loss = orc.dice_loss(p, t)
    loss.backward()
    c = size // 2
    s = slice(c - 3, c + 3)
    assert abs(float(loss) - float(g[key + "/loss"])) < 2e-6
    close(logits.detach()[:, :, s, s, s], g[key + "/logits_crop"], rtol=1e-4)
    close(p.detach()[:, :, s, s, s], g[key + "/probs_crop"], rtol=1e-4)
    close(logits.detach().double().sum(dim=(2, 3, 4)), g[key + "/logits_sum"], rtol=1e-4)
    # gradient parity is norm-relative (SURVEY hard part 7)
    total = float(g[key + "/gnorm_total"])
    for n, q in P.items():
        ref = float(g[key + "/gnorm/" + n])
        assert abs(float(q.grad.double().norm()) - ref) <= 1e-4 * total + 1e-3 * ref, n
        gk = key + "/grad/" + n
        if gk in g.files:
            d = np.abs(q.grad.numpy().astype(np.float64) - g[gk]).max()
            assert d <= 1e-4 * max(ref, 1e-4 * total), (n, d)
    if kind == "supernet":
        gene = orc.supernet_genotype(P, cfg.n_nodes)
        assert [n for n, _ in gene.down] == list(g[key + "/gene_down"])
        assert [i for _, i in gene.down] == list(g[key + "/gene_down_idx"])
        assert [n for n, _ in gene.up] == list(g[key + "/gene_up"])
        assert [i for _, i in gene.up] == list(g[key + "/gene_up_idx"])
    if adam:
        P2 = orc.make_params(specs, requires_grad=True)
        losses = orc.adam_reference_steps(
            P2, lambda Q: orc.dice_loss(_net_forward(kind, gname, depth, Q, x, False), t), adam)
        np.testing.assert_allclose(losses, g[key + "/adam_losses"], rtol=0, atol=5e-5)
        for n, q in P2.items():
            ref = float(g[key + "/adam%d/pnorm/%s" % (adam, n)])
            assert abs(float(q.detach().double().norm()) - ref) <= 2e-4 * max(ref, 1e-3), n


@pytest.mark.parametrize("key,shape", gc.dice_cases())
def test_dice(golden, key, shape):
    g = golden("small")
    p = T(gc.case_probs(key, shape), True)
    t = T(gc.case_targets(key, shape))
    l = orc.dice_loss(p, t)
    l.backward()
    assert abs(float(l) - float(g[key + "/loss"])) < 1e-6
    close(p.grad, g[key + "/dp"], rtol=1e-5)


@pytest.mark.parametrize("key", gc.geno_cases())
def test_genotype(golden, key):
    g = golden("small")
    a1 = gc.case_alpha_matrix(key + "/a1", 9, 5)
    gd = orc.parse_genotype(a1, gc.case_alpha_matrix(key + "/a2d", 9, 6), 3, True)
    gu = orc.parse_genotype(a1, gc.case_alpha_matrix(key + "/a2u", 9, 4), 3, False)
    assert [n for n, _ in gd] == list(g[key + "/down_names"]) and [i for _, i in gd] == list(g[key + "/down_idx"])
    assert [n for n, _ in gu] == list(g[key + "/up_names"]) and [i for _, i in gu] == list(g[key + "/up_idx"])


@pytest.mark.parametrize("key", ["tie/%s/%d" % (d, i) for d in ("down", "up") for i in range(4)])
def test_oracle_genotype_on_near_tied_scores(golden, key):
    g = golden("geno_ties")
    got = orc.parse_genotype(g[key + "/a1"], g[key + "/a2"], 3, "/down/" in key)
    assert [n for n, _ in got] == list(g[key + "/names"]) and [i for _, i in got] == list(g[key + "/idx"])


def _net2_forward(kind, gname, depth, opt, P, x, key, batch):
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    gate = T(gc.case_drop_gate(key, batch, cfg.n_nodes * cfg.init_n_kernels, opt["drop"])) if opt.get("drop") else None
    if kind == "searched":
        return orc.searched_forward(P, x, getattr(orc, gname), cfg, drop_mask=gate, return_logits=True)
    return orc.supernet_forward(P, x, cfg, drop_mask=gate, return_logits=True, normal_w_share=bool(opt.get("wshare")))


@pytest.mark.parametrize("key,kind,gname,depth,size,batch,opt", [c for c in gc.net2_cases() if c[4] <= 32])
def test_nets2(golden, key, kind, gname, depth, size, batch, opt):
    """train-mode head with a known Dropout3d mask (prim_ops.py:66,72-73) and the shared-alpha supernet (nas.py:109-113)"""
    g = golden("nets2")
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    specs = orc.searched_param_specs(cfg, getattr(orc, gname)) if kind == "searched" else orc.supernet_param_specs(cfg, bool(opt.get("wshare")))
    P = orc.make_params(specs, requires_grad=True)
    xn, tn = gc.net_batch(key, batch, size)
    p, logits = _net2_forward(kind, gname, depth, opt, P, T(xn), key, batch)
    loss = orc.dice_loss(p, T(tn))
    loss.backward()
    c = size // 2
    s = slice(c - 3, c + 3)
    assert abs(float(loss) - float(g[key + "/loss"])) < 2e-6
    close(logits.detach()[:, :, s, s, s], g[key + "/logits_crop"], rtol=1e-4)
    close(p.detach()[:, :, s, s, s], g[key + "/probs_crop"], rtol=1e-4)
    total = float(g[key + "/gnorm_total"])
    for n, q in P.items():
        ref = float(g[key + "/gnorm/" + n])
        assert abs(float(q.grad.double().norm()) - ref) <= 1e-4 * total + 1e-3 * ref, n
        gk = key + "/grad/" + n
        if gk in g.files:
            d = np.abs(q.grad.numpy().astype(np.float64) - g[gk]).max()
            assert d <= 1e-4 * max(ref, 1e-4 * total), (n, d)


def test_search_trajectory(golden):
    """search step (search.py:211-238) on the oracle: alpha gradients after each architecture pass, both losses, alphas
    after the update, against the trajectory recorded from the reference modules + torch.optim.Adam"""
    g = golden("nets2")
    key, depth, size, batch, steps = gc.search_cases()[0]
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    P = orc.make_params(orc.supernet_param_specs(cfg), requires_grad=True)
    x, t, vx, vt = (T(a) for a in gc.search_batches(key, batch, size))
    anames = ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up")
    oa = torch.optim.Adam([P[n] for n in anames])
    ok = torch.optim.Adam([v for n, v in P.items() if n.startswith("kernel.")])
    for step in range(steps):
        oa.zero_grad()
        la = orc.dice_loss(orc.supernet_forward(P, vx, cfg), vt)
        la.backward()
        for n in anames:
            close(P[n].grad, g["%s/step%d/dalpha/%s" % (key, step, n)], rtol=2e-3 if step else 2e-4)
        oa.step()
        ok.zero_grad()
        lw = orc.dice_loss(orc.supernet_forward(P, x, cfg), t)
        lw.backward()
        ok.step()
        np.testing.assert_allclose([float(la), float(lw)], g["%s/step%d/losses" % (key, step)], rtol=0, atol=2e-5)


def test_search_benchmarked_size_first_step(golden):
    """BASELINE configs[2] at the benchmarked size (2 + 2 patches of 4x64^3, depth 4, head Dropout3d(0.1) with explicit masks;
    fixture search64.npz from the reference): the oracle's FIRST search step -- alpha gradients, both losses, the stored
    kernel-weight gradients and every gradient norm of the weight pass.  (One step only: ~40 s of CPU.)"""
    g = golden("search64")
    key, depth, size, batch, steps, p = gc.search_bench_case()
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    P = orc.make_params(orc.supernet_param_specs(cfg), requires_grad=True)
    x, t, vx, vt = (T(a) for a in gc.search_batches(key, batch, size))
    ga, gw = (T(a) for a in gc.search_drop_gates(key, steps, batch, cfg.n_nodes * cfg.init_n_kernels, p)[0])
    anames = ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up")
    oa = torch.optim.Adam([P[n] for n in anames])
    la = orc.dice_loss(orc.supernet_forward(P, vx, cfg, drop_mask=ga), vt)
    la.backward()
    for n in anames:
        close(P[n].grad, g["%s/step0/dalpha/%s" % (key, n)], rtol=2e-4)
    oa.step()
    for n in anames:
        close(P[n].detach(), g["%s/step0/alpha/%s" % (key, n)], rtol=1e-5)
    for q in P.values():
        q.grad = None
    lw = orc.dice_loss(orc.supernet_forward(P, x, cfg, drop_mask=gw), t)
    lw.backward()
    np.testing.assert_allclose([float(la), float(lw)], g["%s/step0/losses" % key], rtol=0, atol=2e-5)
    total = float(g["%s/step0/gnorm_total" % key])
    n_full = 0
    for n, q in P.items():
        if not n.startswith("kernel."):
            continue
        ref = float(g["%s/step0/gnorm/%s" % (key, n)])
        assert abs(float(q.grad.double().norm()) - ref) <= 1e-3 * ref + 1e-4 * total, n
        gk = "%s/step0/grad/%s" % (key, n)
        if gk in g.files:
            n_full += 1
            assert np.abs(q.grad.numpy() - g[gk]).max() <= 1e-3 * np.abs(g[gk]).max() + 1e-5 * total, n
    assert n_full >= 8
