"""GPU: the node-level launches of the supernet backward / forward through their Python wrappers (kernels.node_bwd_prologue,
node_bwd_apply_sum, node_fwd_coeffs -> n3d_affine_act_bwd_reduceN with 16 terms, n3d_node_bwd_coeffs, n3d_affine_act_bwd_apply_sum,
n3d_node_fwd_coeffs).  The reference has no counterpart (cell.py:29-32 sums `w * op(x)` term by term through autograd); what is
checked is the contract in include/n3d.h: the merged launches do the per-term work of the separate ones, bit for bit, and the
sum-by-target apply equals the sequence of single apply launches and a torch restatement of it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _view(B, Cc, S, rng):
    from nas_3d_unet_amd import kernels as K
    t = K.empty_ndhwc(B, Cc, S, S, S, torch.device("cuda"))
    t.copy_(torch.from_numpy(rng.standard_normal((B, Cc, S, S, S)).astype(np.float32)).cuda())
    return K.as_view(t)


@pytest.mark.parametrize("B,Cc,S", [(2, 16, 8), (3, 8, 16), (2, 64, 4)])
def test_apply_sum_equals_the_single_launches_and_torch(B, Cc, S):
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd._lib import ACCUMULATE, RELU
    rng = np.random.default_rng(B * 100 + Cc)
    dev_ = torch.device("cuda")
    dout = _view(B, Cc, S, rng)
    coef = lambda: torch.from_numpy(rng.standard_normal((B, Cc)).astype(np.float32)).to(dev_)
    # two targets: the first gets an SE-type term (no mask, no C) then an identity-type term (mask, all three coefficients) on top of
    # its previous content, the second a single masked term written fresh
    raws = [_view(B, Cc, S, rng) for _ in range(3)]
    terms = [(raws[0], None, None, False, coef(), coef(), None), (raws[1], coef(), coef(), True, coef(), coef(), coef()),
             (raws[2], coef(), coef(), True, coef(), coef(), coef())]
    prev = _view(B, Cc, S, rng)
    tgt_a, tgt_b = K.like(prev), K.like(prev)
    tgt_a.t.copy_(prev.t)
    ref_a, ref_b = K.like(prev), K.like(prev)
    ref_a.t.copy_(prev.t)
    for (raw, a, b, relu, cA, cB, cC), (target, acc) in zip(terms, ((ref_a, True), (ref_a, True), (ref_b, False))):
        K.affine_act_bwd_apply(dout, raw, a, b, cA, cB, cC, target, (RELU if relu else 0) | (ACCUMULATE if acc else 0))
    items = [terms[0] + (tgt_a, True), terms[1] + (tgt_a, True), terms[2] + (tgt_b, False)]
    K.node_bwd_apply_sum(dout, items)
    torch.cuda.synchronize()
    assert torch.equal(tgt_a.t, ref_a.t) and torch.equal(tgt_b.t, ref_b.t)
    # torch restatement (fp64): x = cA * g + cB + cC * raw, g = dout where a * raw + b > 0
    def term(raw, a, b, relu, cA, cB, cC):
        bc = lambda c: c.double()[:, :, None, None, None]
        g = dout.t.double()
        if relu and a is not None:
            g = torch.where(bc(a) * raw.t.double() + bc(b) > 0, g, torch.zeros_like(g))
        x = (bc(cA) if cA is not None else 1.0) * g + (bc(cB) if cB is not None else 0.0)
        return x + (bc(cC) * raw.t.double() if cC is not None else 0.0)
    want_a = prev.t.double() + term(*terms[0]) + term(*terms[1])
    want_b = term(*terms[2])
    assert float((tgt_a.t.double() - want_a).abs().max()) < 2e-5 and float((tgt_b.t.double() - want_b).abs().max()) < 2e-5


def test_apply_sum_rejects_what_it_cannot_order():
    from nas_3d_unet_amd import kernels as K
    rng = np.random.default_rng(5)
    dout = _view(2, 8, 8, rng)
    r = _view(2, 8, 8, rng)
    c = torch.ones((2, 8), device="cuda")
    ta, tb = K.like(r), K.like(r)
    with pytest.raises(K.N3DError):      # a target's terms must be consecutive
        K.node_bwd_apply_sum(dout, [(r, None, None, False, c, c, None, ta, False), (r, None, None, False, c, c, None, tb, False),
                                    (r, None, None, False, c, c, None, ta, True)])
    with pytest.raises(K.N3DError):      # at most four terms per target
        K.node_bwd_apply_sum(dout, [(r, None, None, False, c, c, None, ta, False)] * 5)


@pytest.mark.parametrize("B", [2, 3])
def test_node_coefficient_launches_equal_the_separate_ones(B):
    """forward: node_fwd_coeffs == gn_coeffsN + se_gate_fwdN; backward: node_bwd_prologue == reduceN + gn_bwd_coeffsN + se_gate_bwdN"""
    from nas_3d_unet_amd import kernels as K
    rng = np.random.default_rng(17 + B)
    dev_ = torch.device("cuda")
    Cc, S, G = 16, 8, 1
    N = S ** 3
    lin = lambda i, o: torch.nn.Linear(i, o).to(dev_)
    raws = [_view(B, Cc, S, rng) for _ in range(5)]
    gam = [torch.nn.Parameter(torch.from_numpy(rng.standard_normal(Cc).astype(np.float32)).to(dev_)) for _ in range(3)]
    bet = [torch.nn.Parameter(torch.from_numpy(rng.standard_normal(Cc).astype(np.float32)).to(dev_)) for _ in range(3)]
    fcs = [torch.nn.Sequential(lin(Cc, 1), torch.nn.ReLU(), lin(1, Cc)) for _ in range(2)]
    stats = [K.channel_stats(r) for r in raws]
    gn_terms = [(raws[i], stats[i][0], stats[i][1], gam[i], bet[i]) for i in range(3)]
    se_terms = [(stats[3 + i][0], stats[3 + i][1], fcs[i]) for i in range(2)]
    # ---- forward
    ga, sa = K.node_fwd_coeffs(gn_terms, G, 1e-5, se_terms)
    gb = K.gn_coeffsN(gn_terms, G, 1e-5)
    sb = K.se_gate_fwdN(se_terms, N, B, Cc)
    torch.cuda.synchronize()
    for x, y in zip(ga + sa, gb + sb):
        for u, v in zip(x, y):
            assert torch.equal(u, v)
    # ---- backward
    dout = _view(B, Cc, S, rng)
    wts = [torch.from_numpy(rng.uniform(0.1, 1.0, 1).astype(np.float32)).to(dev_) for _ in range(5)]
    da = torch.zeros(8, device=dev_)

    def gn_dicts():
        return [dict(raw=raws[i], a=gb[i][0], b=gb[i][1], mr=gb[i][2], sumraw=None, gamma=gam[i], beta=bet[i], wptr=wts[i].data_ptr(), relu=True,
                     conv_bias=None, draw=K.like(raws[i]), dalpha_ptr=da.data_ptr() + 4 * i) for i in range(3)]

    def gates(sums_of):
        return [dict(sums=sums_of(i)[0], rows=sums_of(i)[1], wptr=wts[3 + i].data_ptr(), mean=sb[i][0], hidden=sb[i][1], gate=sb[i][2], fc=fcs[i],
                     dalpha_ptr=da.data_ptr() + 4 * (3 + i)) for i in range(2)]

    singles = [(raws[3 + i], sb[i][2], None, False) for i in range(2)]
    # separate launches
    for p in gam + bet + [q for fc in fcs for q in fc.parameters()]:
        p._n3d_grad = None
    g1 = K.GnGroupBwd(dout, gn_dicts(), G)
    g1.reduce(); g1.coeffs(); g1.apply()
    red1 = K.affine_act_bwd_reduceN(dout, singles)
    se1 = K.se_gate_bwdN(gates(lambda i: red1[i]), N, B, Cc)
    torch.cuda.synchronize()
    da1 = da.clone()
    da.zero_()
    # merged launches
    g2 = K.GnGroupBwd(dout, gn_dicts(), G)
    red2, se2, _ = K.node_bwd_prologue(dout, [g2], singles, [(i, {k: v for k, v in gates(lambda i: (None, 0))[i].items() if k not in ("sums", "rows")})
                                                            for i in range(2)])
    g2.apply()
    torch.cuda.synchronize()
    assert torch.equal(da, da1)
    assert torch.equal(g1.sums, g2.sums) and torch.equal(g1.coef, g2.coef)
    for (a1, r1), (a2, r2) in zip(red1, red2):
        assert r1 == r2 and torch.equal(a1, a2)
    for x, y in zip(se1, se2):
        for u, v in zip(x, y):
            assert torch.equal(u, v)
    for t1, t2 in zip(g1.terms, g2.terms):
        assert torch.equal(t1["draw"].t, t2["draw"].t)
    for (dg1, db1, _), (dg2, db2, _) in zip(g1.outs, g2.outs):
        assert torch.equal(dg1, dg2) and torch.equal(db1, db2)
