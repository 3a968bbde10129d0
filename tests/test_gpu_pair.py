"""GPU parity of the node-level pair kernels (n3d_affine_act_gn2 / _bwd_reduce2 / _bwd_apply_gn2) against the two
single-op launches they replace, and against the torch-CPU formula of the reference node
  relu(GN_a(raw_a)) + [relu](GN_b(raw_b))   (searched.py:45-50 with prim_ops.py:56-63,75-80)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import assert_close

pytestmark = pytest.mark.gpu


def group_count(c):
    return 1 if c % 16 else c // 16


@pytest.mark.parametrize("C,shape,B,relu_b", [(4, (8, 8, 16), 2, True), (8, (4, 8, 8), 2, False), (16, (4, 4, 8), 3, True),
                                               (32, (4, 4, 4), 2, True), (64, (2, 2, 2), 2, True),
                                               (4, (32, 32, 32), 2, True), (8, (16, 32, 32), 5, False)])  # last two: > 64 partial rows
def test_pair_forward_backward(C, shape, B, relu_b):
    from nas_3d_unet_amd import kernels as K
    dev = torch.device("cuda")
    rng = np.random.default_rng(C)
    G = group_count(C)
    raws = [rng.standard_normal((B, C) + shape).astype(np.float32) * 1.5 + 0.3 for _ in range(2)]
    gam = [rng.standard_normal(C).astype(np.float32) * 0.5 + 1.0 for _ in range(2)]
    bet = [rng.standard_normal(C).astype(np.float32) * 0.2 for _ in range(2)]
    dnode = rng.standard_normal((B, C) + shape).astype(np.float32)
    relus = [True, relu_b]
    # ---- torch CPU reference
    rc = [torch.from_numpy(r).requires_grad_(True) for r in raws]
    gc = [torch.from_numpy(g).requires_grad_(True) for g in gam]
    bc = [torch.from_numpy(b).requires_grad_(True) for b in bet]
    terms = []
    for k in range(2):
        z = F.group_norm(rc[k], G, gc[k], bc[k], 1e-5)
        terms.append(F.relu(z) if relus[k] else z)
    yc = terms[0] + terms[1]
    (yc * torch.from_numpy(dnode)).sum().backward()
    # ---- HIP pair kernels
    rv = [K.as_view(torch.from_numpy(r).to(dev)) for r in raws]
    gp = [torch.nn.Parameter(torch.from_numpy(g).to(dev)) for g in gam]
    bp = [torch.nn.Parameter(torch.from_numpy(b).to(dev)) for b in bet]
    stats = [K.channel_stats(v) for v in rv]
    assert K.pair_shape_ok(C)
    out = K.as_view(K.empty_ndhwc(B, C, *shape, dev))
    sv = K.affine_act_gn2([(rv[k], stats[k][0], stats[k][1], gp[k], bp[k], None, relus[k]) for k in range(2)], G, 1e-5, out, 0)
    assert_close(out.t, yc, 2e-5, "node output")
    dv = K.as_view(torch.from_numpy(dnode).to(dev))
    draws = [K.as_view(K.empty_ndhwc(B, C, *shape, dev)) for _ in range(2)]
    tl = [dict(raw=rv[k], a=sv[k][0], b=sv[k][1], mr=sv[k][2], sumraw=sv[k][3], gamma=gp[k], beta=bp[k], wptr=None, relu=relus[k],
               conv_bias=None, draw=draws[k]) for k in range(2)]
    outs = K.affine_act_bwd_gn2(dv, tl, G)
    for k in range(2):
        assert_close(draws[k].t, rc[k].grad, 1e-4, "d raw %d" % k)
        assert_close(outs[k][0], gc[k].grad, 1e-4, "dgamma %d" % k)
        assert_close(outs[k][1], bc[k].grad, 1e-4, "dbeta %d" % k)
    # ---- and against the single-op launches (same arithmetic, different launch grouping)
    out1 = K.as_view(K.empty_ndhwc(B, C, *shape, dev))
    for k in range(2):
        fl = (K.RELU if relus[k] else 0) | (K.ACCUMULATE if k else 0)
        if stats[k][1] <= K.fused_max_rows():
            K.affine_act_gn(rv[k], stats[k][0], stats[k][1], gp[k], bp[k], G, 1e-5, None, out1, fl)
        else:
            a, b, _, _ = K.gn_coeffs(stats[k][0], stats[k][1], gp[k], bp[k], B, C, G, rv[k].N)
            K.affine_act(rv[k], a, b, None, out1, fl)
    assert torch.equal(out1.t, out.t), "pair forward differs from the two single launches"


def test_pair_rejects_unsupported_shape():
    from nas_3d_unet_amd import kernels as K
    assert not K.pair_ok(12, 1, 4, 4, 2)     # C not a power of two
    assert not K.pair_ok(16, 1, 1000, 4, 2)  # too many partial rows for the fused prologue
    assert K.pair_ok(64, 4, 1, 1, 2)


@pytest.mark.parametrize("C,shape,B,nterms,two", [
    (16, (8, 8, 8), 2, 2, False),    # mode 2: one workgroup per (group, sample), 2048 quads each (the 8^3 level of the benchmarked net)
    (32, (8, 8, 8), 2, 2, False),    # ... two groups
    (32, (8, 8, 8), 2, 2, True),     # ... the two preprocess ops of a cell (separate output gradients)
    (16, (8, 8, 8), 3, 2, False),    # ... three samples (a ticket counter wrapping at 2)
    (8, (8, 8, 16), 2, 1, False),    # mode 2, a single epilogue
    (32, (4, 4, 4), 2, 1, False),    # mode 1 (all samples in one workgroup), a single epilogue
    (64, (2, 2, 2), 2, 1, False),    # ... samples padded to whole waves
    (16, (8, 8, 8), 2, 1, False),
])
def test_one_launch_epilogue_backward(C, shape, B, nterms, two):
    """n3d_affine_act_bwd_small (reduction, coefficients, parameter gradients incl. the conv-bias gradient and d(raw) in one
    launch) against torch on the CPU and against the reduce2 + apply_gn2 pair; launched repeatedly (the ticket words reset themselves)"""
    from nas_3d_unet_amd import kernels as K
    dev = torch.device("cuda")
    rng = np.random.default_rng(C + B)
    G = group_count(C)
    N = shape[0] * shape[1] * shape[2]
    mode = int(K._lib.load().n3d_bwd_small_mode(B, N, C, G))
    K._small_modes[(B, N, C, G)] = mode      # (the trainers leave mode 2 off by default: no faster than the two launches; this tests the kernel)
    assert mode == (1 if B * ((N * (C // G // 4) + 63) // 64 * 64) <= 2048 else 2)
    raws = [rng.standard_normal((B, C) + shape).astype(np.float32) * 1.5 + 0.3 for _ in range(nterms)]
    gam = [rng.standard_normal(C).astype(np.float32) * 0.5 + 1.0 for _ in range(nterms)]
    bet = [rng.standard_normal(C).astype(np.float32) * 0.2 for _ in range(nterms)]
    dn = [rng.standard_normal((B, C) + shape).astype(np.float32) for _ in range(2 if two else 1)]
    relus = [True, False][:nterms]
    rc = [torch.from_numpy(r).requires_grad_(True) for r in raws]
    gc = [torch.from_numpy(g).requires_grad_(True) for g in gam]
    bc = [torch.from_numpy(b).requires_grad_(True) for b in bet]
    loss = 0
    for k in range(nterms):
        z = F.group_norm(rc[k], G, gc[k], bc[k], 1e-5)
        z = F.relu(z) if relus[k] else z
        loss = loss + (z * torch.from_numpy(dn[k if two else 0])).sum()
    loss.backward()
    rv = [K.as_view(torch.from_numpy(r).to(dev)) for r in raws]
    gp = [torch.nn.Parameter(torch.from_numpy(g).to(dev)) for g in gam]
    bp = [torch.nn.Parameter(torch.from_numpy(b).to(dev)) for b in bet]
    cb = [torch.nn.Parameter(torch.zeros(C, device=dev)) for _ in range(nterms)]    # a conv bias in front of the norm: gradient = sum d(raw)
    sv = []
    for k in range(nterms):
        st, rows = K.channel_stats(rv[k])
        sv.append(K.gn_coeffs(st, rows, gp[k], bp[k], B, C, G, N, 1e-5))
    dv = [K.as_view(torch.from_numpy(d).to(dev)) for d in dn]

    def terms():
        return [dict(raw=rv[k], a=sv[k][0], b=sv[k][1], mr=sv[k][2], sumraw=sv[k][3], gamma=gp[k], beta=bp[k], wptr=None, relu=relus[k],
                     conv_bias=cb[k], draw=K.as_view(K.empty_ndhwc(B, C, *shape, dev))) for k in range(nterms)]
    results = []
    for rep in range(3):
        tl = terms()
        outs = K.affine_act_bwd_small(dv[0], tl, G, dv[1] if two else None)
        results.append((tl, outs))
    torch.cuda.synchronize()
    tl, outs = results[0]
    for k in range(nterms):
        assert_close(tl[k]["draw"].t, rc[k].grad, 1e-4, "d raw %d" % k)
        assert_close(outs[k][0], gc[k].grad, 1e-4, "dgamma %d" % k)
        assert_close(outs[k][1], bc[k].grad, 1e-4, "dbeta %d" % k)
        ref = rc[k].grad.sum(dim=(0, 2, 3, 4))
        assert float((outs[k][2].cpu() - ref).abs().max()) <= 1e-4 * float(gc[k].grad.abs().max()) + 1e-5, "conv-bias gradient %d" % k
    for tl2, outs2 in results[1:]:
        for k in range(nterms):
            assert torch.equal(tl2[k]["draw"].t, tl[k]["draw"].t)
            assert all(torch.equal(a, b) for a, b in zip(outs2[k], outs[k])), "a repeated launch differs (ticket words not reset?)"
    if nterms == 2:
        # the two-launch path on the same operands
        prev, K.SMALL_NODE_BACKWARD = K.SMALL_NODE_BACKWARD, False
        try:
            tl3 = terms()
            outs3 = K.affine_act_bwd_gn2(dv[0], tl3, G, dv[1] if two else None)
        finally:
            K.SMALL_NODE_BACKWARD = prev
        for k in range(2):
            assert_close(tl3[k]["draw"].t, tl[k]["draw"].t.cpu(), 2e-5, "d raw %d vs two launches" % k)
            assert_close(outs3[k][0], outs[k][0].cpu(), 2e-5, "dgamma %d vs two launches" % k)
