"""GPU: the fused head (Dropout3d -> 1x1x1 conv -> sigmoid [-> Dice], nas.py:50-52 / searched.py:91-93 / loss.py:12-14)
through the C ABI (n3d_head_fwd / n3d_head_bwd / n3d_dropout3d_gate) against torch fp32 on the CPU and the reference's
golden Dice vectors."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_common as gc
from _util import assert_close, dev
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


def _reference(x, w, b, gate, t, smooth=1e-6):
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    u = x * gate[:, :, None, None, None] if gate is not None else x
    logits = F.conv3d(u, w, b)
    p = torch.sigmoid(logits)
    loss = orc.dice_loss(p, t, smooth)
    loss.backward()
    return p.detach(), logits.detach(), loss.detach(), x.grad, w.grad, b.grad


@pytest.mark.parametrize("ci,co,shape,batch,use_gate", [(12, 3, (8, 12, 16), 2, True), (12, 3, (32, 32, 32), 2, False), (4, 1, (4, 6, 10), 3, True),
                                                       (24, 2, (6, 6, 6), 1, True), (32, 4, (4, 8, 8), 5, False), (8, 3, (5, 3, 7), 2, True)])
@pytest.mark.parametrize("mode", ["dice", "plain"])
def test_head_matches_torch(ci, co, shape, batch, use_gate, mode):
    from nas_3d_unet_amd import kernels as K
    rng = np.random.default_rng(ci * 100 + co)
    xn = rng.standard_normal((batch, ci) + shape).astype(np.float32)
    wn = (rng.standard_normal((co, ci, 1, 1, 1)) * 0.4).astype(np.float32)
    bn = rng.standard_normal(co).astype(np.float32) * 0.2
    tn = (rng.uniform(0, 1, (batch, co) + shape) < 0.3).astype(np.float32)
    gn = ((rng.uniform(0, 1, (batch, ci)) >= 0.5) / 0.5).astype(np.float32) if use_gate else None
    pr, lr, lossr, dxr, dwr, dbr = _reference(torch.from_numpy(xn), torch.from_numpy(wn), torch.from_numpy(bn),
                                              torch.from_numpy(gn) if use_gate else None, torch.from_numpy(tn))
    xv = K.as_view(dev(xn))
    w, b, t = dev(wn), dev(bn), dev(tn)
    gate = dev(gn) if use_gate else None
    p, logits, sums, loss = K.head_fwd(xv, w, b, gate, t if mode == "dice" else None, want_logits=True)
    assert_close(p, pr, 2e-6, "p")
    assert_close(logits, lr, 2e-6, "logits")
    dx = K.as_view(K.empty_ndhwc(batch, ci, *shape, xv.t.device))
    dw, db = torch.empty_like(w), torch.empty_like(b)
    if mode == "dice":
        assert abs(float(loss) - float(lossr)) < 1e-6
        K.head_bwd(xv, w, b, gate, dx, dw, db, t=t, sums=sums)
    else:
        # d loss / d p computed on the host from the reference probabilities
        pq = pr.clone().requires_grad_(True)
        orc.dice_loss(pq, torch.from_numpy(tn)).backward()
        K.head_bwd(xv, w, b, gate, dx, dw, db, dp=pq.grad.cuda())
    assert_close(dx.t, dxr, 2e-5, "dx")
    assert_close(dw, dwr, 2e-5, "dw")
    assert_close(db, dbr, 2e-5, "db")
    # accumulate mode: dx += ...
    K.head_bwd(xv, w, b, gate, dx, None, None, t=t, sums=sums, accumulate=True) if mode == "dice" else None
    if mode == "dice":
        assert_close(dx.t, 2 * dxr, 2e-5, "dx accumulate")


@pytest.mark.parametrize("ci,shape,batch,use_gate,storage", [(12, (8, 12, 16), 2, True, "f32"), (12, (32, 32, 32), 2, False, "f32"),
                                                             (8, (5, 3, 7), 3, True, "f32"), (12, (16, 16, 16), 2, False, "bf16")])
def test_head_byte_targets_are_bit_identical_to_float_targets(ci, shape, batch, use_gate, storage):
    """generator.py:230-248 yields three BOOLEAN maps and train.py:118 casts them to float: the head passes read the bytes as they are
    (n3d_head.t_dtype = N3D_U8) -- the same loss, Dice sums and gradients bit for bit, a quarter of the target traffic"""
    from nas_3d_unet_amd import kernels as K
    rng = np.random.default_rng(ci + batch)
    xn = rng.standard_normal((batch, ci) + shape).astype(np.float32)
    w = dev((rng.standard_normal((3, ci, 1, 1, 1)) * 0.4).astype(np.float32))
    b = dev(rng.standard_normal(3).astype(np.float32) * 0.2)
    tb = rng.uniform(0, 1, (batch, 3) + shape) < 0.3
    gate = dev(((rng.uniform(0, 1, (batch, ci)) >= 0.5) / 0.5).astype(np.float32)) if use_gate else None
    x = dev(xn)
    if storage == "bf16":
        x = K.empty_ndhwc(batch, ci, *shape, x.device, torch.bfloat16).copy_(x)
    xv = K.as_view(x)
    res = {}
    for name, t in (("float", dev(tb.astype(np.float32))), ("bytes", dev(tb.astype(np.uint8)))):
        p, _, sums, loss = K.head_fwd(xv, w, b, gate, t)
        dx = K.as_view(K.empty_ndhwc(batch, ci, *shape, x.device, x.dtype), bf16_ok=True)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        K.head_bwd(xv, w, b, gate, dx, dw, db, t=t, sums=sums)
        res[name] = [v.clone() for v in (p, sums, loss, dx.t, dw, db)]
    for a, c, what in zip(res["float"], res["bytes"], ("p", "sums", "loss", "dx", "dw", "db")):
        assert torch.equal(a, c), what
    # a strided byte target (a channel slice of a wider buffer) goes through the same element strides
    wide = dev(np.concatenate([tb.astype(np.uint8), np.zeros((batch, 1) + shape, np.uint8)], axis=1))
    _, _, sums2, loss2 = K.head_fwd(xv, w, b, gate, wide[:, :3])
    assert torch.equal(sums2, res["float"][1]) and torch.equal(loss2, res["float"][2])


def test_head_byte_targets_need_three_channels_and_a_known_type():
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd._lib import N3DError
    xv = K.as_view(torch.randn(1, 8, 4, 4, 4).cuda())
    for co, t in ((2, torch.zeros(1, 2, 4, 4, 4, dtype=torch.uint8)), (3, torch.zeros(1, 3, 4, 4, 4, dtype=torch.int32))):
        with pytest.raises(N3DError):
            K.head_fwd(xv, torch.randn(co, 8, 1, 1, 1).cuda(), torch.zeros(co).cuda(), None, t.cuda())


def test_head_bf16_storage():
    """bf16 storage of the head input / its gradient (BASELINE configs[4]): same kernels, conversions on load / store.
    Reference: fp32 torch on the bf16-rounded input; dx compared after rounding to bf16 (one ulp = 2^-8 relative)."""
    from nas_3d_unet_amd import kernels as K
    rng = np.random.default_rng(5)
    batch, ci, co, shape = 2, 12, 3, (8, 16, 16)
    xb = torch.from_numpy(rng.standard_normal((batch, ci) + shape).astype(np.float32)).bfloat16()
    wn = (rng.standard_normal((co, ci, 1, 1, 1)) * 0.4).astype(np.float32)
    bn = rng.standard_normal(co).astype(np.float32) * 0.2
    tn = (rng.uniform(0, 1, (batch, co) + shape) < 0.3).astype(np.float32)
    pr, lr, lossr, dxr, dwr, dbr = _reference(xb.float(), torch.from_numpy(wn), torch.from_numpy(bn), None, torch.from_numpy(tn))
    xd = K.empty_ndhwc(batch, ci, *shape, torch.device("cuda"), torch.bfloat16)
    xd.copy_(xb.cuda())
    xv = K.as_view(xd, bf16_ok=True)
    w, b, t = dev(wn), dev(bn), dev(tn)
    p, _, sums, loss = K.head_fwd(xv, w, b, None, t)
    assert_close(p, pr, 2e-6, "p")
    assert abs(float(loss) - float(lossr)) < 1e-6
    dx = K.as_view(K.empty_ndhwc(batch, ci, *shape, xv.t.device, torch.bfloat16), bf16_ok=True)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    K.head_bwd(xv, w, b, None, dx, dw, db, t=t, sums=sums)
    assert_close(dw, dwr, 2e-5, "dw")
    assert_close(dx.t.float(), dxr, 2.0 ** -8, "dx (bf16)")
    assert torch.equal(dx.t.cpu(), dxr.bfloat16()) or float((dx.t.float().cpu() - dxr.bfloat16().float()).abs().max()) <= float(dxr.abs().max()) * 2.0 ** -7


@pytest.mark.parametrize("key,shape", gc.dice_cases())
def test_dice_loss_golden(golden, key, shape):
    """WeightedDiceLoss on the GPU (n3d_dice_fwd / n3d_dice_bwd) against the reference's own loss and d loss / d p"""
    from nas_3d_unet_amd import loss
    g = golden("small")
    p = dev(gc.case_probs(key, shape), True)
    t = dev(gc.case_targets(key, shape))
    l = loss.WeightedDiceLoss()(p, t)
    l.backward()
    assert abs(float(l) - float(g[key + "/loss"])) < 1e-6
    assert_close(p.grad, g[key + "/dp"], 1e-5, key + " dp")
    # the same loss through NDHWC-strided probabilities (what the op-by-op head emits)
    p2 = dev(gc.case_probs(key, shape)).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    l2 = loss.WeightedDiceLoss()(p2, t)
    l2.backward()
    assert abs(float(l2) - float(g[key + "/loss"])) < 1e-6
    assert_close(p2.grad, g[key + "/dp"], 1e-5, key + " dp (NDHWC)")


def test_dropout_gate_generator():
    """n3d_dropout3d_gate: values are exactly {0, 1/(1-p)}, equal to the host-callable generator n3d_dropout3d_uniform for
    the same (seed, counter, index), the counter advances per launch, keep rate ~ 1 - p"""
    from nas_3d_unet_amd import _lib, kernels as K
    lib = _lib.load()
    seed = 0x1234567887654321
    st = torch.tensor([0x87654321 - (1 << 32), 0x12345678, 0], dtype=torch.int32, device="cuda")
    B, Cc, p = 64, 12, 0.5
    g0 = K.dropout3d_gate(st, p, B, Cc).cpu().numpy().ravel()
    g1 = K.dropout3d_gate(st, p, B, Cc).cpu().numpy().ravel()
    assert int(st[2]) == 2
    for counter, got in ((0, g0), (1, g1)):
        want = np.array([2.0 if lib.n3d_dropout3d_uniform(C.c_uint64(seed), counter, i) >= p else 0.0 for i in range(B * Cc)], dtype=np.float32)
        assert np.array_equal(got, want)
    assert not np.array_equal(g0, g1)
    assert 0.4 < (g0 > 0).mean() < 0.6


def test_train_mode_head_draws_fresh_masks_under_graph_replay():
    """Trainer(graph=True) with the head Dropout3d active: the mask comes from device state, so consecutive replays see
    different masks (losses differ run to run while the weights are frozen by lr = 0) and the step stays finite"""
    from test_gpu_nets import build_net
    from nas_3d_unet_amd.train import Trainer
    net, head = build_net("searched", "G_CONV", 2, keep_dropout=True)
    net.train()
    rng = np.random.default_rng(3)
    x = dev(rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32))
    tr = Trainer(net, lr=0.0, graph=True)
    losses = [float(tr.step(x, t)) for _ in range(6)]
    assert all(np.isfinite(losses))
    assert len({round(l, 7) for l in losses}) >= 3, losses


# ---- node-planar head input (include/n3d.h, n3d_head.node_c; fused.PLANAR_LAST) ----------------------------------------------
@pytest.mark.parametrize("cn,nn,co,shape,batch", [(4, 3, 3, (8, 12, 16), 2), (8, 3, 2, (6, 6, 6), 1), (4, 1, 1, (4, 6, 10), 3), (8, 4, 4, (4, 8, 8), 2)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_head_on_node_planar_input_equals_the_concatenated_one(cn, nn, co, shape, batch, dtype):
    """the same head launches on `nn` dense node tensors (three pointers) and on their concatenation (one pitched record per
    voxel): probabilities, loss, dx, dw, db must be bit-identical -- only addresses differ (cell.py:82, searched.py:51)"""
    from nas_3d_unet_amd import kernels as K
    rng = np.random.default_rng(cn * 10 + nn)
    ci = cn * nn
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    xn = rng.standard_normal((batch, ci) + shape).astype(np.float32)
    w = dev((rng.standard_normal((co, ci, 1, 1, 1)) * 0.4).astype(np.float32))
    b = dev(rng.standard_normal(co).astype(np.float32) * 0.2)
    t = dev((rng.uniform(0, 1, (batch, co) + shape) < 0.3).astype(np.float32))
    gate = dev(((rng.uniform(0, 1, (batch, ci)) >= 0.5) / 0.5).astype(np.float32))
    cat = K.empty_ndhwc(batch, ci, *shape, torch.device("cuda"), td)
    cat.copy_(dev(xn))
    xv = K.as_view(cat)
    pl = K.empty_planar(nn, batch, cn, *shape, torch.device("cuda"), td)
    for k in range(nn):
        pl.nodes[k].t.copy_(cat[:, k * cn:(k + 1) * cn])
    assert K.as_planar(pl.t).t.data_ptr() == pl.t.data_ptr()
    out = []
    for x in (xv, pl):
        p, logits, sums, loss = K.head_fwd(x, w, b, gate, t, want_logits=True)
        dx = K.as_view(K.empty_ndhwc(batch, ci, *shape, torch.device("cuda"), td)) if x is xv else K.empty_planar(nn, batch, cn, *shape, torch.device("cuda"), td)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        K.head_bwd(x, w, b, gate, dx, dw, db, t=t, sums=sums)
        torch.cuda.synchronize()
        dxc = dx.t if x is xv else torch.cat([n.t for n in dx.nodes], dim=1)
        out.append((p, logits, float(loss), dxc.float(), dw, db))
    for a_, b_ in zip(out[0], out[1]):
        assert (a_ == b_) if isinstance(a_, float) else torch.equal(a_, b_)


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
@pytest.mark.parametrize("path", ["autograd", "pipeline"])
def test_last_cell_node_planar_equals_concatenated(storage, path):
    """SearchedNet with the last cell's nodes kept dense (fused.PLANAR_LAST) against the concatenation buffer: same kernels on
    the same values -- loss and every parameter gradient bit-identical, through autograd (forward_loss) and through the trainers'
    autograd-free pipeline"""
    from nas_3d_unet_amd import fused, unet
    from nas_3d_unet_amd.train import Trainer
    from test_gpu_nets import build_net
    rng = np.random.default_rng(71)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    res = []
    for planar in (False, True):
        prev, fused.PLANAR_LAST = fused.PLANAR_LAST, planar
        try:
            net, _ = build_net("searched", "G_CONV", 4)
            unet.set_storage(net, storage)
            if path == "autograd":
                l, p = net.forward_loss(x, t)
                l.backward()
                grads = {n: q.grad.clone() for n, q in net.named_parameters()}
            else:
                tr = Trainer(net, graph=False, side_wgrad=False)
                assert tr._direct_ok()
                l = tr._pipeline(x, t, cell_hook=lambda k: None)
                tr.ctx.flush_final()
                grads = {"flat": tr.fp.grad.clone()}
            torch.cuda.synchronize()
            res.append((float(l), grads))
        finally:
            fused.PLANAR_LAST = prev
    assert res[0][0] == res[1][0]
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


def test_supernet_last_cell_node_planar_equals_concatenated():
    """the supernet's last cell with dense node tensors (a node sums 10-22 weighted primitives into its buffer, cell.py:76-82): loss,
    alpha gradients and every parameter gradient bit-identical to the concatenation buffer"""
    from nas_3d_unet_amd import fused, nas
    from _util import fill_module
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(79)
    x = dev(rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32))
    res = []
    for planar in (False, True):
        prev, fused.PLANAR_LAST = fused.PLANAR_LAST, planar
        try:
            net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
            fill_module(net)
            net.kernel.last_conv[0].dropout = None
            net = net.cuda()
            with torch.no_grad():
                for a in net.alphas():
                    a.copy_(torch.from_numpy(np.random.default_rng(3).standard_normal(tuple(a.shape)).astype(np.float32)).cuda())
            l, _ = net.forward_loss(x, t)
            l.backward()
            torch.cuda.synchronize()
            res.append((float(l), {n: q.grad.clone() for n, q in net.named_parameters() if q.grad is not None}))
        finally:
            fused.PLANAR_LAST = prev
    assert res[0][0] == res[1][0] and res[0][1].keys() == res[1][1].keys()
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


# ---- node-planar tensors as operands of the 1x1x1 preprocess convs (include/n3d.h, "node-planar tensors"; fused.PLANAR_INNER) -----------
@pytest.mark.parametrize("storage", ["fp32", "bf16"])
@pytest.mark.parametrize("cn,co,shape", [(8, 4, (32, 32, 32)), (4, 8, (32, 32, 64)), (8, 4, (33, 32, 32))])
def test_k1_conv_on_node_planar_operands_equals_the_concatenated_form(storage, cn, co, shape):
    """the 1x1x1 streaming kernels with a node-planar x side (pitch = cn < channels = 3 cn: three dense node tensors) against the same
    calls on the concatenation buffer: forward (ReLU on load, statistics rows), data gradient (ReLU mask, accumulate) and weight
    gradient -- the same arithmetic in the same order, bit for bit; a ragged voxel count included"""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd._lib import ACCUMULATE, RELU_IN
    dt = torch.bfloat16 if storage == "bf16" else torch.float32
    rng = np.random.default_rng(5 + cn + shape[0])
    B, nn = 2, 3
    d = torch.device("cuda")
    xn = rng.standard_normal((B, nn * cn) + shape).astype(np.float32)
    with K.storage(dt):
        xc = K.as_view(K.empty_ndhwc(B, nn * cn, *shape, d))
        xc.t.copy_(torch.from_numpy(xn).to(d))
        xp = K.empty_planar(nn, B, cn, *shape, d)
        for k in range(nn):
            xp.nodes[k].t.copy_(xc.t[:, k * cn:(k + 1) * cn])
        assert xp.ld == cn and xp.C == nn * cn
        w = torch.from_numpy((rng.standard_normal((co, nn * cn, 1, 1, 1)) * 0.3).astype(np.float32)).to(d)
        b = torch.from_numpy(rng.standard_normal(co).astype(np.float32) * 0.1).to(d)
        g = K.conv_geom(B, *shape, nn * cn, co, 1, 1, 1, 0)
        # forward
        ys = []
        for x in (xc, xp):
            y = K.as_view(K.empty_ndhwc(B, co, *shape, d))
            rows = K.conv_stats_rows(g, False, 0, x, y)
            stats = torch.zeros((B, rows, co, 2), dtype=torch.float64, device=d)
            K.conv_fwd(g, x, w, b, y, RELU_IN, None, stats, False)
            ys.append((y.t.clone(), stats.clone()))
        assert torch.equal(ys[0][0], ys[1][0]) and torch.equal(ys[0][1], ys[1][1])
        ref = torch.nn.functional.conv3d(torch.relu(xc.t.float()), w, b)
        assert float((ys[1][0].float() - ref).abs().max()) <= (2.0 ** -7 if storage == "bf16" else 1e-4) * float(ref.abs().max())
        # data gradient: dx (+)= relu'(x) * W^T dy, into the concatenation buffer / into the three node tensors
        dy = K.as_view(K.empty_ndhwc(B, co, *shape, d))
        dy.t.copy_(torch.from_numpy(rng.standard_normal((B, co) + shape).astype(np.float32)).to(d))
        d0 = torch.from_numpy(rng.standard_normal((B, nn * cn) + shape).astype(np.float32)).to(d)
        for acc in (0, ACCUMULATE):
            dxc = K.as_view(K.empty_ndhwc(B, nn * cn, *shape, d))
            dxc.t.copy_(d0)
            dxp = K.like(xp)
            assert isinstance(dxp, K.Planar)
            for k in range(nn):
                dxp.nodes[k].t.copy_(dxc.t[:, k * cn:(k + 1) * cn])
            K.conv_bwd_data(g, dy, w, dxc, acc, xc, None, False)
            K.conv_bwd_data(g, dy, w, dxp, acc, xp, None, False)
            for k in range(nn):
                assert torch.equal(dxp.nodes[k].t, dxc.t[:, k * cn:(k + 1) * cn]), (acc, k)
        # weight gradient
        dws = []
        for x in (xc, xp):
            dw, db = torch.zeros_like(w), torch.zeros_like(b)
            K.conv_bwd_weight(g, x, dy, dw, db, RELU_IN, None, False, defer=False)
            dws.append((dw, db))
        torch.cuda.synchronize()
        assert torch.equal(dws[0][0], dws[1][0]) and torch.equal(dws[0][1], dws[1][1])


def test_node_planar_operands_are_refused_where_no_kernel_takes_them():
    """a pitch smaller than the channel count reaches only the 1x1x1 streaming kernels: anything else must fail loudly, not misread"""
    from nas_3d_unet_amd import kernels as K
    d = torch.device("cuda")
    xp = K.empty_planar(3, 2, 8, 16, 16, 16, d)          # 4096 voxels: below the streaming kernels' volume
    w = torch.zeros((4, 24, 1, 1, 1), device=d)
    y = K.as_view(K.empty_ndhwc(2, 4, 16, 16, 16, d))
    with pytest.raises(K.N3DError, match="node-planar"):
        K.conv_fwd(K.conv_geom(2, 16, 16, 16, 24, 4, 1, 1, 1, 0), xp, w, None, y, 0, None, None, False)
    xp = K.empty_planar(3, 2, 8, 32, 32, 32, d)
    w3 = torch.zeros((4, 24, 3, 3, 3), device=d)
    y = K.as_view(K.empty_ndhwc(2, 4, 32, 32, 32, d))
    with pytest.raises(K.N3DError, match="node-planar|pitch"):
        K.conv_fwd(K.conv_geom(2, 32, 32, 32, 24, 4, 3, 1, 1, 1), xp, w3, None, y, 0, None, None, False)


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
@pytest.mark.parametrize("path", ["autograd", "pipeline"])
def test_inner_cell_node_planar_equals_concatenated(storage, path):
    """SearchedNet at 4 x 64^3 with up-cell 3's nodes kept dense (fused.PLANAR_INNER: 3 x 8 channels on 32^3 voxels, read by the last cell's
    x1 preprocess conv) against the concatenation buffer: loss and every parameter gradient bit-identical"""
    from nas_3d_unet_amd import fused, kernels as K, unet
    from nas_3d_unet_amd.train import Trainer
    from test_gpu_nets import build_net
    rng = np.random.default_rng(171)
    x = dev(rng.standard_normal((2, 4, 64, 64, 64)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 64, 64, 64)) < 0.3).astype(np.float32))
    res, made = [], []
    orig = K.empty_planar
    for planar in (False, True):
        prev, fused.PLANAR_INNER = fused.PLANAR_INNER, planar
        K.empty_planar = lambda *a, **k: (made.append((planar, a[2])), orig(*a, **k))[1]
        try:
            net, _ = build_net("searched", "G_CONV", 4)
            unet.set_storage(net, storage)
            if path == "autograd":
                l, p = net.forward_loss(x, t)
                l.backward()
                grads = {n: q.grad.clone() for n, q in net.named_parameters()}
            else:
                tr = Trainer(net, graph=False, side_wgrad=False)
                l = tr._pipeline(x, t, cell_hook=lambda k: None)
                tr.ctx.flush_final()
                grads = {"flat": tr.fp.grad.clone()}
            torch.cuda.synchronize()
            res.append((float(l), grads))
        finally:
            fused.PLANAR_INNER = prev
            K.empty_planar = orig
    assert any(pl and cn == 8 for pl, cn in made), "the inner cell never took the node-planar form"
    assert not any((not pl) and cn == 8 for pl, cn in made)
    assert res[0][0] == res[1][0]
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n
