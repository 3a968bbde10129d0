"""GPU: the trainer's batched mechanics (flat parameters + in-place gradients, one-launch weight
packing, deferred weight-gradient reductions, fused Adam, HIP-graph replay) must reproduce the
reference's training trajectory: losses of 3 Adam steps vs the golden vectors (torch.optim.Adam on
the reference modules) and vs the same net stepped eagerly with torch.optim.Adam."""
import numpy as np
import pytest
import torch

import golden_common as gc
from test_gpu_nets import build_net
from _util import dev
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("key,kind,gname,depth,size,batch,adam", [c for c in gc.net_cases() if c[6] and c[1] == "searched"])
def test_trainer_matches_reference_adam(golden, graph, key, kind, gname, depth, size, batch, adam):
    from nas_3d_unet_amd.train import Trainer
    g = golden("nets")
    net, head = build_net(kind, gname, depth)
    xn, tn = gc.net_batch(key, batch, size)
    x, t = dev(xn), dev(tn)
    tr = Trainer(net, graph=graph)
    losses = [float(tr.step(x, t)) for _ in range(adam)]
    np.testing.assert_allclose(losses, g[key + "/adam_losses"], rtol=0, atol=2e-4)
    for n, q in net.named_parameters():
        ref = float(g[key + "/adam%d/pnorm/%s" % (adam, n)])
        # Adam moves every element by ~lr per step whatever the gradient's size, so an element whose gradient is
        # fp32 noise (e.g. a conv bias ahead of a GroupNorm) may legitimately end up to adam*lr away
        assert abs(float(q.detach().double().norm()) - ref) <= 5e-4 * ref + adam * 1e-3, n


@pytest.mark.parametrize("graph", [False, True])
def test_trainer_on_a_net_with_odd_channel_counts_matches_torch_adam(graph):
    """round 5: init_n_kernels = 6 (stems 18, node widths 12 / 24 / 6: nas.py:13-26, searched.py:55-66 take any) -- Trainer trains the
    net's zero-padded twin (unet.PaddedTwin): three Adam steps against the fp64 oracle + torch.optim.Adam (train.py:49,117-128); the
    trained parameters reach the user's module at check_sync(); padded entries never move"""
    from nas_3d_unet_amd import searched
    from nas_3d_unet_amd.train import Trainer
    from test_gpu_nets import _genotype_for
    cfg = orc.NetCfg(4, 6, 3, 2, 3, True)
    gene = _genotype_for(cfg.n_nodes)
    net = searched.SearchedNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, cfg.channel_change,
                               searched.Genotype(list(gene.down), list(gene.up)))
    from _util import fill_module
    fill_module(net)
    net.last_conv[0].dropout = None
    net = net.cuda()
    rng = np.random.default_rng(23)
    xn = rng.standard_normal((2, 4, 16, 16, 32)).astype(np.float32)
    tn = (rng.uniform(0, 1, (2, 3, 16, 16, 32)) < 0.3).astype(np.float32)
    P = orc.make_params(orc.searched_param_specs(cfg, gene), dtype=torch.float64, requires_grad=True)
    opt = torch.optim.Adam(list(P.values()))
    ref = []
    for _ in range(3):
        opt.zero_grad()
        l = orc.dice_loss(orc.searched_forward(P, torch.from_numpy(xn).double(), gene, cfg), torch.from_numpy(tn).double())
        l.backward(); opt.step()
        ref.append(float(l))
    tr = Trainer(net, graph=graph)
    assert tr._twin is not None and tr.net is not net
    losses = [float(tr.step(dev(xn), dev(tn))) for _ in range(3)]
    np.testing.assert_allclose(losses, ref, rtol=0, atol=2e-4)
    tr.check_sync()            # -> sync_to_module(): the user's module holds the trained parameters in the reference's shapes
    for n, q in net.named_parameters():
        r = float(P[n].detach().norm())
        assert abs(float(q.detach().double().norm()) - r) <= 5e-4 * r + 3e-3, n
    # padded entries are still exactly zero: total |twin| == total |module|
    tot_t = sum(float(p.detach().abs().double().sum()) for p in tr.net.parameters())
    tot_m = sum(float(p.detach().abs().double().sum()) for p in net.parameters())
    assert abs(tot_t - tot_m) <= 1e-6 * tot_m


def test_lr_schedule_follows_through_graph_replay():
    """ReduceLROnPlateau (train.py:50,77): a halved learning rate must take effect inside an already captured HIP graph
    (the Adam kernel reads the rate from a device scalar).  Checked against the same trainer run eagerly."""
    from nas_3d_unet_amd.train import Trainer
    key, kind, gname, depth, size, batch, adam = [c for c in gc.net_cases() if c[6] and c[1] == "searched"][0]
    xn, tn = gc.net_batch(key, batch, size)
    x, t = dev(xn), dev(tn)
    out = []
    for graph in (False, True):
        net, _ = build_net(kind, gname, depth)
        # (the same single-stream schedule on both sides: the side-stream schedule differs from it by an fp32 rounding of the
        # preprocess epilogue backward, which Adam turns into 1e-3 on some weights within three steps -- tools/chaos_probe.py, profiles/r04_chaos_probe.log)
        tr = Trainer(net, graph=graph, side_wgrad=False)
        tr.step(x, t)
        tr.step(x, t)
        tr.set_lr(tr.lr * 0.5)
        tr.step(x, t)
        l = float(tr.step(x, t))
        out.append((l, torch.cat([p.detach().flatten() for p in net.parameters()]).double().cpu()))
    assert abs(out[0][0] - out[1][0]) < 2e-5
    assert float((out[0][1] - out[1][1]).abs().max()) < 2e-5
    # and the schedule really changed the trajectory
    net, _ = build_net(kind, gname, depth)
    tr = Trainer(net, graph=True)
    for _ in range(4):
        l_const = float(tr.step(x, t))
    assert abs(l_const - out[1][0]) > 1e-6


def test_remainder_batch_runs_eagerly():
    """the reference's generator yields a smaller last batch; the captured graph is for one shape, so that step
    must run eagerly instead of broadcasting the sample into the captured batch"""
    from nas_3d_unet_amd.train import Trainer
    key, kind, gname, depth, size, batch, adam = [c for c in gc.net_cases() if c[6] and c[1] == "searched"][0]
    xn, tn = gc.net_batch(key, batch, size)
    x, t = dev(xn), dev(tn)
    out = []
    for graph in (True, False):
        net, _ = build_net(kind, gname, depth)
        tr = Trainer(net, graph=graph)
        tr.step(x, t)
        out.append(float(tr.step(x[:1], t[:1])))
        out.append(float(tr.step(x, t)))
    np.testing.assert_allclose(out[:2], out[2:], rtol=0, atol=2e-6)


@pytest.mark.parametrize("graph", [True, False])
def test_byte_targets_train_exactly_like_float_targets(graph):
    """the reference's generator yields boolean maps (generator.py:230-248) that train.py:118 casts to float: a trainer fed the bytes
    (datastep.patch_batch(target_dtype=torch.uint8)) takes the same steps bit for bit -- the head passes read N3D_U8 targets"""
    from nas_3d_unet_amd.train import Trainer
    key, kind, gname, depth, size, batch, adam = [c for c in gc.net_cases() if c[6] and c[1] == "searched"][0]
    xn, tn = gc.net_batch(key, batch, size)
    tb = (tn > 0.5)
    x = dev(xn)
    losses = []
    for t in (dev(tb.astype(np.float32)), dev(tb.astype(np.uint8))):
        net, _ = build_net(kind, gname, depth)
        tr = Trainer(net, graph=graph)
        losses.append([float(tr.step(x, t)) for _ in range(3)])
        if graph:
            assert tr.input_buffers()[1].dtype == t.dtype
    assert losses[0] == losses[1], losses


def test_data_parallel_property_on_the_hip_path():
    """SURVEY 5.8 / 8(e) on ONE GPU through the HIP kernels: the gradient of a batch of 4 equals the mean of the gradients of
    its two halves (GroupNorm / SE are per sample, Dice is a mean over (b, c) rows) -- what the all-reduce relies on."""
    from nas_3d_unet_amd.train import Trainer
    rng = np.random.default_rng(31)
    xn = rng.standard_normal((4, 4, 32, 32, 32)).astype(np.float32)
    tn = (rng.uniform(0, 1, (4, 3, 32, 32, 32)) < 0.3).astype(np.float32)
    for gname in ("G_CONV", "G_ALL"):
        net, _ = build_net("searched", gname, 4)
        tr = Trainer(net, graph=False)
        tr._fwd_bwd(dev(xn), dev(tn))
        whole = tr.fp.grad.clone()
        tr._fwd_bwd(dev(xn[:2]), dev(tn[:2]))
        half = tr.fp.grad.clone()
        tr._fwd_bwd(dev(xn[2:]), dev(tn[2:]))
        half += tr.fp.grad
        half *= 0.5
        tot = float(whole.double().norm())
        assert float((whole - half).double().norm()) <= 2e-5 * tot, gname


@pytest.mark.parametrize("gname", ["G_ALL", "G_CONV"])
def test_trainer_gradients_equal_autograd_gradients(gname):
    """The trainer's launch batching (pre-packed weights, weight-gradient slabs finalized in ONE launch at the end of backward,
    gradients written in place into the flat buffer) must not change any gradient: every parameter against the plain
    autograd path of the same net.  (Round 2 found the channel sums of a transposed depthwise conv's bias gradient landing
    on its still-pending weight-gradient slabs.)"""
    from nas_3d_unet_amd import loss
    from nas_3d_unet_amd.train import Trainer
    rng = np.random.default_rng(37)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    net, _ = build_net("searched", gname, 4)
    loss.WeightedDiceLoss()(net(x), t).backward()
    ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref.values())))
    net2, _ = build_net("searched", gname, 4)
    tr = Trainer(net2, graph=False)
    for _ in range(2):     # second pass: frozen context (pre-packed weights)
        tr._fwd_bwd(x, t)
        for n, p in net2.named_parameters():
            assert float((p.grad - ref[n]).double().norm()) <= 1e-5 * tot, n


@pytest.mark.parametrize("schedule", ["single-stream", "side-stream"])
def test_benchmarked_configuration_vs_oracle(schedule):
    """BASELINE configs[1] in one piece, as bench.py times it: SearchedNet / G_conv, batch 2, 4x64^3 fp32, TRAIN mode (head
    Dropout3d(0.5) active, searched.py:91-93; the mask made explicit so that the oracle can use the same one), `Trainer(graph=True)`
    on either schedule.  Step 1: loss, probabilities and every parameter's gradient against the CPU oracle
    (oracle.searched_forward(drop_mask=) + Dice, train.py:121-125); steps 2-3 (graph replay): losses and weights against
    torch.optim.Adam on the oracle (train.py:49,128)."""
    from nas_3d_unet_amd.programs import forced_dropout_gate
    from nas_3d_unet_amd.train import Trainer
    from oracle import ref_path as orc
    key = "bench/searched/G_CONV/d4s64/b2/drop"
    xn, tn = gc.net_batch(key, 2, 64)
    gate_n = gc.case_drop_gate(key, 2, 12, 0.5)
    assert 0 < (gate_n == 0).sum() < gate_n.size          # a real mask: some channels dropped, some kept
    # ---- oracle trajectory
    P = orc.make_params(orc.searched_param_specs(orc.DEFAULT_CFG, orc.G_CONV), requires_grad=True)
    opt = torch.optim.Adam(list(P.values()))
    xo, to, go = torch.from_numpy(xn), torch.from_numpy(tn), torch.from_numpy(gate_n)
    ref_losses, ref_p, ref_g = [], None, None
    for it in range(3):
        opt.zero_grad()
        po = orc.searched_forward(P, xo, orc.G_CONV, drop_mask=go)
        lo = orc.dice_loss(po, to)
        lo.backward()
        if it == 0:
            ref_p = po.detach().clone()
            ref_g = {n: q.grad.detach().clone() for n, q in P.items()}
        ref_losses.append(float(lo.detach()))
        opt.step()
        if it == 0:
            ref_w1 = {n: q.detach().clone() for n, q in P.items()}      # the weights after the FIRST update
    # ---- HIP
    net, head = build_net("searched", "G_CONV", 4, keep_dropout=True)
    net.train()
    x, t, gate = dev(xn), dev(tn), dev(gate_n)
    with forced_dropout_gate(gate):
        with torch.no_grad():
            l0, p0 = net.forward_loss(x, t)
        assert abs(float(l0) - ref_losses[0]) < 5e-6
        assert float((p0.cpu() - ref_p).abs().max()) < 2e-5
        tr = Trainer(net, graph=True, side_wgrad="force" if schedule == "side-stream" else False)
        losses = [float(tr.step(x, t))]
        if schedule == "side-stream":
            assert tr._use_side, "the side stream was not accepted on this box"
        total = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ref_g.values())))
        for n, q in net.named_parameters():
            d = float((q.grad.cpu() - ref_g[n]).abs().max())
            # (conv biases ahead of a GroupNorm have a mathematically zero gradient: at 2 x 64^3 the reference's own value is
            # ~1e-3 of fp32 cancellation noise, hence the absolute term)
            assert d <= 2e-4 * float(ref_g[n].abs().max()) + 3e-5 * total, (n, d)
        # the first update, element by element (round-3 review: a norm cannot see a sign error in one layer).  At step 1 Adam moves
        # an element by lr * g / (|g| + eps), i.e. by lr towards -sign(g) wherever |g| >> eps, so every element whose gradient
        # stands clear of the gradient tolerance above (|g| > 1e-3 max|g| + 1e-4 |g_total|: its sign is certain) must match the
        # oracle's torch.optim.Adam to rounding; a wrong sign would be 2 lr = 2e-3 off
        w1 = {n: q.detach().cpu().clone() for n, q in net.named_parameters()}
        checked = 0
        for n, q in w1.items():
            g = ref_g[n]
            sel = g.abs() > 1e-3 * float(g.abs().max()) + 1e-4 * total
            checked += int(sel.sum())
            if bool(sel.any()):
                d = float((q - ref_w1[n])[sel].abs().max())
                assert d <= 2e-5, (n, d, int(sel.sum()))
        assert checked > 1000, checked
        losses += [float(tr.step(x, t)) for _ in range(2)]
        torch.cuda.synchronize()
        tr.check_sync()
    np.testing.assert_allclose(losses, ref_losses, rtol=0, atol=2e-4)
    for n, q in net.named_parameters():
        ref = float(P[n].detach().double().norm())
        # (Adam moves an element whose gradient is fp32 noise by ~lr per step whatever its sign: see test_trainer_matches_reference_adam)
        assert abs(float(q.detach().double().norm()) - ref) <= 5e-4 * ref + 3e-3, n


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_stem_without_its_raw_output_matches_the_stored_form(storage):
    """stem0 ('weight_norm' 1x1x1 conv, nas.py:28 / searched.py:69) in recompute form (n3d_conv_k1_norm_*: statistics pass, then conv +
    normalise in one pass; backward recomputes raw from the 4-channel input and never writes d(raw)) against the ordinary form
    (raw stored, epilogue passes): same loss, same probabilities (the forward arithmetic is identical), every gradient within the
    fp32 reduction-order noise (1e-5 of the gradient norm; the stem's own parameters included)"""
    from nas_3d_unet_amd import programs as P, unet
    rng = np.random.default_rng(73)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    res = []
    for rc in (False, True):
        prev, P.RECOMPUTE_K1 = P.RECOMPUTE_K1, rc
        try:
            net, _ = build_net("searched", "G_CONV", 4)
            unet.set_storage(net, storage)
            l, p = net.forward_loss(x, t)
            l.backward()
            torch.cuda.synchronize()
            res.append((float(l), p.clone(), {n: q.grad.clone() for n, q in net.named_parameters()}))
        finally:
            P.RECOMPUTE_K1 = prev
    if storage == "fp32":
        assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
    else:   # bf16: the stored form rounds raw to bf16 before it normalises, the recompute form normalises the fp32 value
        assert abs(res[0][0] - res[1][0]) < 2e-4
    tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in res[0][2].values())))
    tol = 1e-5 if storage == "fp32" else 2e-2
    for n, g in res[0][2].items():
        assert float((g - res[1][2][n]).double().norm()) <= tol * tot, n
    assert any(n.startswith("stem0") for n in res[0][2])
