"""GPU: supernet search step (architecture pass on the validation batch, weight pass on the training batch,
search.py:211-238) against the CPU oracle driven by torch.optim.Adam, depth-2 supernet on 16^3 patches."""
import numpy as np
import pytest
import torch

from _util import fill_module
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph", [False, True])
def test_search_step_matches_oracle(graph):
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(11)
    mk = lambda: (rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32),
                  (rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32))
    (xn, tn), (vxn, vtn) = mk(), mk()
    # ---- oracle: two Adam optimisers, alternating passes
    P = orc.make_params(orc.supernet_param_specs(cfg), requires_grad=True)
    alphas = [P[n] for n in ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up")]
    kern = [v for n, v in P.items() if n.startswith("kernel.")]
    oa, ok = torch.optim.Adam(alphas), torch.optim.Adam(kern)
    ref = []
    for _ in range(2):
        oa.zero_grad()
        la = orc.dice_loss(orc.supernet_forward(P, torch.from_numpy(vxn), cfg), torch.from_numpy(vtn))
        la.backward(); oa.step()
        ok.zero_grad()
        lw = orc.dice_loss(orc.supernet_forward(P, torch.from_numpy(xn), cfg), torch.from_numpy(tn))
        lw.backward(); ok.step()
        ref.append((float(la), float(lw)))
    # ---- HIP
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net)
    net.kernel.last_conv[0].dropout = None
    net = net.cuda()
    tr = SearchTrainer(net, graph=graph)
    x, t, vx, vt = (torch.from_numpy(a).cuda() for a in (xn, tn, vxn, vtn))
    got = []
    for _ in range(2):
        la, lw = tr.step(x, t, vx, vt)
        got.append((float(la), float(lw)))
    np.testing.assert_allclose(np.array(got), np.array(ref), rtol=0, atol=3e-4)
    for n in ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up"):
        mine = getattr(net, n).detach().cpu().numpy()
        # Adam moves each alpha by ~lr per step; compare with slack for sign flips of noise-level gradients
        assert np.abs(mine - P[n].detach().numpy()).max() <= 2.5e-3, n
    gene = net.get_gene()
    assert len(gene.down) == 6 and len(gene.up) == 6


@pytest.mark.parametrize("graph", [False, True])
def test_search_step_with_byte_targets_equals_float_targets(graph):
    """both passes of the search step (search.py:211-238) on the generator's boolean maps as bytes: the same losses bit for bit"""
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(12)
    xn, vxn = (rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32) for _ in range(2))
    tb, vtb = (rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3 for _ in range(2))
    out = []
    for dt in (np.float32, np.uint8):
        net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
        fill_module(net)
        net.kernel.last_conv[0].dropout = None
        tr = SearchTrainer(net.cuda(), graph=graph)
        x, t, vx, vt = (torch.from_numpy(a).cuda() for a in (xn, tb.astype(dt), vxn, vtb.astype(dt)))
        out.append([tuple(float(v) for v in tr.step(x, t, vx, vt)) for _ in range(2)])
    assert out[0] == out[1], out


@pytest.mark.parametrize("graph", [True, False])
def test_search_trajectory_depth4_matches_reference(golden, graph):
    """BASELINE configs[2] as benchmarked -- depth-4 supernet through SearchTrainer (HIP-graph replay) -- against the trajectory
    recorded from the reference modules + torch.optim.Adam (tests/golden/make_golden.py gen_nets2): the ALPHA GRADIENTS after
    each architecture pass element by element, both losses per step, the kernel-weight gradient norms of the weight pass."""
    import golden_common as gc
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    g = golden("nets2")
    key, depth, size, batch, steps = gc.search_cases()[0]
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net)
    net.kernel.last_conv[0].dropout = None
    net = net.cuda()
    tr = SearchTrainer(net, graph=graph)
    x, t, vx, vt = (torch.from_numpy(a).cuda() for a in gc.search_batches(key, batch, size))
    for step in range(steps):
        la, lw = tr.step(x, t, vx, vt)
        np.testing.assert_allclose([float(la), float(lw)], g["%s/step%d/losses" % (key, step)], rtol=0, atol=2e-5)
        scale = max(np.abs(g["%s/step%d/dalpha/%s" % (key, step, n)]).max() for n in ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up"))
        for n in ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up"):
            ref = g["%s/step%d/dalpha/%s" % (key, step, n)]
            mine = getattr(net, n).grad.detach().cpu().numpy()   # the arch pass's gradient (the weight pass leaves it alone)
            # step 0: same weights on both sides; step 1: after one Adam step of lr-sized moves driven by fp32-noise-level
            # differences, so a looser bound
            # relative to the largest alpha gradient of the pass (alpha2_down's are ~50x smaller than the others here)
            tol = (5e-4 * np.abs(ref).max()) if step == 0 else 5e-3 * scale
            assert np.abs(mine - ref).max() <= tol, (n, step, np.abs(mine - ref).max(), np.abs(ref).max())
            # rows no edge uses have an exactly zero gradient (cell.py:79-80)
            assert np.array_equal(mine[np.abs(ref).max(axis=1) == 0], np.zeros_like(mine[np.abs(ref).max(axis=1) == 0]))
        # the weight pass runs AFTER the alpha update: Adam moves every alpha by ~lr whatever its gradient's size, so alphas whose
        # gradient is fp32 noise may move the other way than in the reference and the two supernets differ at the 1e-3 level
        # from here on (the weight-pass gradients on identical weights are pinned by test_gpu_nets' supernet cases)
        total = float(g["%s/step%d/gnorm_total" % (key, step)])
        mine_total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in net.kernel.parameters())))
        assert abs(mine_total - total) <= 5e-3 * total, (step, mine_total, total)
        for n, q in net.kernel.named_parameters():
            ref = float(g["%s/step%d/gnorm/kernel.%s" % (key, step, n)])
            mine = float(q.grad.double().norm())
            assert abs(mine - ref) <= 1e-2 * ref + 2e-3 * total, (n, step, mine, ref)


@pytest.mark.parametrize("graph", [False, True])
def test_search_step_on_a_supernet_with_odd_channel_counts_matches_oracle(graph):
    """round 5: init_n_kernels = 6 (nas.py:13-26 takes any) -- SearchTrainer trains the kernel net's zero-padded twin (unet.PaddedTwin) with
    the shell's own alphas: two search steps against the fp64 oracle + two torch.optim.Adam (search.py:211-238); the kernel weights reach
    shell.kernel at check_sync(), padded entries never move"""
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.NetCfg(4, 6, 3, 2, 3, True)
    rng = np.random.default_rng(31)
    mk = lambda: (rng.standard_normal((2, 4, 8, 8, 16)).astype(np.float32), (rng.uniform(0, 1, (2, 3, 8, 8, 16)) < 0.3).astype(np.float32))
    (xn, tn), (vxn, vtn) = mk(), mk()
    P = orc.make_params(orc.supernet_param_specs(cfg), dtype=torch.float64, requires_grad=True)
    anames = ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up")
    oa, ok = torch.optim.Adam([P[n] for n in anames]), torch.optim.Adam([v for n, v in P.items() if n.startswith("kernel.")])
    D = lambda a: torch.from_numpy(a).double()
    ref = []
    for _ in range(2):
        oa.zero_grad()
        la = orc.dice_loss(orc.supernet_forward(P, D(vxn), cfg), D(vtn)); la.backward(); oa.step()
        ok.zero_grad()
        lw = orc.dice_loss(orc.supernet_forward(P, D(xn), cfg), D(tn)); lw.backward(); ok.step()
        ref.append((float(la.detach()), float(lw.detach())))
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net)
    net.kernel.last_conv[0].dropout = None
    net = net.cuda()
    tr = SearchTrainer(net, graph=graph)
    assert tr._twin is not None
    x, t, vx, vt = (torch.from_numpy(a).cuda() for a in (xn, tn, vxn, vtn))
    got = [tuple(float(v) for v in tr.step(x, t, vx, vt)) for _ in range(2)]
    np.testing.assert_allclose(np.array(got), np.array(ref), rtol=0, atol=3e-4)
    tr.check_sync()
    for n in anames:
        assert np.abs(getattr(net, n).detach().cpu().numpy() - P[n].detach().numpy()).max() <= 2.5e-3, n
    for n, q in net.kernel.named_parameters():
        r = float(P["kernel." + n].detach().norm())
        assert abs(float(q.detach().double().norm()) - r) <= 5e-4 * r + 3e-3, n
    tot_t = sum(float(p.detach().abs().double().sum()) for p in tr._twin.twin.parameters())
    tot_m = sum(float(p.detach().abs().double().sum()) for p in net.kernel.parameters())
    assert abs(tot_t - tot_m) <= 1e-6 * tot_m


@pytest.mark.parametrize("schedule", ["single-stream", "side-stream"])
def test_search_benchmarked_configuration_vs_reference(golden, schedule):
    """BASELINE configs[2] in ONE piece at the benchmarked size, as bench.py times it: depth-4 supernet, 2 + 2 patches of 4x64^3
    fp32, TRAIN mode (head Dropout3d(0.1), nas.py:50-52; the masks made explicit, one per pass), SearchTrainer(graph=True) pinned
    to either schedule, against the trajectory the REFERENCE modules + two torch.optim.Adam produced (tests/golden/make_golden.py
    gen_search64, search.py:211-238): alpha gradients element by element, both losses, small kernel-weight gradients of the first
    weight pass element by element, every kernel-weight gradient norm, alphas after each step.  Tolerances: those of
    test_search_trajectory_depth4_matches_reference."""
    import golden_common as gc
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.programs import forced_dropout_gate
    from nas_3d_unet_amd.train import SearchTrainer
    g = golden("search64")
    key, depth, size, batch, steps, p = gc.search_bench_case()
    cfg = orc.DEFAULT_CFG._replace(depth=depth)
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net)
    assert abs(net.kernel.last_conv[0].dropout.p - p) < 1e-12
    net = net.cuda()
    net.train()
    gates = gc.search_drop_gates(key, steps, batch, cfg.n_nodes * cfg.init_n_kernels, p)
    assert all(0 < (ga == 0).sum() and 0 < (gw == 0).sum() for ga, gw in gates)     # real masks
    ga, gw = torch.from_numpy(gates[0][0]).cuda(), torch.from_numpy(gates[0][1]).cuda()
    tr = SearchTrainer(net, graph=True, side_wgrad="force" if schedule == "side-stream" else False)
    x, t, vx, vt = (torch.from_numpy(a).cuda() for a in gc.search_batches(key, batch, size))
    anames = ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up")
    with forced_dropout_gate({"arch": ga, "weight": gw}):
        for step in range(steps):
            ga.copy_(torch.from_numpy(gates[step][0]))      # the captured graphs read these two tensors
            gw.copy_(torch.from_numpy(gates[step][1]))
            la, lw = tr.step(x, t, vx, vt)
            if schedule == "side-stream":
                assert tr._use_side, "the side streams were not accepted on this box"
            else:
                assert not tr._use_side and tr.side is None
            np.testing.assert_allclose([float(la), float(lw)], g["%s/step%d/losses" % (key, step)], rtol=0, atol=2e-5)
            scale = max(np.abs(g["%s/step%d/dalpha/%s" % (key, step, n)]).max() for n in anames)
            for n in anames:
                ref = g["%s/step%d/dalpha/%s" % (key, step, n)]
                mine = getattr(net, n).grad.detach().cpu().numpy()
                tol = (5e-4 * np.abs(ref).max()) if step == 0 else 5e-3 * scale
                assert np.abs(mine - ref).max() <= tol, (n, step, np.abs(mine - ref).max(), np.abs(ref).max())
                unused = np.abs(ref).max(axis=1) == 0
                assert np.array_equal(mine[unused], np.zeros_like(mine[unused]))
                # the alphas after the step's Adam update: lr-sized moves, so a sign flip of a noise-level gradient is 2e-3
                a_ref = g["%s/step%d/alpha/%s" % (key, step, n)]
                assert np.abs(getattr(net, n).detach().cpu().numpy() - a_ref).max() <= 2.5e-3 * (step + 1), (n, step)
            total = float(g["%s/step%d/gnorm_total" % (key, step)])
            mine_total = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in net.kernel.parameters())))
            assert abs(mine_total - total) <= 5e-3 * total, (step, mine_total, total)
            checked = 0
            for n, q in net.kernel.named_parameters():
                ref = float(g["%s/step%d/gnorm/kernel.%s" % (key, step, n)])
                mine = float(q.grad.double().norm())
                assert abs(mine - ref) <= 1e-2 * ref + 2e-3 * total, (n, step, mine, ref)
                gk = "%s/step0/grad/kernel.%s" % (key, n)
                if step == 0 and gk in g.files:
                    # element by element (a norm cannot see a sign error): stems, preprocess convs and the head
                    d = np.abs(q.grad.cpu().numpy() - g[gk]).max()
                    assert d <= 1e-2 * np.abs(g[gk]).max() + 2e-4 * total, (n, d, np.abs(g[gk]).max())
                    checked += 1
            assert step > 0 or checked >= 8, checked
        torch.cuda.synchronize()
        tr.check_sync()
    assert tr.sync_timeouts() == 0
    for n, q in net.kernel.named_parameters():
        ref = float(g["%s/final/pnorm/kernel.%s" % (key, n)])
        assert abs(float(q.detach().double().norm()) - ref) <= 5e-4 * ref + 3e-3, n


def test_search_step_with_shared_normal_alphas_matches_oracle():
    """normal_w_share=True (nas.py:109-113): alpha1_up IS alpha1_down, so its gradient collects the stride-1 edges of both the
    down and the up cells.  One search step through SearchTrainer against the oracle + torch.optim.Adam."""
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(19)
    mk = lambda: (rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32), (rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32))
    (xn, tn), (vxn, vtn) = mk(), mk()
    P = orc.make_params(orc.supernet_param_specs(cfg, True), requires_grad=True)
    anames = [n for n in ("alpha2_down", "alpha2_up", "alpha1_down") if n in P]
    la = orc.dice_loss(orc.supernet_forward(P, torch.from_numpy(vxn), cfg, normal_w_share=True), torch.from_numpy(vtn))
    la.backward()
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, True, cfg.channel_change)
    assert net.alpha1_up is net.alpha1_down and len(list(net.alphas())) == 3
    fill_module(net)
    net.kernel.last_conv[0].dropout = None
    net = net.cuda()
    tr = SearchTrainer(net, graph=False)
    la_hip, _ = tr.step(*(torch.from_numpy(a).cuda() for a in (xn, tn, vxn, vtn)))
    assert abs(float(la_hip) - float(la)) < 5e-6
    for n in anames:
        ref = P[n].grad.numpy()
        mine = getattr(net, n).grad.cpu().numpy()
        assert np.abs(mine - ref).max() <= 5e-4 * np.abs(ref).max(), n


def test_search_trainer_resumes_loaded_adam_state_in_graph_mode():
    """the first graph-mode step must not wipe optimizer state loaded before it (the resume path of
    search.py:108-127).  Load moments + step counters, take one graph step, compare with the same step taken eagerly."""
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(23)
    mk = lambda: (rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32), (rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32))
    (xn, tn), (vxn, vtn) = mk(), mk()
    batches = [torch.from_numpy(a).cuda() for a in (xn, tn, vxn, vtn)]
    res = []
    for graph in (False, True):
        net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
        fill_module(net)
        net.kernel.last_conv[0].dropout = None
        tr = SearchTrainer(net.cuda(), graph=graph)
        gen = torch.Generator().manual_seed(1)
        tr.fp.exp_avg.copy_(torch.randn(tr.fp.numel, generator=gen) * 1e-3)
        tr.fp.exp_avg_sq.copy_(torch.rand(tr.fp.numel, generator=gen) * 1e-5)
        tr.a_m.copy_(torch.randn(tr.a_m.numel(), generator=gen) * 1e-3)
        tr.a_v.copy_(torch.rand(tr.a_v.numel(), generator=gen) * 1e-5)
        tr.fp.step.fill_(40); tr.a_step.fill_(40)
        tr.step(*batches)
        assert int(tr.fp.step) == 41 and int(tr.a_step) == 41
        res.append((tr.fp.flat.clone(), tr.aflat.clone(), tr.fp.exp_avg.clone(), tr.a_m.clone()))
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-6 + 1e-5 * float(a.abs().max())


def test_search_trainer_remainder_batch_runs_eagerly():
    """a last batch of another size must not be broadcast into the captured batch"""
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(29)
    mk = lambda b: (torch.from_numpy(rng.standard_normal((b, 4, 16, 16, 16)).astype(np.float32)).cuda(),
                    torch.from_numpy((rng.uniform(0, 1, (b, 3, 16, 16, 16)) < 0.3).astype(np.float32)).cuda())
    (x, t), (vx, vt), (x1, t1), (vx1, vt1) = mk(2), mk(2), mk(1), mk(1)
    out = []
    for graph in (True, False):
        net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
        fill_module(net)
        net.kernel.last_conv[0].dropout = None
        tr = SearchTrainer(net.cuda(), graph=graph)
        tr.step(x, t, vx, vt)
        la, lw = tr.step(x1, t1, vx1, vt1)
        out.append((float(la), float(lw)))
    np.testing.assert_allclose(out[0], out[1], rtol=0, atol=2e-6)
