"""GPU: the side-stream schedule of the weight-gradient kernels (train.SideSchedule, n3d_sync_signal / n3d_sync_wait).

The hand-off primitives must order two streams and must never hang; the schedule must not change a gradient (the reference
runs one stream, train.py:117-128): every parameter gradient against the plain one-stream trainer, eagerly and replayed from
the three captured graphs, for the searched net and for the supernet's weight pass (search.py:233-238)."""
import numpy as np
import pytest
import torch

from test_gpu_nets import build_net
from _util import dev, fill_module
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


def _words(n=16):
    w = torch.zeros(n, dtype=torch.int32, device="cuda")   # [0] step of stream A, [1] time-outs, [2] step of stream B, [4..] flags
    w[0] = 1
    w[2] = 1
    return w


def test_sync_handoff_orders_two_streams():
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.train import reserve_side_streams
    w = _words()
    p = lambda i: w.data_ptr() + 4 * i
    # the process's reserved side streams: two arbitrary torch-pool streams may share a hardware queue (late in a long test session
    # they did, and every wait below ran into its time-out) -- which is why the trainers never use those for a hand-off
    a, b = reserve_side_streams(torch.device("cuda", torch.cuda.current_device()))[:2]
    src = torch.zeros(1 << 22, device="cuda")
    dst = torch.zeros(4, 1 << 22, device="cuda")
    torch.cuda.synchronize()
    for r in range(4):
        # the consumer is enqueued FIRST: without the device-side wait it would copy the previous round's values
        with torch.cuda.stream(b):
            K.sync_wait(p(4), p(2), p(1), True)
            dst[r].copy_(src)
        with torch.cuda.stream(a):
            for _ in range(8):
                src.add_(1.0)          # several dependent kernels ahead of the signal
            K.sync_signal(p(4), p(0), True)
            # ... and the producer must not run ahead into the next round before the consumer has read: the reverse hand-off
            K.sync_wait(p(5), p(0), p(1), False)
        with torch.cuda.stream(b):
            K.sync_signal(p(5), p(2), False)
    torch.cuda.synchronize()
    assert int(w[1]) == 0, "a device-side wait timed out"
    for r in range(4):
        assert float(dst[r].min()) == float(dst[r].max()) == 8.0 * (r + 1), r
    assert int(w[0]) == 5 and int(w[2]) == 5


def test_sync_wait_is_bounded():
    """a wait whose flag never arrives gives up after max_polls tries, counts a time-out and lets its stream go on"""
    from nas_3d_unet_amd import kernels as K
    w = _words()
    p = lambda i: w.data_ptr() + 4 * i
    out = torch.zeros(8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        K.sync_wait(p(4), p(0), p(1), False, max_polls=2000)
        out.fill_(3.0)
    torch.cuda.synchronize()
    assert int(w[1]) == 1 and float(out.sum()) == 24.0
    with pytest.raises(Exception):
        K.sync_wait(p(4), p(0), p(1), False, max_polls=0)


def test_sync_wait2_waits_for_both_flags():
    """n3d_sync_wait2: the stream goes on only when BOTH flags have reached its step; bounded like the single wait"""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.train import reserve_side_streams
    w = _words()
    p = lambda i: w.data_ptr() + 4 * i
    a, b, c = reserve_side_streams(torch.device("cuda", torch.cuda.current_device()), 3)[:3]
    w[3] = 1     # the third stream's step word
    src0, src1 = torch.zeros(1 << 22, device="cuda"), torch.zeros(1 << 22, device="cuda")
    dst = torch.zeros(2, 1 << 22, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(c):            # the consumer first: it must hold until both producers have signalled
        K.sync_wait2(p(4), p(5), p(2), p(1), True)
        dst[0].copy_(src0)
        dst[1].copy_(src1)
    with torch.cuda.stream(a):
        for _ in range(6):
            src0.add_(1.0)
        K.sync_signal(p(4), p(0), True)
    with torch.cuda.stream(b):
        for _ in range(9):
            src1.add_(1.0)
        K.sync_signal(p(5), p(3), True)
    torch.cuda.synchronize()
    assert int(w[1]) == 0, "the two-flag wait timed out"
    assert float(dst[0].min()) == float(dst[0].max()) == 6.0 and float(dst[1].min()) == float(dst[1].max()) == 9.0
    # one flag missing: a time-out is counted, the stream goes on
    w2 = _words()
    q = lambda i: w2.data_ptr() + 4 * i
    out = torch.zeros(8, device="cuda")
    with torch.cuda.stream(c):
        K.sync_signal(q(4), q(0), False)
        K.sync_wait2(q(4), q(5), q(0), q(1), False, max_polls=2000)
        out.fill_(2.0)
    torch.cuda.synchronize()
    assert int(w2[1]) == 1 and float(out.sum()) == 16.0


def _batch(seed, size=32, batch=2):
    rng = np.random.default_rng(seed)
    x = dev(rng.standard_normal((batch, 4, size, size, size)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (batch, 3, size, size, size)) < 0.3).astype(np.float32))
    return x, t


@pytest.mark.parametrize("gname", ["G_CONV", "G_ALL"])
def test_side_schedule_gradients_equal_plain(gname):
    from nas_3d_unet_amd.train import Trainer
    x, t = _batch(41)
    net, _ = build_net("searched", gname, 4)
    plain = Trainer(net, graph=False, side_wgrad=False)
    plain._fwd_bwd(x, t)
    ref = plain.fp.grad.clone()
    tot = float(ref.double().norm())
    net2, _ = build_net("searched", gname, 4)
    tr = Trainer(net2, graph=False, side_wgrad=True)
    assert tr.side is not None and tr._side_ok(), "the side stream was not accepted on this box"
    for _ in range(3):        # first pass records the pack jobs; later ones run pre-packed
        tr.fp.grad.zero_()
        tr._side_step_eager(x, t)
        torch.cuda.synchronize()
        tr.check_sync()
        assert int((tr.side.sync[8:8 + tr.side.JOIN] > 0).sum()) >= 6, "no cut points were placed"
        assert float((tr.fp.grad - ref).double().norm()) <= 1e-5 * tot
    # the C in {4, 8} levels run the SAME weight-gradient kernels on the same operands: bit-identical there
    names = [n for n, _ in net2.named_parameters()]
    for n, q, off in zip(names, tr.fp.params, tr.fp.offsets):
        if n.startswith("up_cells.4._ops") and n.endswith("conv.weight"):
            assert torch.equal(tr.fp.grad[off:off + q.numel()], ref[off:off + q.numel()]), n


def test_side_schedule_graph_replay_matches_plain_trainer():
    from nas_3d_unet_amd.train import Trainer
    x, t = _batch(43)
    out = []
    for side in (False, "force"):
        net, _ = build_net("searched", "G_CONV", 4)
        tr = Trainer(net, graph=True, side_wgrad=side)
        losses = [float(tr.step(x, t)) for _ in range(4)]
        torch.cuda.synchronize()
        tr.check_sync()
        if side:
            assert tr._use_side and tr._side_graphs is not None
        out.append((losses, tr.fp.flat.clone()))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=0, atol=2e-5)
    # Adam moves an element whose gradient is fp32 noise by ~lr per step either way, and the two schedules differ by an fp32 rounding
    # of the preprocess epilogue backward: on this net a 1e-6 nudge of 1 % of the weights after step 1 grows to 8e-4 of the norm by
    # step 4 (tools/chaos_probe.py, profiles/r04_chaos_probe.log).  A weight gradient that went missing moves a whole layer by 4 lr: ~1e-2 of the norm.
    d = (out[0][1] - out[1][1]).abs()
    assert float(d.double().norm()) <= 2e-3 * float(out[0][1].double().norm())


@pytest.mark.parametrize("size", [32, 64])
def test_side_schedule_is_reproducible_run_to_run(size):
    """three streams, ~60 device-side hand-offs per step, gradient buffers that the side stream accumulates into behind deferred
    joins: two trainers built alike must end up with bit-identical weights (a missing join would show as run-to-run noise)"""
    from nas_3d_unet_amd.train import Trainer
    x, t = _batch(47, size=size)
    flats = []
    for _ in range(2):
        torch.manual_seed(5)
        net, _ = build_net("searched", "G_CONV", 4)
        tr = Trainer(net, graph=True, side_wgrad="force")
        for _ in range(5):
            tr.step(x, t)
        torch.cuda.synchronize()
        tr.check_sync()
        assert tr._use_side
        flats.append(tr.fp.flat.clone())
    assert torch.equal(flats[0], flats[1])


def test_one_slow_sample_does_not_retire_the_side_schedule():
    """the replay monitor (every 256th step bracketed with events): a single slow sample -- the bracket also holds whatever the host did
    between the graph launches -- is measured again; only three slow samples in a row drop the side schedule for the plain graph"""
    import warnings
    from nas_3d_unet_amd.train import Trainer
    x, t = _batch(53)
    net, _ = build_net("searched", "G_CONV", 4)
    tr = Trainer(net, graph=True, side_wgrad="force")
    for _ in range(3):
        tr.step(x, t)
    assert tr._use_side
    real = tr.schedule_times
    tr.schedule_times = (1e-9, 1e-9)      # every sample now counts as slow against this "plain graph time"
    tr._n_steps = 255                  # the next replay is a sampled one
    tr.step(x, t)                          # sampled (slow sample 1) ...
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        tr.step(x, t)                      # ... evaluated here: slow -> measured again, nothing dropped
        assert tr._use_side and tr.side.recheck == 1
        tr.schedule_times = real
        tr.step(x, t)                      # the re-measurement (an honest sample) ...
        tr.step(x, t)                      # ... evaluated: fine again
    assert tr._use_side and tr.side.recheck == 0 and tr.side.slow_run == 0
    # three slow samples in a row do retire it
    tr.schedule_times = (1e-9, 1e-9)
    tr._n_steps = 255
    with pytest.warns(UserWarning, match="degraded"):
        for _ in range(7):
            tr.step(x, t)
    assert not tr._use_side
    l = float(tr.step(x, t))               # the plain graph takes over
    assert np.isfinite(l)
    tr.check_sync()


def test_schedule_choice_leaves_the_state_alone():
    """the default trainer times both captured schedules on the real step at capture time: weights, Adam moments, step counters
    and the Dropout3d generator must come out of that exactly as they went in"""
    from nas_3d_unet_amd import programs as P
    from nas_3d_unet_amd.train import Trainer
    x, t = _batch(47)
    res = []
    for side in (False, None):
        torch.manual_seed(5)
        net, head = build_net("searched", "G_CONV", 4, keep_dropout=True)
        net.train()
        P.dropout_state(head[0].dropout, torch.device("cuda", torch.cuda.current_device()), seed=1234)   # both nets: the same masks
        tr = Trainer(net, graph=True, side_wgrad=side)
        losses = [float(tr.step(x, t)) for _ in range(3)]
        res.append((losses, tr.fp.flat.clone(), int(tr.fp.step), tr))
    tr = res[1][3]
    assert tr.schedule_times is not None and min(tr.schedule_times) > 0
    assert res[0][2] == res[1][2] == 3
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=2e-5)   # same dropout masks, same trajectory
    d = (res[0][1] - res[1][1]).abs()
    assert float(d.double().norm()) <= 2e-3 * float(res[0][1].double().norm())     # (see the replay test above for the bound)


def test_search_weight_pass_on_the_side_stream():
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(53)
    mk = lambda: (dev(rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32)),
                  dev((rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32)))
    (x, t), (vx, vt) = mk(), mk()
    res = []
    for side, graph in ((False, False), (True, False), ("force", True)):
        net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
        fill_module(net)
        net.kernel.last_conv[0].dropout = None
        net = net.cuda()
        tr = SearchTrainer(net, graph=graph, side_wgrad=side)
        if side:
            assert tr.side is not None
        got = [tuple(float(v) for v in tr.step(x, t, vx, vt)) for _ in range(2)]
        torch.cuda.synchronize()
        tr.check_sync()
        res.append((got, tr.fp.grad.clone(), tr.agrad.clone()))
    tot, atot = float(res[0][1].double().norm()), float(res[0][2].double().norm())
    for got, g, ag in res[1:]:
        np.testing.assert_allclose(np.array(got), np.array(res[0][0]), rtol=0, atol=2e-5)
        # gradients of the second step (the buffers hold the last pass): the first steps' Adam updates may differ by noise-level signs
        assert float((g - res[0][1]).double().norm()) <= 2e-3 * tot
        assert float((ag - res[0][2]).double().norm()) <= 2e-3 * atot


def test_supernet_forward_on_two_streams_is_bit_identical():
    """fused._run_forward_side: the off-chain edges of every supernet cell run their weight ops on the side stream, the epilogues stay
    on the main stream in the usual order -- same kernels on the same operands, so probabilities, loss and every gradient must equal
    the single-stream forward bit for bit (cell.py:76-82 semantics are untouched)."""
    from nas_3d_unet_amd import kernels as K, loss, nas
    from nas_3d_unet_amd.train import SideSchedule
    cfg = orc.DEFAULT_CFG._replace(depth=3)
    rng = np.random.default_rng(61)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net)
    net.kernel.last_conv[0].dropout = None
    net = net.cuda()
    with torch.no_grad():                      # alphas away from the uniform point: every MixedOp weight differs
        for a in net.alphas():
            a.copy_(torch.from_numpy(rng.standard_normal(tuple(a.shape)).astype(np.float32)).cuda())
    out = []
    for two_streams in (False, True, True):
        for p in net.parameters():
            p.grad = None
        if two_streams:
            sd = SideSchedule(torch.device("cuda", torch.cuda.current_device()), K.StepContext(torch.device("cuda")))
            assert sd.stream is not None, "the side stream was not accepted on this box"
            with sd.forward_mode():
                p_ = net(x)
            l = loss.WeightedDiceLoss()(p_, t)
            l.backward()
            sd.launch_side(redirect=True)
            sd.finish()
            torch.cuda.synchronize()
            sd.check()
            assert int((sd.sync[8:8 + sd.JOIN] > 0).sum()) >= 20, "no hand-offs were placed"
        else:
            p_ = net(x)
            l = loss.WeightedDiceLoss()(p_, t)
            l.backward()
        out.append((p_.detach().clone(), float(l.detach()), {n: q.grad.clone() for n, q in net.named_parameters() if q.grad is not None}))
    for got in out[1:]:
        assert torch.equal(got[0], out[0][0])
        assert got[1] == out[0][1]
        assert got[2].keys() == out[0][2].keys()
        for n in got[2]:
            assert torch.equal(got[2][n], out[0][2][n]), n


# ---- a timed-out hand-off must never reach the weights (n3d_adam_step_guarded; train.py:121-128: a step is forward, backward,
# update -- there is no "update from garbage") ---------------------------------------------------------------------------------
def test_guarded_adam_withholds_the_update():
    """the kernel-level contract: time-outs != acknowledged (or a peer's flag) -> parameters, moments and the step counter are
    bit-unchanged, the loss reads NaN, the host word is set; equal counters -> the ordinary update, bit-identical to n3d_adam_step"""
    from nas_3d_unet_amd import kernels as K
    n = 10000
    g = torch.randn(n, device="cuda")
    mk = lambda: (torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), K.step_counter("cuda"))
    words = torch.zeros(4, dtype=torch.int32, device="cuda")     # [0] time-outs, [1] acknowledged
    flag = torch.zeros(1, device="cuda")
    hw = K.HostWord()
    p0, m0, v0, st0 = mk()
    K.adam_step(p0, g, m0, v0, st0)                               # reference: the unguarded update
    for bad_words, bad_flag in ((False, False), (True, False), (False, True)):
        p, m, v, st = mk()
        loss = torch.full((), 0.5, device="cuda")
        words[0], words[1] = (3 if bad_words else 2), 2
        flag[0] = 1.0 if bad_flag else 0.0
        hw.clear()
        guard = K.UpdateGuard(words.data_ptr(), words.data_ptr() + 4, flag.data_ptr(), loss.data_ptr(), hw.ptr)
        K.adam_step(p, g, m, v, st, guard=guard)
        torch.cuda.synchronize()
        if bad_words or bad_flag:
            assert torch.equal(p, torch.ones_like(p)) and float(m.abs().max()) == 0.0 and float(v.abs().max()) == 0.0
            assert int(st) == 0 and int(st._base[1] if st._base is not None else 0) == 0
            assert torch.isnan(loss) and hw.value == 1
        else:
            assert torch.equal(p, p0) and torch.equal(m, m0) and torch.equal(v, v0) and int(st) == 1
            assert float(loss) == 0.5 and hw.value == 0


@pytest.mark.parametrize("graph", [True, False])
def test_timed_out_handoff_never_reaches_the_weights(graph):
    """two good steps, then a time-out is counted on the device (what n3d_sync_wait does when it gives up) in the middle of the
    run: the step's update is withheld (weights, moments, step counter bit-unchanged), its loss is NaN, the NEXT step() raises --
    and keeps raising -- until recover(), after which training goes on (on one stream)."""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.train import Trainer
    x, t = _batch(59)
    net, _ = build_net("searched", "G_CONV", 4)
    tr = Trainer(net, graph=graph, side_wgrad="force" if graph else True)
    assert tr.side is not None
    for _ in range(2):
        l = tr.step(x, t)
    torch.cuda.synchronize()
    assert np.isfinite(float(l)) and int(tr.fp.step) == 2 and tr.sync_timeouts() == 0
    w, m, v = tr.fp.flat.clone(), tr.fp.exp_avg.clone(), tr.fp.exp_avg_sq.clone()
    tr.side.sync[1] += 1                       # a wait gave up
    l = tr.step(x, t)
    torch.cuda.synchronize()
    assert torch.isnan(l), "the loss of a withheld step must read NaN"
    assert torch.equal(tr.fp.flat, w) and torch.equal(tr.fp.exp_avg, m) and torch.equal(tr.fp.exp_avg_sq, v) and int(tr.fp.step) == 2
    for _ in range(2):
        with pytest.raises(K.N3DError, match="timed out"):
            tr.step(x, t)
    with pytest.raises(K.N3DError, match="timed out"):
        tr.check_sync()
    assert torch.equal(tr.fp.flat, w) and int(tr.fp.step) == 2
    tr.recover()
    l = tr.step(x, t)
    torch.cuda.synchronize()
    tr.check_sync()
    assert np.isfinite(float(l)) and int(tr.fp.step) == 3 and not torch.equal(tr.fp.flat, w)
    assert not tr._use_side


def test_a_real_timed_out_wait_is_caught():
    """the side graph is launched first and every wait of this capture gives up after ONE poll (a side wait far ahead of its
    signal): real time-outs, counted by n3d_sync_wait itself.  No update may happen, from the first replay on."""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.train import Trainer
    x, t = _batch(61)
    net, _ = build_net("searched", "G_CONV", 4)
    tr = Trainer(net, graph=True, side_wgrad="force")
    assert tr.side is not None
    w = tr.fp.flat.clone()
    K.SYNC_MAX_POLLS[0] = 1
    try:
        l = tr.step(x, t)                      # warm-up passes + capture + first replay
        torch.cuda.synchronize()
    finally:
        K.SYNC_MAX_POLLS[0] = None
    assert tr.sync_timeouts() > 0, "no wait timed out: the injection did not work"
    assert torch.isnan(l) and torch.equal(tr.fp.flat, w) and int(tr.fp.step) == 0
    with pytest.raises(K.N3DError, match="timed out"):
        tr.step(x, t)
    tr.recover()
    for _ in range(2):
        l = tr.step(x, t)
    torch.cuda.synchronize()
    tr.check_sync()
    assert np.isfinite(float(l)) and int(tr.fp.step) == 2 and not torch.equal(tr.fp.flat, w)


def test_search_trainer_withholds_both_updates():
    from nas_3d_unet_amd import kernels as K, nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(67)
    mk = lambda: (dev(rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32)),
                  dev((rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32)))
    (x, t), (vx, vt) = mk(), mk()
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net)
    net.kernel.last_conv[0].dropout = None
    tr = SearchTrainer(net.cuda(), graph=True, side_wgrad="force")
    assert tr.side is not None
    tr.step(x, t, vx, vt)
    torch.cuda.synchronize()
    w, a = tr.fp.flat.clone(), tr.aflat.clone()
    tr.side.sync[1] += 1
    la, lw = tr.step(x, t, vx, vt)
    torch.cuda.synchronize()
    assert torch.isnan(la) and torch.isnan(lw)
    assert torch.equal(tr.fp.flat, w) and torch.equal(tr.aflat, a) and int(tr.fp.step) == 1 and int(tr.a_step) == 1
    with pytest.raises(K.N3DError, match="timed out"):
        tr.step(x, t, vx, vt)
    tr.recover()
    la, lw = tr.step(x, t, vx, vt)
    torch.cuda.synchronize()
    tr.check_sync()
    assert np.isfinite(float(la)) and np.isfinite(float(lw)) and int(tr.fp.step) == 2 and not torch.equal(tr.aflat, a)
