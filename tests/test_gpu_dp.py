"""GPU: the data-parallel code paths in a 1-rank RCCL group on one GPU (what N > 1 runs, minus the peers): single-bucket and
bucketed-overlap schedules of Trainer (graph segments split inside the backward walk), the C ABI's n3d_comm_* wrapper, and
SearchTrainer's two exchanges -- each against the plain single-GPU trainer on the same inputs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from test_gpu_nets import build_net
from _util import dev, fill_module
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["N3D_FORCE_DP"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()
    os.environ.pop("N3D_FORCE_DP", None)


def _losses_and_weights(tr, x, t, n=3):
    losses = [float(tr.step(x, t)) for _ in range(n)]
    return losses, tr.fp.flat.clone()


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("buckets,comm", [(1, "torch"), (2, "torch"), (3, "torch"), (2, "rccl"), (1, "rccl")])
def test_trainer_dp_schedules_match_single_gpu(one_rank_group, graph, buckets, comm):
    from nas_3d_unet_amd.train import Trainer
    rng = np.random.default_rng(41)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    os.environ.pop("N3D_FORCE_DP")
    try:
        net, _ = build_net("searched", "G_CONV", 4)
        # the bucketed exchange replays the single-stream schedule (graph segments), everything else the side-stream one: the
        # reference trainer is given the same schedule, so the comparison is bit for bit (the two schedules differ by an fp32
        # rounding of the preprocess epilogue backward, and Adam turns 1e-6 on a weight into 1e-3 within three steps:
        # tools/chaos_probe.py, profiles/r04_chaos_probe.log; tests/test_gpu_side.py compares the schedules themselves)
        # (graph, one bucket: both trainers are PINNED to the side schedule -- left to themselves each would pick by its own
        # wall-clock timing, and the comparison below is bit for bit)
        sched = "force" if graph else (True if buckets > 1 else None)
        ref = Trainer(net, graph=graph, side_wgrad=sched)
        assert not ref.dp_path
        lr_, wr = _losses_and_weights(ref, x, t)
    finally:
        os.environ["N3D_FORCE_DP"] = "1"
    net, _ = build_net("searched", "G_CONV", 4)
    tr = Trainer(net, graph=graph, n_buckets=buckets, comm=comm, side_wgrad=sched)
    assert tr.dp_path and len(tr.sync.ranges) == buckets and tr.sync.backend == comm
    # "rccl": the process's shared n3d_comm communicator; "torch": torch.distributed on a comm stream that is never captured
    assert (tr.sync._comm is not None) if comm == "rccl" else (tr.sync._torch_stream is not None)
    l, w = _losses_and_weights(tr, x, t)
    assert ref._use_side == tr._use_side or buckets > 1
    if buckets > 1:
        # the bucketed exchange rides on the side-stream schedule: bucket j's all-reduce on the weight-gradient stream, tied to the chain
        # by device flags (graph: the stream's graph comes in `buckets` segments with the all-reduces launched between them)
        assert tr.side is not None and tr._buckets is not None
        if graph:
            assert tr._use_side and tr._side_wsegs == buckets
    assert l == lr_ and torch.equal(w, wr)


def test_default_exchange_is_the_shared_rccl_communicator(one_rank_group):
    """data-parallel trainers exchange through n3d_comm_* on ONE communicator per process group (comm.for_group), also the search
    trainer's two exchanges -- no torch.distributed collective in the step, so ProcessGroupNCCL's watchdog has nothing to poll
    while a stream is being captured (comm.py; profiles/r05_dp_capture_loop.log)"""
    from nas_3d_unet_amd import comm as C, nas
    from nas_3d_unet_amd.train import SearchTrainer, Trainer
    net, _ = build_net("searched", "G_CONV", 2)
    a = Trainer(net, graph=False)
    net2, _ = build_net("searched", "G_CONV", 2)
    b = Trainer(net2, graph=False, n_buckets=2)
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    shell = fill_module(nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)).cuda()
    c = SearchTrainer(shell, graph=False)
    comms = [a.sync._comm, b.sync._comm, c.sync_alpha._comm, c.sync_kernel._comm]
    assert all(x is not None and x is comms[0] for x in comms) and comms[0] is C.for_group(None, torch.device("cuda", 0))
    assert all(s.backend == "rccl" and s._torch_stream is None for s in (a.sync, b.sync, c.sync_alpha, c.sync_kernel))
    # broadcast / all_true on a 1-rank group: identities
    w = torch.arange(8, dtype=torch.float32, device="cuda")
    comms[0].broadcast(w)
    torch.cuda.synchronize()
    assert torch.equal(w.cpu(), torch.arange(8, dtype=torch.float32))
    assert comms[0].all_true(True) and not comms[0].all_true(False)


@pytest.mark.parametrize("buckets", [2, 3])
def test_bucketed_exchange_without_a_side_stream_is_one_bucket(one_rank_group, buckets):
    """side_wgrad=False: no device-flag schedule, nothing for a bucket to overlap with -- the exchange stays ONE bucket behind the
    step's graph (round 5: the event-tied one-graph-per-bucket fallback of round 2 is gone); same weights as the plain
    single-stream trainer, bit for bit"""
    from nas_3d_unet_amd.train import Trainer
    rng = np.random.default_rng(41)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    os.environ.pop("N3D_FORCE_DP")
    try:
        net, _ = build_net("searched", "G_CONV", 4)
        ref = Trainer(net, graph=True, side_wgrad=False)
        lr_, wr = _losses_and_weights(ref, x, t)
    finally:
        os.environ["N3D_FORCE_DP"] = "1"
    net, _ = build_net("searched", "G_CONV", 4)
    tr = Trainer(net, graph=True, n_buckets=buckets, side_wgrad=False)
    l, w = _losses_and_weights(tr, x, t)
    assert tr.dp_path and tr._buckets is None and len(tr.sync.ranges) == 1 and tr._graph is not None
    assert l == lr_ and torch.equal(w, wr)


def test_replay_monitor_retires_the_side_schedule_collectively(one_rank_group):
    """the 256-step re-timing of a replayed side schedule decides by an all-reduce ("my sample was fast" on every
    rank), three slow samples in a row retire it -- on every rank in the same step.  Here: a 1-rank group whose plain-schedule time
    is set absurdly low, so that every sample counts as slow."""
    import warnings
    from nas_3d_unet_amd.train import Trainer
    rng = np.random.default_rng(47)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    net, _ = build_net("searched", "G_CONV", 4)
    tr = Trainer(net, graph=True, side_wgrad="force")
    tr.step(x, t)
    assert tr.dp_path and tr._use_side
    calls = []
    real = tr.sync.all_true
    tr.sync.all_true = lambda flag: (calls.append(bool(flag)), real(flag))[1]
    tr.schedule_times = (1e-7, 1e-7)
    tr._n_steps = 254
    with warnings.catch_warnings(record=True) as wn:
        warnings.simplefilter("always")
        for _ in range(10):
            tr.step(x, t)
    assert calls[:3] == [False, False, False] and not tr._use_side and any("degraded" in str(w.message) for w in wn)
    tr.step(x, t)        # "force" had no plain graph: this step captured one and trains on
    torch.cuda.synchronize()
    assert tr._graph is not None and tr.sync_timeouts() == 0


@pytest.mark.parametrize("buckets", [1, 2])
def test_dp_peer_flag_withholds_the_update_on_every_rank(one_rank_group, buckets):
    """data parallel: the "a hand-off of this rank timed out" word rides in front of the gradients through the SUM all-reduce, and
    the guarded Adam of EVERY rank tests the sum -- here a 1-rank RCCL group: the flag written before the exchange must still
    withhold the update after it (weights bit-unchanged, loss NaN, next step raises, recover() continues)"""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.train import Trainer
    rng = np.random.default_rng(45)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    net, _ = build_net("searched", "G_CONV", 4)
    tr = Trainer(net, graph=True, side_wgrad="force", n_buckets=buckets)
    assert tr.dp_path and tr.side is not None and len(tr.sync.ranges) == buckets
    tr.step(x, t)
    torch.cuda.synchronize()
    assert float(tr.fp.grad_full[0]) == 0.0 and int(tr.fp.step) == 1
    w = tr.fp.flat.clone()
    tr.side.sync[1] += 1
    l = tr.step(x, t)
    torch.cuda.synchronize()
    assert float(tr.fp.grad_full[0]) == 1.0, "the flag did not travel with the gradients"
    assert torch.isnan(l) and torch.equal(tr.fp.flat, w) and int(tr.fp.step) == 1
    with pytest.raises(K.N3DError, match="timed out"):
        tr.step(x, t)
    tr.recover()       # (buckets > 1: the exchange continues as event-tied graph segments, which refresh the flag themselves)
    for k in range(2):
        l = tr.step(x, t)
        torch.cuda.synchronize()
        assert np.isfinite(float(l)) and int(tr.fp.step) == 2 + k and float(tr.fp.grad_full[0]) == 0.0


def test_search_trainer_dp_matches_single_gpu(one_rank_group):
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(43)
    mk = lambda: (dev(rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32)), dev((rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32)))
    (x, t), (vx, vt) = mk(), mk()
    out = []
    for dp in (False, True):
        if not dp:
            os.environ.pop("N3D_FORCE_DP")
        try:
            net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
            fill_module(net)
            net.kernel.last_conv[0].dropout = None
            tr = SearchTrainer(net.cuda(), graph=True)
            assert tr.dp_path == dp
            ls = [tuple(float(v) for v in tr.step(x, t, vx, vt)) for _ in range(3)]
            out.append((ls, tr.fp.flat.clone(), tr.aflat.clone()))
        finally:
            os.environ["N3D_FORCE_DP"] = "1"
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=0, atol=2e-6)
    assert float((out[0][1] - out[1][1]).abs().max()) <= 2e-6 and float((out[0][2] - out[1][2]).abs().max()) <= 2e-6


def test_comm_wrapper_allreduce_one_rank():
    """n3d_comm_* through the C ABI alone (no torch.distributed): unique id -> init -> in-place SUM all-reduce -> destroy"""
    import ctypes as C
    from nas_3d_unet_amd import _lib
    lib = _lib.load()
    assert lib.n3d_comm_available() == 1
    idb = (C.c_char * 128)()
    _lib.check(lib.n3d_comm_unique_id(idb), "unique_id")
    comm = C.c_void_p()
    _lib.check(lib.n3d_comm_init(idb, 1, 0, C.byref(comm)), "init")
    g = torch.arange(1000, dtype=torch.float32, device="cuda")
    _lib.check(lib.n3d_comm_allreduce_sum(comm, C.c_void_p(g.data_ptr()), g.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "allreduce")
    torch.cuda.synchronize()
    assert torch.equal(g.cpu(), torch.arange(1000, dtype=torch.float32))
    _lib.check(lib.n3d_comm_destroy(comm), "destroy")


def _two_rank_worker(rank, world, port, out, graph, buckets):
    """one of two processes sharing cuda:0; gloo carries the CUDA gradient buffer (RCCL refuses two ranks on one device)"""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.pop("N3D_FORCE_DP", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nas_3d_unet_amd.train import Trainer
    from test_gpu_nets import build_net as bn
    rng = np.random.default_rng(77)
    xs = rng.standard_normal((2 * world, 4, 16, 16, 16)).astype(np.float32)
    ts = (rng.uniform(0, 1, (2 * world, 3, 16, 16, 16)) < 0.3).astype(np.float32)
    net, _ = bn("searched", "G_CONV", 2)
    if rank == 1:   # rank 0's weights must win
        with torch.no_grad():
            for q in net.parameters():
                q.add_(0.25)
    tr = Trainer(net, graph=graph, n_buckets=buckets)
    assert tr.dp_path and tr.sync.world == world and len(tr.sync.ranges) == buckets
    x, t = dev(xs[2 * rank:2 * rank + 2]), dev(ts[2 * rank:2 * rank + 2])
    losses = [float(tr.step(x, t)) for _ in range(3)]
    torch.save({"losses": losses, "flat": tr.fp.flat.detach().cpu()}, out + ".r%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graph,buckets", [(False, 2), (True, 1), (True, 2)])
def test_two_processes_on_one_gpu_equal_the_global_batch(tmp_path, graph, buckets):
    """world_size 2 for real (two processes, each running the HIP trainer on its shard, gradients exchanged through GradSync):
    single bucket after the step, or buckets exchanged while the backward walk continues (eager, and as HIP-graph segments);
    three Adam steps give both ranks identical weights, equal to one process training on the concatenated batch (GroupNorm is
    per sample and the Dice loss a mean over (b, c) rows, so the mean of the shard gradients is the global-batch gradient)."""
    import torch.multiprocessing as mp
    from nas_3d_unet_amd.train import Trainer
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "dp2")
    mp.spawn(_two_rank_worker, args=(2, port, out, graph, buckets), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".r0"), torch.load(out + ".r1")
    assert torch.equal(r0["flat"], r1["flat"])
    # one process, global batch of 4
    forced = os.environ.pop("N3D_FORCE_DP", None)
    try:
        rng = np.random.default_rng(77)
        xs = rng.standard_normal((4, 4, 16, 16, 16)).astype(np.float32)
        ts = (rng.uniform(0, 1, (4, 3, 16, 16, 16)) < 0.3).astype(np.float32)
        net, _ = build_net("searched", "G_CONV", 2)
        # the bucketed exchange replays the single-stream schedule (graph segments), everything else the side-stream one: the
        # reference trainer is given the same schedule, so the comparison is bit for bit (the two schedules differ by an fp32
        # rounding of the preprocess epilogue backward, and Adam turns 1e-6 on a weight into 1e-3 within three steps:
        # tools/chaos_probe.py, profiles/r04_chaos_probe.log; tests/test_gpu_side.py compares the schedules themselves)
        # (graph, one bucket: both trainers are PINNED to the side schedule -- left to themselves each would pick by its own
        # wall-clock timing, and the comparison below is bit for bit)
        sched = "force" if graph else (True if buckets > 1 else None)
        ref = Trainer(net, graph=graph, side_wgrad=sched)
        lr_ = [float(ref.step(dev(xs), dev(ts))) for _ in range(3)]
        wr = ref.fp.flat.detach().cpu()
    finally:
        if forced is not None:
            os.environ["N3D_FORCE_DP"] = forced
    # the global loss is the mean of the two shard losses
    np.testing.assert_allclose([(a + b) / 2 for a, b in zip(r0["losses"], r1["losses"])], lr_, rtol=0, atol=5e-6)
    scale = float(wr.abs().max())
    assert float((r0["flat"] - wr).abs().max()) <= 2e-5 * scale, float((r0["flat"] - wr).abs().max())


def test_bench_on_two_gpus_when_the_node_has_them():
    """`python bench.py --gpus 2` starts its own ranks (one process per GPU, RCCL over xGMI) and prints ONE JSON line on rank 0.  Skipped
    on a single-GPU box (this pool); the day a multi-GPU node runs the suite it exercises the real N > 1 path, bucketed exchange
    included."""
    import json
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT="29533")
    env.pop("N3D_FORCE_DP", None)
    for extra in ([], ["--buckets", "2"]):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-cpu-baseline",
                              "--no-other-configs", "--no-kernel-table", "--no-roofline"] + extra, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["sync_timeouts"] == 0 and d["value"] > 0
        assert d["config"]["dp_buckets"] == (2 if extra else 1)
