"""Data-parallel path on CPU with gloo, world_size 2 (the GPU run uses the same GradSync over RCCL).

Property: GroupNorm is per sample and the Dice loss is a mean over (b, c) rows, so with equal
shards the mean of the per-rank gradients equals the single-process gradient of the concatenated batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nas_3d_unet_amd.train import GradSync, flatten_params
    from oracle import ref_path as orc
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    P = orc.make_params(orc.searched_param_specs(cfg, orc.G_ALL), requires_grad=True, salt=rank * 0)
    params = list(P.values())
    flat, grad, _ = flatten_params(params)
    # rank 0's weights win (broadcast), like Trainer.__init__
    if rank == 1:
        with torch.no_grad():
            flat.add_(1.0)
    dist.broadcast(flat, src=0)
    # after the broadcast every rank holds rank 0's weights (= the closed-form fill)
    fresh = orc.make_params(orc.searched_param_specs(cfg, orc.G_ALL))
    bcast_err = max(float((p.detach() - q).abs().max()) for p, q in zip(params, fresh.values()))
    torch.save({"bcast": bcast_err}, out + ".bcast%d" % rank)
    rng = np.random.default_rng(5)
    xs = rng.standard_normal((2 * world, 4, 16, 16, 16)).astype(np.float32)
    ts = (rng.uniform(0, 1, (2 * world, 3, 16, 16, 16)) < 0.3).astype(np.float32)
    x, t = torch.from_numpy(xs[2 * rank:2 * rank + 2]), torch.from_numpy(ts[2 * rank:2 * rank + 2])
    loss = orc.dice_loss(orc.searched_forward(P, x, orc.G_ALL, cfg), t)
    loss.backward()
    sync = GradSync(grad, None, n_buckets=3)
    assert len(sync.edges) == 4 and sync.edges[-1] == grad.numel()
    sync.all_reduce()
    grad.div_(world)
    if rank == 0:
        # single-process reference on the concatenated batch
        Q = orc.make_params(orc.searched_param_specs(cfg, orc.G_ALL), requires_grad=True)
        l2 = orc.dice_loss(orc.searched_forward(Q, torch.from_numpy(xs), orc.G_ALL, cfg), torch.from_numpy(ts))
        l2.backward()
        tot = float(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in Q.values())))
        worst = 0.0
        for (n, q), p in zip(Q.items(), params):
            worst = max(worst, float((q.grad.double() - p.grad.double()).norm()) / tot)
        torch.save({"worst": worst}, out)
    dist.barrier()
    dist.destroy_process_group()


def _bucket_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nas_3d_unet_amd import searched
    from nas_3d_unet_amd.train import GradSync, Trainer
    from oracle import ref_path as orc
    torch.manual_seed(100 + rank)          # different initial weights per rank: the constructor must broadcast rank 0's
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(*orc.G_CONV))
    res = {}
    for nb in (2, 3):
        tr = Trainer(net, graph=False, n_buckets=nb) if nb == 2 else Trainer(searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(*orc.G_CONV)),
                                                                             graph=False, n_buckets=nb)
        # (the bucketed exchange itself rides on the side-stream schedule, which a CPU trainer does not have: the exchange stays one
        # bucket there; the plan and the bucket-by-bucket exchange are host logic and are what this test covers)
        assert tr._buckets is None and len(tr.sync.ranges) == 1
        plan = tr._bucket_plan()
        assert plan is not None and len(plan) == nb, plan
        sync = GradSync(tr.fp.grad, None, ranges=[r for _, r in plan], header=tr.fp.grad_full)
        # issue order = backward completion order: tail of the flat buffer first, contiguous, disjoint, covering everything
        assert plan[0][1][1] == tr.fp.numel and plan[-1][1][0] == 0 and plan[-1][0] == -1
        for (k0, (a0, b0)), (k1, (a1, b1)) in zip(plan[:-1], plan[1:]):
            assert a0 == b1 and a0 < b0 and (k1 == -1 or k1 < k0)
        # the cell that closes a bucket owns the bucket's first parameter
        cells = list(tr.model.down_cells) + list(tr.model.up_cells)
        first = {o: p for p, o in zip(tr.fp.params, tr.fp.offsets)}
        for k, (a, b) in plan[:-1]:
            assert any(first[a] is q for q in cells[k].parameters())
        # the exchange, bucket by bucket in issue order, is the all-reduce of the whole buffer
        tr.fp.grad.copy_(torch.arange(tr.fp.numel, dtype=torch.float32) * (rank + 1) * 1e-3)
        for j in range(len(plan)):
            sync.reduce_range(j)
        want = torch.arange(tr.fp.numel, dtype=torch.float32) * 1e-3 * sum(r + 1 for r in range(world))
        res["nb%d" % nb] = float((tr.fp.grad - want).abs().max() / want.abs().max())
        res["frac%d" % nb] = [(b - a) / tr.fp.numel for _, (a, b) in plan]
    # the weights every rank ends up with are rank 0's
    w = tr.fp.flat.clone()
    dist.broadcast(w, src=0)
    res["bcast"] = float((w - tr.fp.flat).abs().max())
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_bucket_plan_and_bucketed_exchange(tmp_path):
    """the bucketed-overlap schedule of train.Trainer (n_buckets >= 2): bucket ranges follow the backward completion order
    (head and late cells first) and exchanging them one by one over gloo equals one all-reduce of the flat buffer"""
    out = str(tmp_path / "res.pt")
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["nb2"] < 1e-6 and res["nb3"] < 1e-6 and res["bcast"] == 0.0, res
    assert res["frac2"][0] >= 0.85, res          # the first bucket carries the parameter-heavy deep cells


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_dp_two_ranks_equals_global_batch(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res["worst"] < 1e-5, res
    for rank in range(2):
        assert torch.load(out + ".bcast%d" % rank)["bcast"] == 0.0


def test_gradsync_single_process_is_noop():
    sys.path.insert(0, ROOT)
    from nas_3d_unet_amd.train import GradSync
    g = torch.arange(10, dtype=torch.float32)
    GradSync(g, None, 4).all_reduce()
    assert torch.equal(g, torch.arange(10, dtype=torch.float32))


def _id_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nas_3d_unet_amd import comm
    got = []
    for k in range(3):                       # three communicators in a row: the store keys must not collide
        c = object.__new__(comm.Comm)        # (no RCCL on a CPU box: only the id exchange of Comm.__init__ is exercised)
        c.rank, c.world = rank, world
        mine = bytes([17 * (rank + 1) + k]) * 128
        got.append(c._exchange_id(None, mine))
    sub = dist.new_group([0, 1])
    c = object.__new__(comm.Comm)
    c.rank, c.world = dist.get_rank(sub), dist.get_world_size(sub)
    got.append(c._exchange_id(sub, bytes([99 + rank]) * 128))
    torch.save(got, out + ".id%d" % rank)
    # for_group: CPU tensors / a gloo group keep torch.distributed (no communicator is made)
    assert comm.for_group(None, torch.device("cpu")) is None
    dist.barrier()
    dist.destroy_process_group()


def test_comm_unique_id_travels_through_the_rendezvous_store(tmp_path):
    """nas_3d_unet_amd.comm: rank 0's 128-byte ncclUniqueId reaches every rank through torch.distributed's key-value store -- no
    collective, so ProcessGroupNCCL has no Work object for its watchdog to poll -- once per communicator, keys
    numbered per group in creation order"""
    out = str(tmp_path / "ids")
    mp.spawn(_id_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".id0"), torch.load(out + ".id1")
    assert r0 == r1
    assert r0[:3] == [bytes([17 + k]) * 128 for k in range(3)] and r0[3] == bytes([99]) * 128
