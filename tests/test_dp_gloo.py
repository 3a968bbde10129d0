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
