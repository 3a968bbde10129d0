#!/usr/bin/env python3
"""Probe (round 3): does running the two samples of the B=2 train step as two concurrently launched
HIP-graph chains on two streams beat the single batched chain?  Timing only -- both chains write the same gradient slices.

    python tools/two_chain_probe.py [--size 64] [--steps 30]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nas_3d_unet_amd import fused as F, head as H, kernels as K, programs as P, searched  # noqa: E402
from nas_3d_unet_amd.train import Trainer, _Ctx  # noqa: E402


def chain(tr, ctx, x, t, seed, pack):
    m = tr.model
    plan = getattr(m, "_net_plan", None)
    if plan is None:
        plan = m._net_plan = F.net_plan(m, supernet=False)
    op = m.last_conv[0]
    with torch.no_grad(), K.step_context(ctx):
        if pack:
            ctx.pack_all()
        nctx = _Ctx((False, False) + (False,) * 4 + (True,) * len(plan.params))
        body = F.NetFn.forward(nctx, plan, x, None, None, None, None, *plan.params)
        gate = P.draw_gate(op.dropout, op.training, body.shape[0], body.shape[1], body.device)
        hctx = _Ctx((False, False, True, False, True, True))
        loss, _ = H.HeadDiceFn.forward(hctx, gate, 1e-6, body, t, op.conv.weight, op.conv.bias)
        dbody = H.HeadDiceFn.backward(hctx, seed, None)[2]
        prev, F.REUSE_GRAD_OUTPUT = F.REUSE_GRAD_OUTPUT, True
        try:
            F.NetFn.backward(nctx, dbody)
        finally:
            F.REUSE_GRAD_OUTPUT = prev
        ctx.flush_final()
    return loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev)
    net.train()
    tr = Trainer(net, graph=True, side_wgrad=False)   # the plain single-stream trainer: this probe makes its own streams
    xn, tn = bench.synthetic_batch(2, args.size, 1234)
    x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)

    def timed(fn, n=args.steps, w=5):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print("V0 batched B=2 graph: %.3f ms" % timed(lambda: tr.step(x, t)), flush=True)

    # per-sample chains
    seed = torch.full((), 0.5, device=dev)
    xs = [x[i:i + 1].contiguous(memory_format=torch.channels_last_3d) for i in range(2)]
    ts = [t[i:i + 1].contiguous() for i in range(2)]
    ctxA = tr.ctx                      # frozen by V0: packed slots exist
    ctxB = K.StepContext(dev)
    ctxB.slots, ctxB.frozen, ctxB.buf, ctxB.jobs, ctxB.njobs = ctxA.slots, True, ctxA.buf, ctxA.jobs, 0
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    # warm-up (allocator, lazily built programs for B=1)
    with torch.cuda.stream(s1):
        for _ in range(2):
            chain(tr, ctxA, xs[0], ts[0], seed, True)
            chain(tr, ctxB, xs[1], ts[1], seed, False)
    torch.cuda.synchronize()
    gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.stream(s1):
        with torch.cuda.graph(gA, capture_error_mode="thread_local"):
            lA = chain(tr, ctxA, xs[0], ts[0], seed, False)
    torch.cuda.synchronize()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(gB, capture_error_mode="thread_local"):
            lB = chain(tr, ctxB, xs[1], ts[1], seed, False)
    torch.cuda.synchronize()
    main_s = torch.cuda.current_stream()

    def pack():
        with K.step_context(ctxA):
            ctxA.pack_all()

    def v1():
        pack()
        gA.replay()
        tr._update()

    def v2():
        pack()
        gA.replay()
        gB.replay()
        tr._update()

    ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()

    def v3():
        pack()
        ev_f.record(main_s)
        s2.wait_event(ev_f)
        with torch.cuda.stream(s2):
            gB.replay()
            ev_j.record(s2)
        gA.replay()
        main_s.wait_event(ev_j)
        tr._update()

    print("V1 one B=1 chain alone (+pack+adam): %.3f ms" % timed(v1), flush=True)
    print("V2 two B=1 chains, same stream: %.3f ms" % timed(v2), flush=True)
    print("V3 two B=1 chains, two streams: %.3f ms" % timed(v3), flush=True)
    print("V3 again: %.3f ms" % timed(v3), flush=True)
    print("losses", float(lA), float(lB))

    # variant: launch both graphs from two host threads
    import threading

    def v4():
        pack()
        ev_f.record(main_s)
        s2.wait_event(ev_f)

        def side():
            with torch.cuda.stream(s2):
                gB.replay()
                ev_j.record(s2)
        th = threading.Thread(target=side)
        th.start()
        gA.replay()
        th.join()
        main_s.wait_event(ev_j)
        tr._update()

    print("V4 two streams, two host threads: %.3f ms" % timed(v4), flush=True)


if __name__ == "__main__":
    main()
