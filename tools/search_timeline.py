#!/usr/bin/env python3
"""Timeline of the side-stream schedule of the two passes of one replayed supernet search step (device clock stamps, N3D_SIDE_TRACE=1;
see side_timeline.py).  usage: search_timeline.py > profiles/rNN_search_timeline.txt"""
import os, sys
os.environ["N3D_SIDE_TRACE"] = "1"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, bench
from nas_3d_unet_amd import nas
from nas_3d_unet_amd.train import SearchTrainer
dev = torch.device("cuda")
torch.manual_seed(1234)
net = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); net.train()
tr = SearchTrainer(net, graph=True, side_wgrad="force")
xn, tn = bench.synthetic_batch(2, 64, 1234); vxn, vtn = bench.synthetic_batch(2, 64, 4321)
x, t, vx, vt = (torch.from_numpy(a).to(dev) for a in (xn, tn, vxn, vtn))
x, vx = bench.to_patch_layout(x), bench.to_patch_layout(vx)
for _ in range(6): tr.step(x, t, vx, vt)
torch.cuda.synchronize()
sd = tr.side
J = sd.JOIN
for name, (g_main, g_side, g_tail) in zip(("architecture pass", "weight pass"), tr._side_graphs):
    for rep in range(3):
        sd.trace.zero_()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); sd.raw_replay(g_side); g_main.replay(); e1.record(); g_tail.replay(); e2.record()
        torch.cuda.synchronize()
    st = sd.trace.cpu().numpy().astype("int64")
    flags = [i for i in range(J) if st[2 * i + 2] != 0 and st[2 * i + 3] != 0]
    if not flags:
        print(name, ": no stamped hand-offs"); continue
    t0 = min(int(st[2 * i + 2]) for i in flags)
    us = lambda v: (int(v) - t0) / 100.0
    print("== %s: main graph %.3f ms + tail %.3f ms; %d stamped hand-offs" % (name, e0.elapsed_time(e1), e1.elapsed_time(e2), len(flags)))
    # joins of the inline side stream (forward / backward off-chain edges): [2i+2] side stored the flag, [2J+8+i] main arrived, [2i+3] main past
    joins = [i for i in flags if st[2 * J + 8 + i] != 0 and st[2 * J + 8 + i] < st[2 * i + 3]]
    waited = 0.0
    rows = []
    for i in sorted(joins, key=lambda i: st[2 * J + 8 + i]):
        arr, sig, past = us(st[2 * J + 8 + i]), us(st[2 * i + 2]), us(st[2 * i + 3])
        if sig > past: continue     # a cut (main signals, the other stream waits): listed below
        rows.append((i, arr, sig, past)); waited += past - arr
    if rows:
        print("joins (main waits for the inline side stream): %d, main spent %.1f us at them in total (a pass-through wait costs ~2 us)" % (len(rows), waited))
        print("%4s %12s %12s %12s %10s %12s" % ("flag", "main arrived", "side stored", "main past", "main wait", "side late by"))
        for i, arr, sig, past in rows:
            print("%4d %12.1f %12.1f %12.1f %10.1f %12.1f" % (i, arr, sig, past, past - arr, max(0.0, sig - arr)))
    flags = [i for i in flags if i not in {r[0] for r in rows}]
    print("%4s %12s %14s %10s %12s" % ("flag", "main signal", "other past wait", "lag", "group ran"))
    busy = 0.0
    for i in sorted(flags, key=lambda i: st[2 * i + 2]):
        m, w = us(st[2 * i + 2]), us(st[2 * i + 3])
        end = st[2 * J + 8 + i]
        ran = (us(end) - w) if end else float("nan")
        if end: busy += ran
        print("%4d %12.1f %14.1f %10.1f %12.1f" % (i, m, w, w - m, ran))
    print("side done %.1f us, main past the join %.1f us, slab reduction launched %.1f us; weight-gradient groups busy %.1f us"
          % (us(st[2 * J + 4]), us(st[2 * J + 5]), us(st[2 * J + 6]) if st[2 * J + 6] else float("nan"), busy))
tr.check_sync()
