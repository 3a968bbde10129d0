#!/usr/bin/env python3
"""Timeline of the side-stream schedule of one replayed train step, from wall-clock stamps the two streams store next to every
hand-off (N3D_SIDE_TRACE=1; rocprofv3's kernel trace serialises the streams and cannot show this).
    python tools/side_timeline.py [--size 64] [--dtype f32|bf16] > profiles/r03_side_timeline.txt
Columns: when the MAIN stream published cut i (microseconds from the step's first cut), when the SIDE stream got past its wait on
it (= its weight-gradient group i starts), the lag between the two, and an upper bound of how long the group ran (until the side
stream got past its NEXT wait, which includes any time it spent waiting there)."""
import argparse, os, sys, time
os.environ["N3D_SIDE_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--raw", default=None, help="also write every non-zero stamp (index, 100 MHz ticks) of the step to this file")
ap.add_argument("--drop", action="store_true", help="capture the schedule WITHOUT the weight-gradient launches (hand-offs only): main stream free of contention")
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev)
net.train()
tr = Trainer(net, graph=True, side_wgrad="force", storage="bf16" if args.dtype == "bf16" else None)
xn, tn = bench.synthetic_batch(2, args.size, 1234)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
if args.drop:
    from nas_3d_unet_amd import kernels as K
    K._DROP_SIDE = True
for _ in range(10):
    tr.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    tr.step(x, t)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 30 * 1e3
tr.check_sync()
st = tr.side.trace.cpu().numpy().astype("int64")
J = tr.side.JOIN
used = int((tr.side.sync[8:8 + J] > 0).sum())
# the stamped hand-offs.  A cut of the backward walk: [2i+2] main stored the flag <= [2i+3] the weight-gradient stream got past its wait <= [2J+8+i]
# its group ended.  A join of the inline side stream: [2i+2] the side stream stored the flag, [2J+8+i] main arrived at its wait < [2i+3] main got past
is_join = lambda i: st[2 * J + 8 + i] != 0 and st[2 * J + 8 + i] < st[2 * i + 3]
cuts = [i for i in range(J) if st[2 * i + 2] != 0 and st[2 * i + 3] != 0 and not is_join(i)]
t0 = int(st[2 * cuts[0] + 2])
us = lambda v: (int(v) - t0) / 100.0      # 100 MHz clock
print("searched-net train step, batch 2, 4x%d^3 %s, side-stream schedule (with the stamps' own launches): %.3f ms per step; %d hand-offs per step, "
      "%d of them cuts of the backward walk (weight-gradient groups), the rest forks / joins of the net's off-chain pieces" % (args.size, args.dtype, ms, used, len(cuts)))
print("%4s %12s %14s %10s %12s %10s" % ("flag", "main signal", "side past wait", "side lag", "group ran", "main seg"))
prev = 0.0
busy = 0.0
for k, i in enumerate(cuts):
    m, w = us(st[2 * i + 2]), us(st[2 * i + 3])
    end = st[2 * J + 8 + i]
    if end:
        ran = us(end) - w
        busy += ran
        txt = "%.1f" % ran
    else:
        nxt = us(st[2 * cuts[k + 1] + 3]) if k + 1 < len(cuts) else us(st[2 * J + 4])
        txt = "<= %.1f" % (nxt - w)
    print("%4d %12.1f %14.1f %10.1f %12s %10.1f" % (i, m, w, w - m, txt, m - prev))
    prev = m
# joins of the inline side stream (off-chain pieces of the net): [2i+2] the side stream stored flag i, [2J+8+i] main arrived at its wait, [2i+3] main past
rows = []
for i in range(J):
    if not (st[2 * i + 2] and st[2 * i + 3] and is_join(i)):
        continue
    arr, sig, past = us(st[2 * J + 8 + i]), us(st[2 * i + 2]), us(st[2 * i + 3])
    if arr <= past and sig <= past:
        rows.append((arr, i, sig, past))
if rows:
    rows.sort()
    print("joins (main waits for the inline side stream): %d; main spent %.1f us at them in total (a pass-through wait costs ~3 us)" % (len(rows), sum(r[3] - r[0] for r in rows)))
    print("%4s %12s %12s %12s %10s" % ("flag", "main arrived", "side stored", "main past", "main wait"))
    for arr, i, sig, past in rows:
        print("%4d %12.1f %12.1f %12.1f %10.1f" % (i, arr, sig, past, past - arr))
span = us(st[2 * J + 6]) if st[2 * J + 6] else us(st[2 * J + 5])
print("busy fractions over the %.1f us from the first cut to the slab reduction: main stream 1.00 by construction (the dependent chain), "
      "weight-gradient stream %.2f (%.1f us of kernels in %d groups); over the whole %.0f us step: %.2f"
      % (span, busy / span if span > 0 else 0.0, busy, len(cuts), ms * 1e3, busy / (ms * 1e3)))
if args.raw:
    with open(args.raw, "w") as f:
        f.write("# raw n3d_stamp dump of one replayed step (100 MHz device clock ticks); index map in train.SideSchedule.__init__: [2i+2] main stored flag i, "
                "[2i+3] weight-gradient / side stream past its wait on flag i, [%d+i] group behind cut i finished, [%d] side done, [%d] main past the join, [%d] slab reduction launched\n"
                % (2 * J + 8, 2 * J + 4, 2 * J + 5, 2 * J + 6))
        for i, v in enumerate(st):
            if v:
                f.write("%d %d\n" % (i, int(v)))
print("side stream done at %.1f us; main stream past the join at %.1f us (main's last cut at %.1f us); slab reduction launched by %.1f us"
      % (us(st[2 * J + 4]), us(st[2 * J + 5]), us(st[2 * cuts[-1] + 2]), us(st[2 * J + 6])))
