"""Launch table of one supernet search step (eager, one stream): every libn3d call recorded, replay-timed per (entry, shape) group,
summed by entry and by (entry, shape).  usage: search_table.py [top] [file for the launch-order listing]"""
import sys, os, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
import bench, kernel_table
from nas_3d_unet_amd import nas
from nas_3d_unet_amd.train import SearchTrainer
dev = torch.device("cuda")
top = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(1234)
net = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); net.train()
tr = SearchTrainer(net, graph=False, side_wgrad=False)
xn, tn = bench.synthetic_batch(2, 64, 1234); vxn, vtn = bench.synthetic_batch(2, 64, 4321)
x, t, vx, vt = (torch.from_numpy(a).to(dev) for a in (xn, tn, vxn, vtn))
x, vx = bench.to_patch_layout(x), bench.to_patch_layout(vx)
for _ in range(3): tr.step(x, t, vx, vt)
torch.cuda.synchronize()
with kernel_table.Recorder() as rec:
    tr.step(x, t, vx, vt)
    torch.cuda.synchronize()
times, by_name, by_sig, cost = {}, collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(lambda: [0, 0.0]), {}
for name, args in rec.calls:
    sig, flop, byts = kernel_table.describe(name, args)
    if sig not in times:
        try:
            times[sig] = float("nan") if name in kernel_table.NO_REPLAY else kernel_table._time_call(name, args, dev)
        except Exception:
            times[sig] = float("nan")
    us = times[sig]
    if us == us:
        by_name[name][0] += 1; by_name[name][1] += us
        by_sig[(name, kernel_table._shape_text(sig)[:120])][0] += 1; by_sig[(name, kernel_table._shape_text(sig)[:120])][1] += us
        cost[(name, kernel_table._shape_text(sig)[:120])] = (flop, byts)
if len(sys.argv) > 2:     # the whole step in launch order
    with open(sys.argv[2], "w") as f:
        for i, (name, args) in enumerate(rec.calls):
            sig, _, _ = kernel_table.describe(name, args)
            f.write("%4d %7.2f  %-30s %s\n" % (i, times[sig], name, kernel_table._shape_text(sig)[:120]))
tot = sum(v[1] for v in by_name.values())
print("search step, eager single stream: %d libn3d launches, replay-timed sum %.2f ms" % (len(rec.calls), tot / 1e3))
print("== by entry point")
for n, (k, us) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
    print("%5d %9.1f us %5.1f%%  avg %6.2f  %s" % (k, us, 100 * us / tot, us / k, n))
print("== top (entry, shape) groups; roofline of one call: fraction of max(FLOP / 157.3 TF, bytes / 8 TB/s) -- 'lat' = coefficient kernel, no roofline")
for (n, s), (k, us) in sorted(by_sig.items(), key=lambda kv: -kv[1][1])[:top]:
    flop, byts = cost.get((n, s), (None, None))
    if n in kernel_table.LATENCY_ONLY:
        rf = "  lat"
    elif flop is None:
        rf = "    ?"
    else:
        t_m, t_h = flop / 157.3e12 * 1e6, byts / 8e12 * 1e6
        rf = "%s %.2f" % ("M" if t_m > t_h else "H", max(t_m, t_h) / (us / k))
    print("%5d %9.1f us %5.1f%%  avg %6.2f  %6s  %-28s %s" % (k, us, 100 * us / tot, us / k, rf, n, s))
if kernel_table.describe.errors:
    print("cost-model errors:", kernel_table.describe.errors)
