#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints, for the LAST full train step in it, what each HIP queue did: busy time, the
device-side waits (sync_wait_kernel) with their durations, and how the side queue's kernels overlap the main chain.
    python tools/trace_timeline.py <kernel_trace.csv> [--dump]"""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step ends with adam_kernel on the main queue
adams = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"] and "inc" not in r["Kernel_Name"]]
if len(adams) < 3:
    sys.exit("need >= 3 steps in the trace")
lo, hi = rows[adams[-3]]["e"], rows[adams[-2]]["e"]
step = [r for r in rows if r["s"] >= lo and r["e"] <= hi + 50000]
step = [r for r in step if r["s"] < hi]
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
print("step wall %.1f us, %d kernels" % ((hi - lo) / 1e3, len(step)))
for q, rs in byq.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    waits = [(r["s"], r["e"]) for r in rs if "sync_wait" in r["Kernel_Name"]]
    wt = sum(e - s for s, e in waits)
    print("queue %s: %d kernels, busy %.1f us (of which device-side waits %.1f us in %d), first %.1f last %.1f" % (
        q, len(rs), busy / 1e3, wt / 1e3, len(waits), (rs[0]["s"] - lo) / 1e3, (rs[-1]["e"] - lo) / 1e3))
    for s, e in waits:
        print("    wait at %.1f for %.1f us" % ((s - lo) / 1e3, (e - s) / 1e3))
if "--dump" in sys.argv:
    for r in step:
        print("%8.1f %7.2f q%s %s" % ((r["s"] - lo) / 1e3, (r["e"] - r["s"]) / 1e3, r["Queue_Id"], r["Kernel_Name"][:90]))
