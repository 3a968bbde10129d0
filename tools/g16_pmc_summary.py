#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes over tools/g16_pmc.py: median counter value per dispatch for every (kernel, grid size)
pair -- the same kernel template serves several levels, the grid tells them apart -- joined with the kernel-trace durations of
the same command.   usage: g16_pmc_summary.py <dir with pmc_*/ and kt/> <out.json>"""
import csv, glob, json, re, statistics, sys, collections
root, out = sys.argv[1], sys.argv[2]
KEEP = ("gemm16", "bwd16", "tile16", "tile32", "wgrad16")


def short(name):
    m = re.search(r"n3d::(\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in KEEP):
            key = "%s grid=%s wg=%s" % (short(r["Kernel_Name"]), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
            vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/kt/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in KEEP):
            gs = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) if "Grid_Size_X" in r else r.get("Grid_Size", "?")
            ws = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1) if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "?")
            dur["%s grid=%s wg=%s" % (short(r["Kernel_Name"]), gs, ws)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {}
for key, cs in sorted(vals.items()):
    e = {k: statistics.median(v) for k, v in cs.items()}
    e["_dispatches"] = min(len(v) for v in cs.values())
    if key in dur:
        e["duration_ns_median"] = statistics.median(dur[key])
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # KiB counters; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B) -- an upper bound for the
        # 16-byte-per-lane gathers of these kernels, exact for their streamed weights
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
    res[key] = e
res["_note"] = ("rocprofv3 --pmc passes over tools/g16_pmc.py, one counter group per run, medians per dispatch, keyed by kernel and grid size; "
                "duration from a --kernel-trace run of the same command (no counters)")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
