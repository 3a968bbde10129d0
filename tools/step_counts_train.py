"""Launches of ONE replayed train step by kernel, with their queue, from a rocprofv3 --kernel-trace CSV of bench.py (the last step = between the last
two Adam launches): what is in the step's graphs besides libn3d's kernels (torch fills / copies).   usage: step_counts_train.py <trace dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "adam_kernel" in n]
a, b = adam[-2], adam[-1]
step = rows[a + 1:b + 1]
print("launches in the last step:", len(step), " span %.1f us" % ((int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3))
byq = collections.Counter(r["Queue_Id"] for r in step)
print("by queue:", dict(byq))
c, t = collections.Counter(), collections.Counter()
for r in step:
    k = (r["Queue_Id"], r["Kernel_Name"][:90])
    c[k] += 1; t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("not libn3d:")
for (q, n), k in c.most_common():
    if "n3d::" not in n and "(anonymous namespace)" not in n:
        print("  queue %s %4d x %7.2f us  %s" % (q, k, t[(q, n)] / k / 1e3, n))
print("flag launches:")
for (q, n), k in c.most_common():
    if "sync_" in n:
        print("  queue %s %4d x %7.2f us  %s" % (q, k, t[(q, n)] / k / 1e3, n))
