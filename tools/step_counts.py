import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "adam_kernel" in n]
print("launches", len(names), "adam launches", len(adam))
a, b = adam[-3], adam[-1]        # one full search step = two adam launches
c = collections.Counter(n[:80] for n in names[a + 1:b + 1])
print("launches in the last search step:", b - a)
t = collections.Counter()
for r in rows[a + 1:b + 1]:
    t[r["Kernel_Name"][:80]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, k in c.most_common(200):
    if "copy" in n.lower() or "at::" in n: print("  %5d %8.1f us  %s" % (k, t[n] / 1e3, n))
print("span of the step under the trace: %.2f ms; sum of kernel durations %.2f ms" % ((int(rows[b]["End_Timestamp"]) - int(rows[a + 1]["Start_Timestamp"])) / 1e6, sum(t.values()) / 1e6))
