#!/usr/bin/env python3
"""Summarise one supernet search step (two Adam launches) from a rocprofv3 kernel trace: tools/search_profile.py <dir> [n]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
step = rows[idx[-3] + 1:idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
print('kernels', len(step), 'span us %.1f busy %.1f' % ((t1 - t0) / 1e3, busy / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r['Kernel_Name'][:90]; d = int(r['End_Timestamp']) - int(r['Start_Timestamp']); agg[n][0] += 1; agg[n][1] += d
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print('%5d %9.1f us avg %6.2f  %s' % (c, d / 1e3, d / 1e3 / c, n))
