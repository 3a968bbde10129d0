#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/collect_pmc_r02.sh for one kernel: median counter value per dispatch.
   usage: pmc_summary.py <dir with pmc_*/> <kernel substring> <out.json>"""
import csv, glob, json, statistics, sys, collections
root, kname, out = sys.argv[1], sys.argv[2], sys.argv[3]
vals = collections.defaultdict(list)
for f in glob.glob(root + '/pmc_*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if kname in r['Kernel_Name']:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: statistics.median(v) for k, v in vals.items()}
res['_dispatches'] = {k: len(v) for k, v in vals.items()}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
