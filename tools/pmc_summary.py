#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel name. usage: pmc_summary.py <counter_collection.csv> [name-filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    if filt and filt not in k:
        continue
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k[:100])
    for c, v in sorted(cs.items()):
        print(f'   {c:32s} mean={sum(v)/len(v):.4g}  (n={len(v)})')
