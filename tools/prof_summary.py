#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv per bench step.
usage: tools/prof_summary.py <kernel_stats.csv> <steps-incl-warmup> [top]"""
import csv
import sys

path, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time per step: {tot / 1e6 / steps:.3f} ms   ({len(rows)} distinct kernels)')
print(f'{"ms/step":>9s} {"%":>6s} {"calls/step":>10s} {"avg us":>9s}  kernel')
for r in rows[:top]:
    n = r['Name'].replace('(anonymous namespace)::', '')
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:9.3f} {float(r['Percentage']):6.2f} {int(r['Calls']) / steps:10.1f} "
          f"{float(r['AverageNs']) / 1e3:9.1f}  {n[:120]}")
