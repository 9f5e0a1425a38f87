#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of bench.py: per queue busy time, idle gaps on the busiest (main) queue and which
kernels follow the largest gaps, for the LAST full step in the trace.  usage: tools/trace_timeline.py <kernel_trace.csv> [steps_in_trace]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:48]
rows.sort(key=lambda r: r['s'])
# step boundaries: the AdamW kernel ends a step
ends = [r['e'] for r in rows if 'adamw_flat' in r['n']]
per_step = collections.defaultdict(list)
marks = sorted(set(ends))
# group adamw launches (5 per step) -> last of each group
step_ends = [marks[i] for i in range(len(marks)) if i + 1 == len(marks) or marks[i + 1] - marks[i] > 2_000_000]
if len(step_ends) < 3:
    sys.exit('could not find step boundaries')
t0, t1 = step_ends[-2], step_ends[-1]
win = [r for r in rows if r['s'] >= t0 and r['e'] <= t1 + 1000]
print(f'last step: {(t1 - t0) / 1e6:.3f} ms, {len(win)} kernels')
byq = collections.defaultdict(list)
for r in win:
    byq[r['Queue_Id']].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -sum(r['e'] - r['s'] for r in kv[1])):
    busy = sum(r['e'] - r['s'] for r in rs)
    print(f'  queue {q}: {len(rs):4d} kernels, busy {busy / 1e6:.3f} ms, span {(rs[-1]["e"] - rs[0]["s"]) / 1e6:.3f} ms')
mainq = max(byq, key=lambda q: sum(r['e'] - r['s'] for r in byq[q]))
rs = byq[mainq]
gaps = []
for a, b in zip(rs, rs[1:]):
    g = b['s'] - a['e']
    if g > 0:
        gaps.append((g, a['n'], b['n']))
print(f'main queue {mainq}: idle between kernels {sum(g for g, _, _ in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; > 20 us:')
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f'   {g / 1e3:8.1f} us  after {a:48s} before {b}')
# union busy of all queues
ev = sorted([(r['s'], 1) for r in win] + [(r['e'], -1) for r in win])
cur = 0; last = None; busy = 0
for t, d in ev:
    if cur > 0: busy += t - last
    cur += d; last = t
print(f'GPU busy with at least one kernel: {busy / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f} ms')
