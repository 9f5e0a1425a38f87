#!/usr/bin/env python3
"""Forward-pass memory-side traffic of one bench step, from a rocprofv3 kernel trace (launch order) and the per-kernel
FETCH_SIZE x2 / WRITE_SIZE table of tools/hbm_traffic.py (same command).  A launch belongs to the forward pass when it starts
before the step's lsap_kernel (the matcher sits between forward and backward).
    python tools/fwd_traffic.py <kernel_trace.csv> <hbm_traffic.json> <algorithmic MiB> <label>"""
import csv
import json
import sys


def key(name, grid):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    name = name.split('(')[0].split('<')[0] if not name.startswith('_Z') else name[:80]
    return f'{name}|{grid}'


def main():
    trace, table, algo, label = sys.argv[1], json.load(open(sys.argv[2])), float(sys.argv[3]), sys.argv[4]
    rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Start_Timestamp']))
    lsap = [i for i, r in enumerate(rows) if 'lsap_kernel' in r['Kernel_Name']]
    opt = [i for i, r in enumerate(rows) if 'adamw_flat' in r['Kernel_Name']]
    end = lsap[-1]
    start = max(i for i in opt if i < end) + 1  # first launch after the previous step's optimizer
    rd = wr = 0.0
    t_busy = 0
    missing = set()
    per = {}
    for r in rows[start:end]:
        k = key(r['Kernel_Name'], r.get('Grid_Size') or r.get('Grid_Size_X'))
        e = table.get(k)
        if e is None:
            missing.add(k)
            continue
        rd += e['read_MB']
        wr += e['write_MB']
        d = per.setdefault(k.split('|')[0], [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e['read_MB']
        d[2] += e['write_MB']
    t0, t1 = int(rows[start]['Start_Timestamp']), int(rows[end]['Start_Timestamp'])
    ms = (t1 - t0) / 1e6
    tot = rd + wr
    print(f'# {label}: forward pass of the last traced step ({end - start} launches, {ms:.2f} ms from first launch to the matcher, under the profiler)')
    print(f'memory-side traffic (FETCH_SIZE x 2 + WRITE_SIZE): read {rd:.0f} MiB + write {wr:.0f} MiB = {tot:.0f} MiB')
    print(f'algorithmic forward bytes (SURVEY.md 8d): {algo:.0f} MiB  ->  measured / algorithmic = {tot / algo:.2f}x')
    tbs = tot * 1.048576e6 / (ms * 1e-3) / 1e12
    print(f'forward HBM rate {tbs:.2f} TB/s = {tbs / 8.0 * 100:.1f} % of the ~8 TB/s peak (profiled wall time): the forward is MFMA / issue bound; '
          f'the line is evidence of streaming (nothing L x L is ever resident or moved)')
    print(f'{"kernel":44s} {"launches":>8s} {"read MiB":>9s} {"write MiB":>9s}')
    for k, (n, a, b) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:14]:
        print(f'{k[:44]:44s} {n:8d} {a:9.0f} {b:9.0f}')
    if missing:
        print('# launches without a counter row (tiny, not in the top-40 table):', len(missing))


if __name__ == '__main__':
    main()
