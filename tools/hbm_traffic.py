#!/usr/bin/env python3
"""Memory-side (HBM + Infinity Cache) bytes per launch and kernel from two rocprofv3 counter passes of the SAME command:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir>/fetch -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <dir>/write -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out prefix> [git-head]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is DOUBLED here
(MI355X_MICROARCH.md, HBM).  Output: <prefix>.json (what bench.py's roofline.traffic reads) and <prefix>.txt."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0].split('<')[0] if not name.startswith('_Z') else name[:80]
        grid = r.get('Grid_Size') or r.get('Grid_Size_X') or '?'
        acc[f'{name}|{grid}'].append(float(r['Counter_Value']))
    return acc


def main():
    fetch, write, prefix = sys.argv[1:4]
    head = sys.argv[4] if len(sys.argv) > 4 else None
    f, w = per_kernel(fetch, 'FETCH_SIZE'), per_kernel(write, 'WRITE_SIZE')
    out = {}
    for k in sorted(set(f) | set(w)):
        rd = 2.0 * sum(f.get(k, [0])) / max(1, len(f.get(k, [0]))) / 1024.0   # KiB -> MiB, doubled (see docstring)
        wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) / 1024.0
        out[k] = {'launches': len(f.get(k, w.get(k, []))), 'read_MB': round(rd, 1), 'write_MB': round(wr, 1)}
    order = sorted(out, key=lambda k: -(out[k]['read_MB'] + out[k]['write_MB']) * out[k]['launches'])
    meta = {'_meta': {'git_head': head, 'command': 'python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (separate --pmc FETCH_SIZE / WRITE_SIZE passes)',
                      'units': 'MiB per launch, memory side; FETCH_SIZE doubled (gfx950)'}}
    json.dump({**meta, **{k: out[k] for k in order}}, open(prefix + '.json', 'w'), indent=1)
    with open(prefix + '.txt', 'w') as t:
        t.write(f'# git {head}; memory-side MiB per launch (FETCH_SIZE x 2, WRITE_SIZE), separate rocprofv3 --pmc passes of bench.py --steps 2 --warmup 1\n')
        t.write(f'{"kernel | grid":84s} {"launches":>8s} {"read MB":>9s} {"write MB":>9s}\n')
        for k in order[:40]:
            t.write(f'{k:84s} {out[k]["launches"]:8d} {out[k]["read_MB"]:9.1f} {out[k]["write_MB"]:9.1f}\n')
    print(open(prefix + '.txt').read())


if __name__ == '__main__':
    main()
