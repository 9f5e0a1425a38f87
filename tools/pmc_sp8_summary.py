#!/usr/bin/env python3
"""Per-dispatch SQ counters of tools/pmc_sp8.sh, grouped by (kernel, waves per SIMD) and normalised per 32 x 32 score block and SIMD.
Every variant launches 256 workgroups (one per CU) of 256 threads (one wave per SIMD) twice, then of 512 threads (two) twice, 4000
iterations of its asm body each; step4's body = 4 blocks (40 gaps), the sp8 bodies = 2 blocks (20 gaps) per wave."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
ITERS = 4000
BLOCKS = {'k_step4': 4, 'k_sp8_phase': 2, 'k_sp8_fine': 2}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, '*', '**', '*counter_collection.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k not in BLOCKS:
            continue
        wg = int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 0)) or 0)
        acc[(k, wg // 256)][r['Counter_Name']].append(float(r['Counter_Value']))
print('per 32x32 block and SIMD (counter sums over the chip / (1024 SIMDs x blocks per wave x waves per SIMD x iterations)); n = dispatches')
for (k, wps), cs in sorted(acc.items()):
    blocks_per_simd = BLOCKS[k] * wps * ITERS
    print(f'{k}  {wps} wave(s) per SIMD')
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        per = m / (1024.0 * blocks_per_simd)
        # SQ_*_CYCLES counters tick per wave (or per SIMD for the BUSY ones) in units the raw dump states; both are printed
        print(f'   {c:30s} raw mean {m:14.5g}   per block per SIMD {per:10.2f}   (n={len(v)})')
