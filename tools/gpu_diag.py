#!/usr/bin/env python3
"""Run every GPU parity check and print a table (does not stop at the first failure).
    python tools/gpu_diag.py [filter-substring ...] > gpurun_out/diag.txt
"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests import gpu_checks as G  # noqa: E402
from tests.helpers import load_golden  # noqa: E402


def main():
    filt = sys.argv[1:]
    groups = [
        ('gemm_nt', G.check_gemm_nt), ('gemm_dgelu', G.check_gemm_dgelu), ('gemm_tn', G.check_gemm_tn), ('small', G.check_small_ops),
        ('layernorm', G.check_layernorm), ('posenc', G.check_posenc), ('attention', G.check_attention),
        ('gate', G.check_gate), ('lsap', G.check_lsap_vs_scipy),
    ]
    for n in ['crit_video_B8_N100_T32', 'crit_frame_B8_N320_T32', 'crit_frame_B3_N8_T4_over', 'crit_video_B2_N4_T4_tall']:
        groups.append((n, lambda n=n: G.check_criterion(*load_golden(n), n)))
    for n in ['tiny_video', 'tiny_frame', 'cfg1_video', 'cfg1_frame', 'mid_video', 'mid32_video']:
        for dt in (torch.float32, torch.bfloat16):
            groups.append((f'head_{n}_{dt}', lambda n=n, dt=dt: G.check_head_case(n, dt)))
    nfail = 0
    for name, fn in groups:
        if filt and not any(f in name for f in filt):
            continue
        t0 = time.time()
        try:
            res = fn()
            torch.cuda.synchronize()
        except Exception:
            print(f'[{name}] EXCEPTION\n{traceback.format_exc()}', flush=True)
            nfail += 1
            continue
        for k, (err, tol) in res.items():
            ok = err <= tol
            nfail += (not ok)
            print(f'{"ok  " if ok else "FAIL"} {k:70s} err={err:.3e} tol={tol:.1e}', flush=True)
        print(f'[{name}] {time.time() - t0:.1f}s', flush=True)
    print(f'TOTAL FAILURES: {nfail}')


if __name__ == '__main__':
    main()
