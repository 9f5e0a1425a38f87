#!/usr/bin/env python3
"""Where a workgroup of the weight-stationary K = 256 GEMM (gemm_wsp_body) spends its cycles: per-wave s_memtime stamps
(entry, W fragment loads issued, bias loads issued, first slabs' DMA issued, first slab landed, its fragments read, its products done,
exit).  Needs a LAB build:  SVOL_BUILD_DEFS=-DSVOL_WS_LAB python -m svol_amd.build --force ; run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = 'cuda'
M, D = 8 * 6272, 256
dbg = torch.zeros(1024 * 4 * 8, dtype=torch.int64, device=DEV)
os.environ['SVOL_WS_DBG_PTR'] = hex(dbg.data_ptr())
from svol_amd import ops  # noqa: E402

g = torch.Generator(device='cpu').manual_seed(0)
x = (torch.randn(M, D, generator=g)).to(torch.bfloat16).to(DEV)
names = ['W fragment loads issued', 'bias / scale loads issued', 'first slabs: DMA issued', 'first slab landed (vmcnt)', 'barrier + fragments read',
         'first products', 'steady loop + last epilogue']
for N, act, pre in [(256, ops.ACT_NONE, False), (512, ops.ACT_NONE, False), (2048, ops.ACT_NONE, False), (2048, ops.ACT_GELU_D, True)]:
    W = (torch.randn(N, D, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    b = torch.zeros(N, device=DEV)
    fn = (lambda: ops.gemm_nt(x, W, b, act, want_pre=True)) if pre else (lambda: ops.gemm_nt(x, W, b))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    dbg.zero_()
    fn()
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(1024, 4, 8)
    t = t[t[:, 0, 0] > 0]
    d = np.diff(t, axis=2)
    print(f'N = {N} act {act} pre {pre}: {ev0.elapsed_time(ev1) * 100:.1f} us per launch, {t.shape[0]} workgroups; entry -> exit median {np.median(t[:, :, 7] - t[:, :, 0]):.0f} cycles,'
          f' first entry -> last exit {int(t[:, :, 7].max() - t[:, :, 0].min())} (per XCD max {max(int(t[i::8, :, 7].max() - t[i::8, :, 0].min()) for i in range(8))})')
    for k, n in enumerate(names):
        v = d[:, :, k]
        print(f'    {n:34s} median {np.median(v):8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}')
    ent = t[:, :, 0].min(axis=1)
    print(f'    workgroup entries spread over {int(ent.max() - ent.min())} cycles (p50 {int(np.median(ent) - ent.min())})')
