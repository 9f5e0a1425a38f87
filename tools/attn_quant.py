#!/usr/bin/env python3
"""Workgroup-count quantisation of the attention kernels: time vs sequence length around the benchmark's L = 6272
(B*H = 64 heads; forward / dK,dV: L/128 workgroups per head, dQ: ceil(L/256))."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops

B, H, DH, D = 8, 8, 32, 256
pm = 1.4426950408889634 / DH ** 0.5


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for L in [int(a) for a in sys.argv[1:]] or [5120, 6016, 6144, 6272, 6400, 6656, 7168, 8192]:
    qkv = (torch.randn(B * L, 3 * D, device='cuda')).to(torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    do = torch.randn(B * L, D, device='cuda').to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    o, lse = ops.attn_fwd(q, k, v, B, H, L, L, DH, None, pm)
    tf = timeit(lambda: ops.attn_fwd(q, k, v, B, H, L, L, DH, None, pm))
    tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, DH, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], None, pm))
    u = (L / 6272) ** 2
    print(f'L={L:5d} wg128={L // 128 * 64:5d} wg256={(L + 255) // 256 * 64:5d}  fwd {tf:.3f} ms ({tf / u:.3f} per 6272^2)   '
          f'bwd {tb:.3f} ms ({tb / u:.3f} per 6272^2)', flush=True)
