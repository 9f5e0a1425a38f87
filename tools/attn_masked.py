#!/usr/bin/env python3
"""Masked / ragged attention launches against the unmasked multiple-of-128 case (B*H = 64 heads, d_h = 32, pre-multiplied q)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops

B, H, DH, D = 8, 8, 32, 256
pm = 1.4426950408889634 / DH ** 0.5


def timeit(fn, iters=30, warm=10):
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


for (L, mode) in [(6272, 'none'), (6272, 'none'), (6272, 'zeros'), (6272, 'pad25'), (6273, 'none'), (6273, 'zeros'), (6400, 'zeros')]:
    qkv = (torch.randn(B * L, 3 * D, device='cuda')).to(torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    do = torch.randn(B * L, D, device='cuda').to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    kb = None
    if mode != 'none':
        kb = torch.zeros(B, L, device='cuda')
        if mode == 'pad25':
            kb[:, L - L // 4:] = float('-inf')
    o, lse = ops.attn_fwd(q, k, v, B, H, L, L, DH, kb, pm)
    tf = timeit(lambda: ops.attn_fwd(q, k, v, B, H, L, L, DH, kb, pm))
    tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, DH, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], kb, pm))
    print(f'L={L:5d} kbias={mode:6s}  fwd {tf:.3f} ms   bwd {tb:.3f} ms', flush=True)
