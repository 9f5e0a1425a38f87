#!/usr/bin/env python3
"""Throughput of the ViT-B/16 extractor's GEMM shapes (256 frames x 197 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops

def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

M = 256 * 197
for (N, K, act, name) in [(2304, 768, ops.ACT_NONE, 'qkv'), (768, 768, ops.ACT_NONE, 'out'), (3072, 768, ops.ACT_GELU, 'fc1'), (768, 3072, ops.ACT_NONE, 'fc2')]:
    A = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.zeros(N, device='cuda')
    t = timeit(lambda: ops.gemm_nt(A, W, b, act))
    print(f'{name:4s} M={M} N={N} K={K}: {t:.3f} ms  {2.0 * M * N * K / t / 1e9:.0f} TFLOP/s', flush=True)
# the fp32-residual-stream variants (out-proj and fc2 write x32 + proj into the fp32 stream)
for (N, K, name) in [(768, 768, 'out+res32'), (768, 3072, 'fc2+res32')]:
    A = (torch.randn(M, K, device='cuda') * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.zeros(N, device='cuda')
    R = torch.randn(M, N, device='cuda')
    t = timeit(lambda: ops.gemm_nt(A, W, b, ops.ACT_NONE, residual=R, out_f32=True))
    gb = (M * K * 2 + 2 * M * N * 4) / 1e9
    print(f'{name:10s} M={M} N={N} K={K}: {t:.3f} ms  {2.0 * M * N * K / t / 1e9:.0f} TFLOP/s  {gb / t * 1e3:.0f} GB/s algorithmic', flush=True)
