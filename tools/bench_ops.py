#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the benchmark (cfg2) shapes, timed with HIP events.
    python tools/bench_ops.py [attn] [gemm] [ln] [--iters N]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops

DEV = 'cuda'
B, H, L, DH, D, F, NQ = 8, 8, 6272, 32, 256, 2048, 100
ITERS = 10
for i, a in enumerate(sys.argv):
    if a == '--iters':
        ITERS = int(sys.argv[i + 1])
which = [a for a in sys.argv[1:] if not a.startswith('--') and not a.isdigit()] or ['attn', 'gemm', 'ln']


def timeit(fn, iters=ITERS, warm=2):
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


def rnd(*shape, dt=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).to(dt)


if 'attn' in which:
    for dt in (torch.bfloat16,):
        qkv = rnd(B * L, 3 * D, dt=dt)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        o, lse = ops.attn_fwd(q, k, v, B, H, L, L, DH)
        do = rnd(B * L, D, dt=dt)
        dqkv = torch.empty_like(qkv)
        fl = 4.0 * L * L * D * B
        for pm in (0.0, 1.4426950408889634 / DH ** 0.5):
            o, lse = ops.attn_fwd(q, k, v, B, H, L, L, DH, None, pm)
            t = timeit(lambda: ops.attn_fwd(q, k, v, B, H, L, L, DH, None, pm))
            print(f'attn_fwd  premul={pm:.3f}: {t:.3f} ms  {fl / t / 1e9:.1f} TFLOP/s (algorithmic 4L^2dB)')
            t = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, H, L, L, DH, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], None, pm))
            print(f'attn_bwd  premul={pm:.3f}: {t:.3f} ms  {2 * fl / t / 1e9:.1f} TFLOP/s (algorithmic 2x fwd)')
        # cross attention shape
        qc = rnd(B * NQ, D, dt=dt)
        kb = torch.zeros(B, L, device=DEV)
        t = timeit(lambda: ops.attn_fwd(qc, k, v, B, H, NQ, L, DH, kb))
        print(f'cross_attn_fwd (Lq=100): {t:.3f} ms')

if 'gemm' in which:
    M = B * L
    for (n, kk, name) in [(2 * D, D, 'qk proj'), (D, D, 'v/out proj'), (F, D, 'fc1'), (D, F, 'fc2'), (D, 512, 'in proj0')]:
        A = rnd(M, kk)
        W = rnd(n, kk, scale=0.05)
        bias = torch.zeros(n, device=DEV)
        t = timeit(lambda: ops.gemm_nt(A, W, bias))
        print(f'gemm_nt {name:12s} M={M} N={n} K={kk}: {t:.3f} ms  {2.0 * M * n * kk / t / 1e9:.1f} TFLOP/s')
        dY = rnd(M, n)
        t = timeit(lambda: ops.gemm_tn(dY, A))
        print(f'gemm_tn {name:12s} Mc={M} N={n} K={kk}: {t:.3f} ms  {2.0 * M * n * kk / t / 1e9:.1f} TFLOP/s')
    X = rnd(M, F)
    t = timeit(lambda: ops.colsum(X))
    print(f'colsum [{M},{F}]: {t:.3f} ms  {X.numel() * 2 / t / 1e6:.0f} GB/s')
    t = timeit(lambda: ops.act_bwd(X, X, ops.ACT_GELU))
    print(f'act_bwd gelu [{M},{F}]: {t:.3f} ms  {X.numel() * 6 / t / 1e6:.0f} GB/s')

if 'gemm2' in which:  # the step's GEMM variants with their real epilogues; GB/s over algorithmic HBM bytes
    M = B * L
    x = rnd(M, D)
    x32 = torch.randn(M, D, device=DEV)
    hid = rnd(M, F)
    W_dd, W_fd, W_df = rnd(D, D, scale=0.05), rnd(F, D, scale=0.05), rnd(D, F, scale=0.05)
    W_2dd = rnd(2 * D, D, scale=0.05)
    b_d, b_f, b_2d = torch.zeros(D, device=DEV), torch.zeros(F, device=DEV), torch.zeros(2 * D, device=DEV)
    qs = torch.ones(2 * D, device=DEV)
    qkv = torch.empty((M, 3 * D), dtype=torch.bfloat16, device=DEV)
    x2 = rnd(M, 2 * D)
    cases = [
        ('qk proj  N=512 K=256 colscale', lambda: ops.gemm_nt(x, W_2dd, b_2d, out=qkv[:, :2 * D], colscale=qs), 2 * D, D, M * D * 2 + M * 2 * D * 2),
        ('v proj   N=256 K=256', lambda: ops.gemm_nt(x, W_dd, b_d, out=qkv[:, 2 * D:]), D, D, M * D * 4),
        ('out proj N=256 K=256 f32+res', lambda: ops.gemm_nt(x, W_dd, b_d, residual=x32, out_f32=True), D, D, M * D * (2 + 4 + 4)),
        ('fc1      N=1024 K=256 gelu+pre', lambda: ops.gemm_nt(x, W_fd, b_f, ops.ACT_GELU, want_pre=True), F, D, M * D * 2 + M * F * 4),
        ('dgelu    N=1024 K=256', lambda: ops.gemm_nt_dgelu(x, W_fd, hid), F, D, M * D * 2 + M * F * 4),
        ('fc2      N=256 K=1024 f32+res', lambda: ops.gemm_nt(hid, W_df, b_d, residual=x32, out_f32=True), D, F, M * F * 2 + M * D * 8),
        ('dx mlp   N=256 K=1024', lambda: ops.gemm_nt(hid, W_df), D, F, M * F * 2 + M * D * 2),
        ('dx qk    N=256 K=512', lambda: ops.gemm_nt(x2, rnd(D, 2 * D, scale=0.05)), D, 2 * D, M * 2 * D * 2 + M * D * 2),
    ]
    for name, fn, n, kk, byts in cases:
        t = timeit(fn)
        print(f'{name:34s}: {t * 1e3:7.1f} us  {2.0 * M * n * kk / t / 1e9:6.1f} TFLOP/s  {byts / t / 1e6:6.0f} GB/s')
    g = rnd(M, D)
    for name, fn, n, kk, byts in [
        ('tn dW proj N=256 K=256', lambda: ops.gemm_tn(g, x), D, D, M * D * 4),
        ('tn dW qk   N=512 K=256', lambda: ops.gemm_tn(x2, x), 2 * D, D, M * D * 6),
        ('tn dW fc2  N=256 K=1024', lambda: ops.gemm_tn(g, hid), D, F, M * (D + F) * 2),
        ('tn dW fc1  N=1024 K=256', lambda: ops.gemm_tn(hid, x), F, D, M * (D + F) * 2),
    ]:
        t = timeit(fn)
        print(f'{name:34s}: {t * 1e3:7.1f} us  {2.0 * M * n * kk / t / 1e9:6.1f} TFLOP/s  {byts / t / 1e6:6.0f} GB/s')

if 'gemm3' in which:  # fc1 epilogue ablation: what bounds the weight-stationary kernel at N = F
    M = B * L
    x = rnd(M, D)
    W_fd = rnd(F, D, scale=0.05)
    b_f = torch.zeros(F, device=DEV)
    for name, fn, byts in [
        ('fc1 none', lambda: ops.gemm_nt(x, W_fd, b_f), M * D * 2 + M * F * 2),
        ('fc1 relu', lambda: ops.gemm_nt(x, W_fd, b_f, ops.ACT_RELU), M * D * 2 + M * F * 2),
        ('fc1 gelu', lambda: ops.gemm_nt(x, W_fd, b_f, ops.ACT_GELU), M * D * 2 + M * F * 2),
        ('fc1 relu+pre', lambda: ops.gemm_nt(x, W_fd, b_f, ops.ACT_RELU, want_pre=True), M * D * 2 + M * F * 4),
        ('fc1 gelu+pre', lambda: ops.gemm_nt(x, W_fd, b_f, ops.ACT_GELU, want_pre=True), M * D * 2 + M * F * 4),
        ('fill [M,F] bf16', (lambda o=torch.empty((M, F), dtype=torch.bfloat16, device=DEV): o.fill_(1.0)), M * F * 2),
    ]:
        t = timeit(fn)
        print(f'{name:34s}: {t * 1e3:7.1f} us  {byts / t / 1e6:6.0f} GB/s')

if 'gemm4' in which:  # round 5: the video half's MLP as the step runs it (fc1 saves gelu', the backward multiplies by it)
    M = B * L
    x = rnd(M, D)
    W_fd = rnd(F, D, scale=0.05)
    W_df = rnd(D, F, scale=0.05)
    b_f = torch.zeros(F, device=DEV)
    dpre = rnd(M, F)
    cs = torch.zeros(F, dtype=torch.float32, device=DEV)
    for name, fn, byts in [
        ('fc1 gelu + gelu\' saved', lambda: ops.gemm_nt(x, W_fd, b_f, ops.ACT_GELU_D, want_pre=True), M * D * 2 + M * F * 4),
        ('dmul (dY W2) * aux + colsum', lambda: ops.gemm_nt_dact(x, W_fd, dpre, ops.ACT_GELU_D, colsum_out=cs), M * D * 2 + M * F * 4),
        ('fc1 none', lambda: ops.gemm_nt(x, W_fd, b_f), M * D * 2 + M * F * 2),
        ('v proj N=256', lambda: ops.gemm_nt(x, W_df[:, :D].contiguous(), b_f[:D].contiguous()), M * D * 4),
        ('qk proj N=512', lambda: ops.gemm_nt(x, W_fd[:2 * D].contiguous(), b_f[:2 * D].contiguous(), colscale=torch.ones(2 * D, device=DEV)), M * D * 6),
    ]:
        t = timeit(fn)
        print(f'{name:34s}: {t * 1e3:7.1f} us  {byts / t / 1e6:6.0f} GB/s')

if 'ln' in which:
    M = B * L
    x = torch.randn(M, D, device=DEV)
    g, b_ = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    pos = rnd(M, D)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b_, torch.bfloat16, pos, want32=True))
    print(f'ln_fwd [{M},{D}] fp32->fp32+2bf16: {t:.3f} ms  {M * D * (4 + 4 + 2 + 2 + 2) / t / 1e6:.0f} GB/s')
    y32, y, yp, mean, rstd = ops.layernorm_fwd(x, g, b_, torch.bfloat16, pos, want32=True)
    t = timeit(lambda: ops.layernorm_bwd(x, y, yp, x, g, mean, rstd, torch.bfloat16, want32=True))
    print(f'ln_bwd [{M},{D}]: {t:.3f} ms  {M * D * (4 + 2 + 2 + 4 + 4 + 2) / t / 1e6:.0f} GB/s')

if 'gate' in which:
    M = B * L
    x32 = torch.randn(B, L, D, device=DEV)
    pos = rnd(B, L, D)
    u = torch.randn(B, H, D, device=DEV) * 0.05
    g, b_ = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    x32.requires_grad_(True); u.requires_grad_(True)
    y32, y, yp = ops.gate(x32, pos, u, g, b_, H)
    t = timeit(lambda: ops.gate(x32, pos, u, g, b_, H))
    print(f'gate fwd: {t * 1e3:.1f} us')
    dy32, dy, dyp = torch.randn_like(y32), torch.randn_like(y), torch.randn_like(yp)
    def bw():
        torch.autograd.grad((y32, y, yp), (x32, u), (dy32, dy, dyp), retain_graph=True)
    t = timeit(bw)
    print(f'gate bwd: {t * 1e3:.1f} us')
