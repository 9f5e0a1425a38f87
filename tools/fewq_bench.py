#!/usr/bin/env python3
"""The query -> video cross-attention backward launch (B 8, H 8, 100 queries, 6272 keys, key-padding bias) alone: wall time per call
and parity of the single-pass few-query kernel (attn_bwd_fq_bf16) against the two-pass kernels (SVOL_ATTN_NO_FEWQ=1 in a second run).
    python tools/fewq_bench.py [out.pt]   /   SVOL_ATTN_NO_FEWQ=1 python tools/fewq_bench.py [ref.pt]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops

dev = 'cuda'
B, H, Lq, Lk, dh = 8, 8, 100, 6272, 32
D = H * dh
g = torch.Generator(device=dev).manual_seed(3)
dt = torch.bfloat16
pm = 1.4426950408889634 / dh ** 0.5
q = (torch.randn(B * Lq, D, device=dev, generator=g) * 0.5 * pm).to(dt)
kv = (torch.randn(B * Lk, 2 * D, device=dev, generator=g) * 0.5).to(dt)
k, v = kv[:, :D], kv[:, D:]
kb = torch.zeros(B, Lk, device=dev)
kb[1::2, 4704:] = float('-inf')
o, lse = ops.attn_fwd(q, k, v, B, H, Lq, Lk, dh, kb, pm)
do = (torch.randn(B * Lq, D, device=dev, generator=g) * 0.5).to(dt)
dq = torch.empty_like(q)
dkv = torch.empty_like(kv)


def run():
    ops.attn_bwd(q, k, v, o, do, lse, B, H, Lq, Lk, dh, dq, dkv[:, :D], dkv[:, D:], kb, pm)


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print(f'cross-attention backward (B{B} H{H} q{Lq} k{Lk}): {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call'
      f'  [{"two-pass kernels" if os.environ.get("SVOL_ATTN_NO_FEWQ") else "single-pass few-query kernel"}]')
if len(sys.argv) > 1:
    torch.save({'dq': dq.float().cpu(), 'dkv': dkv.float().cpu()}, sys.argv[1])
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[2])
    for n_, a_, b_ in (('dq', dq.float().cpu(), ref['dq']), ('dk', dkv[:, :D].float().cpu(), ref['dkv'][:, :D]), ('dv', dkv[:, D:].float().cpu(), ref['dkv'][:, D:])):
        print(f'  {n_}: max |diff| {float((a_ - b_).abs().max()):.3e} of max {float(b_.abs().max()):.3e}; rel L2 {float((a_ - b_).norm() / b_.norm()):.3e}')
