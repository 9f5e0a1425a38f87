#!/usr/bin/env python3
"""The query half of one layer (the N = 100 object queries of 8 videos: the serial chain that is exposed at the end of the forward and
at the start of the backward) in isolation: forward and forward + backward wall time, nothing else on the GPU.
    python tools/query_half_bench.py            (rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops, synthetic as syn
from svol_amd.modeling.svanet import build_svanet

dev = torch.device('cuda', 0)
B, L, N, d = 8, 6272, 100, 256
args = syn.cfg2_args('video_matcher'); args.compute_dtype = 'bf16'
torch.manual_seed(1)
model = build_svanet(args).to(dev).train()
layer = model.transformer.layers[2]
from svol_amd.modeling import cross_modal_transformer as cmt
qdt = torch.float32 if cmt.QUERY_FP32 else torch.bfloat16
m = torch.randn(B, L, d, device=dev).bfloat16().requires_grad_(True)
mpos = torch.randn(B, L, d, device=dev).bfloat16().requires_grad_(True)
qpos = model.query_embed.weight.to(qdt)
kbias = torch.zeros(B, L, device=dev)
def mk():
    o32 = torch.randn(B, N, d, device=dev).requires_grad_(True)
    return (o32, o32.to(qdt), (o32 + qpos).to(qdt))
ops.weights.new_epoch()
def fwd():
    return layer.query_half(mk(), m, mpos, qpos, kbias)
def fb():
    out = fwd()
    g = torch.autograd.grad(out[0].float().sum() + out[1].float().sum() + out[2].float().sum(), (m, mpos), allow_unused=True)
    return g
for f, name in ((fwd, 'forward'), (fb, 'forward + backward')):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    print(f'query half {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms wall (host + GPU, isolated)')
