#!/usr/bin/env python3
"""The forward -> backward TURN in isolation (cfg2 shapes, nothing else on the GPU): heads forward, criterion (cost, LSAP, losses),
weighted total, and the backward down to d(hs) — the 800-row chain the video stream waits for between its last forward kernel and its
first backward kernel.  Wall time per phase (host + GPU, events), to see what a fused program could save.
    python tools/turn_bench.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops, synthetic as syn
from svol_amd.modeling.loss import build_loss
from svol_amd.modeling.svanet import build_svanet

dev = torch.device('cuda', 0)
B, T, N, d, NL = 8, 32, 100, 256, 6
args = syn.cfg2_args('video_matcher'); args.compute_dtype = 'bf16'
torch.manual_seed(1)
model = build_svanet(args).to(dev).train()
crit = build_loss(args).to(dev).train()
tg = syn.synth_targets(B, T, seed=1)
ops.weights.new_epoch()
outs = [torch.randn(B, N, d, device=dev).requires_grad_(True) for _ in range(NL)]


def phases():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    crit.prepack(tg, NL, B, N, dev)
    torch.cuda.synchronize()
    if os.environ.get('TURN_AHEAD'):   # park the GPU behind a spin kernel so that the host issues the whole chain AHEAD of it: GPU-side time
        torch.cuda._sleep(int(float(os.environ['TURN_AHEAD']) * 2.0e6))   # (~ms at ~2 GHz)
    ev[0].record()
    hs = torch.stack(outs)
    if ops.heads_fusable(hs, model.class_embed, model.bbox_embed):   # (SVOL_NO_FUSED_HEADS=1: the per-Linear path)
        lg, bx = ops.heads(hs, model.class_embed, model.bbox_embed)
    else:
        lg = ops.linear(hs, model.class_embed.weight, model.class_embed.bias)
        bx = model.bbox_embed(hs, last_act=ops.ACT_SIGMOID)
    ev[1].record()
    out = {'pred_logits': lg[-1], 'pred_boxes': bx[-1], 'aux_outputs': [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(lg[:-1], bx[:-1])],
           '_svol_stacked': (lg, bx)}
    crit(out, tg)
    ev[2].record()
    tot = crit.weighted_total()
    ev[3].record()
    g = torch.autograd.grad(tot, outs + [p for p in list(model.bbox_embed.parameters()) + list(model.class_embed.parameters())])
    ev[4].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]


for _ in range(5):
    phases()
acc = [0.0] * 4
n = 20
t0 = time.perf_counter()
for _ in range(n):
    p = phases()
    acc = [a + b for a, b in zip(acc, p)]
wall = (time.perf_counter() - t0) / n * 1e3
names = ['heads forward (stack + class + 3-layer box MLP)', 'criterion (cost + LSAP + losses)', 'weighted total', 'backward to d(hs) + head weight gradients']
for nm, a in zip(names, acc):
    print(f'{nm:60s} {a / n * 1e3:8.1f} us')
print(f'{"sum":60s} {sum(acc) / n * 1e3:8.1f} us   (wall incl. prepack + sync {wall:.3f} ms)')
