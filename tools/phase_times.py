#!/usr/bin/env python3
"""GPU time of the phases of one eager training step (cfg2, bf16), from events recorded on the main stream at the Python-level
phase boundaries: zero_grad+prepack | forward (incl. the last layer's query half, which the main stream joins) | criterion |
backward (+ reducer.finish) | optimizer.  Unprofiled, so the gaps are the real ones."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import parallel, synthetic as syn
from svol_amd.modeling.loss import build_loss
from svol_amd.modeling.svanet import build_svanet
dev = torch.device('cuda', 0)
B, T, P = 8, 32, 196
args = syn.cfg2_args('video_matcher'); args.compute_dtype = 'bf16'
torch.manual_seed(1)
model = build_svanet(args).to(dev).train(); crit = build_loss(args).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
reducer = parallel.BucketedGradAllReduce(parallel.arrival_order(model), skip=parallel.unused_parameters(model), ordered=True)
opt = parallel.FlatAdamW(reducer, lr=1e-4, weight_decay=1e-4, params=params)
inp = {k: v.to(dev) for k, v in syn.synth_inputs(args, B, T, P, seed=1).items()}
tg = syn.synth_targets(B, T, seed=1)
names = ['zero_grad+prepack', 'forward', 'criterion+total', 'backward (main stream)', 'finish (side + wgrad tails)', 'optimizer']
def step(ev=None):
    def mark(i):
        if ev is not None:
            ev[i].record()
    mark(0)
    reducer.zero_grad()
    crit.prepack(tg, args.num_layers, B, args.num_queries, dev)
    mark(1)
    out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
    mark(2)
    crit(out, tg)
    loss = crit.weighted_total()
    mark(3)
    loss.backward()
    mark(4)
    reducer.finish()
    mark(5)
    opt.step()
    mark(6)
for _ in range(5): step()
torch.cuda.synchronize()
N = 12
for sync_each in (True, False):
    acc = [0.0] * 6
    evs = []
    for _ in range(N):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        step(ev)
        if sync_each:
            torch.cuda.synchronize()
        evs.append(ev)
    torch.cuda.synchronize()
    for ev in evs[2:]:   # the first steps of the free-running loop still start level with the GPU
        for i in range(6):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
    n = len(evs) - 2
    print('== host synchronised with the GPU after every step' if sync_each else '== free-running (the host runs ahead, as in bench.py)')
    for n_, a in zip(names, acc):
        print(f'{n_:20s} {a / n:7.3f} ms')
    print(f'{"sum":20s} {sum(acc) / n:7.3f} ms')
