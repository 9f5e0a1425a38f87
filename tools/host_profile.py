#!/usr/bin/env python3
"""Where does the HOST time of one eager training step go?  cProfile over N steps of the bench's step (no sync inside the loop)."""
import cProfile
import os
import pstats
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
def step():
    reducer.zero_grad()
    crit.prepack(tg, args.num_layers, B, args.num_queries, dev)
    out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
    ld = crit(out, tg)
    loss = crit.weighted_total()
    loss.backward()
    reducer.finish()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pr = cProfile.Profile()
pr.enable()
for _ in range(N): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
st.sort_stats('cumtime').print_stats(45)
