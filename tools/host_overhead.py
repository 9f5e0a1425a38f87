"""How long does the HOST need to issue one training step (no sync) vs how long the GPU needs to run it?"""
import os, sys, time
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
reducer = parallel.BucketedGradAllReduce(params, skip=parallel.unused_parameters(model))
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True)
inp = {k: v.to(dev) for k, v in syn.synth_inputs(args, B, T, P, seed=1).items()}
tg = syn.synth_targets(B, T, seed=1)
wd = crit.weight_dict
def step(parts=None):
    t = [time.perf_counter()]
    reducer.zero_grad()
    out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask']); t.append(time.perf_counter())
    ld = crit(out, tg); loss = sum(ld[k] * wd[k] for k in ld.keys() if k in wd); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    reducer.finish(); opt.step(); t.append(time.perf_counter())
    if parts is not None: parts.append([b - a for a, b in zip(t, t[1:])])
for _ in range(3): step()
torch.cuda.synchronize()
parts = []
t0 = time.perf_counter()
for _ in range(10): step(parts)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_total = time.perf_counter() - t0
import numpy as np
p = np.array(parts).mean(0) * 1e3
print(f'host issue time per step: {t_issue/10*1e3:.2f} ms (fwd {p[0]:.2f}, criterion {p[1]:.2f}, backward {p[2]:.2f}, finish+opt {p[3]:.2f}); wall per step incl. GPU: {t_total/10*1e3:.2f} ms')
