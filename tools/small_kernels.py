"""Which host ops launch the tiny fill / add / copy kernels of one eager training step (cfg2 shapes)?"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from svol_amd import parallel
from svol_amd import synthetic as syn
from svol_amd.modeling.loss import build_loss
from svol_amd.modeling.svanet import build_svanet
dev = torch.device('cuda')
args = syn.cfg2_args('video_matcher'); args.compute_dtype = 'bf16'
model = build_svanet(args).to(dev).train(); crit = build_loss(args).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
red = parallel.BucketedGradAllReduce(params, skip=parallel.unused_parameters(model))
opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
B, T, P = 2, 32, 196
inp = {k: v.to(dev) for k, v in syn.synth_inputs(args, B, T, P, seed=1).items()}
tg = syn.synth_targets(B, T, seed=1)
wd = crit.weight_dict
def step():
    red.zero_grad()
    out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
    ld = crit(out, tg)
    loss = sum(ld[k] * wd[k] for k in ld if k in wd)
    loss.backward(); red.finish(); opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
targets = ('aten::fill_', 'aten::zero_', 'aten::add_', 'aten::add', 'aten::copy_', 'aten::mul')
for e in prof.events():
    if e.name in targets:
        chain = []
        p = e.cpu_parent
        while p is not None and len(chain) < 3:
            chain.append(p.name); p = p.cpu_parent
        cnt[(e.name, ' < '.join(chain))] += 1
for (n, c), k in cnt.most_common(40):
    print(f'{k:5d}  {n:14s} {c}')
