import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import parallel, ops
from svol_amd import synthetic as syn
from svol_amd.modeling.loss import build_loss
from svol_amd.modeling.svanet import build_svanet
dev = torch.device('cuda')
args = syn.cfg1_args('video_matcher'); args.compute_dtype = 'bf16'
model = build_svanet(args).to(dev).train(); crit = build_loss(args).to(dev)
params = list(model.parameters())
red = parallel.BucketedGradAllReduce(params, bucket_bytes=1 << 20, skip=parallel.unused_parameters(model))
fired = {}
names = {id(p): n for n, p in model.named_parameters()}
for b in red.buckets:
    for p in b['params']:
        s = p._svol_sink
        orig = s.ready
        def mk(orig, p):
            def f():
                fired[names[id(p)]] = fired.get(names[id(p)], 0) + 1
                orig()
            return f
        s.ready = mk(orig, p)
        p.register_post_accumulate_grad_hook(lambda p_: fired.__setitem__(names[id(p_)] + ' (autograd)', 1))
inp = {k: v.to(dev) for k, v in syn.synth_inputs(args, 2, 8, 49, seed=1).items()}
tg = syn.synth_targets(2, 8, seed=1)
red.zero_grad()
out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
ld = crit(out, tg); wd = crit.weight_dict
loss = sum(ld[k] * wd[k] for k in ld if k in wd); loss.backward()
sunk = [n for n in fired if not n.endswith('(autograd)')]
auto = [n for n in fired if n.endswith('(autograd)')]
print('sunk', len(sunk), 'autograd', len(auto))
print('autograd:', auto)
allp = [names[id(p)] for b in red.buckets for p in b['params']]
print('never:', [n for n in allp if n not in fired and n + ' (autograd)' not in fired])
print('pending', [b['pending'] for b in red.buckets])
