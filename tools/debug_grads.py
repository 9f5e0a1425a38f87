import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import gpu_checks as G
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg1_video'
gr = {}
for dt in (torch.float32, torch.bfloat16):
    z, meta, args, out, ld, tot, model, crit = G.run_head_case(name, dt)
    gr[dt] = {k: p.grad.detach().float().cpu() for k, p in model.named_parameters() if p.grad is not None}
    print(dt, 'loss', float(tot), 'mismatch-free' )
rows = []
for k in gr[torch.float32]:
    a, b = gr[torch.float32][k], gr[torch.bfloat16][k]
    rows.append((float((a - b).abs().max() / (a.abs().max() + 1e-12)), float(a.abs().max()), float((a-b).norm()/(a.norm()+1e-12)), k))
for r in rows:
    print('maxrel=%.3e  max|g|=%.3e  l2rel=%.3e  %s' % r)
