import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops
from tests.helpers import head_case
from svol_amd.modeling.svanet import build_svanet

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg1_frame'
rec = {}
def wrap(fname):
    orig = getattr(ops, fname)
    def f(*a, **k):
        out = orig(*a, **k)
        outs = out if isinstance(out, tuple) else (out,)
        rec.setdefault(cur[0], []).append((fname, [o.detach().float().cpu() for o in outs]))
        return out
    setattr(ops, fname, f)
for fn in ['layer_norm', 'self_attn_ln', 'cross_attn_ln', 'mlp_ln', 'gate', 'linear']:
    wrap(fn)
cur = ['']
z, meta, args, sd, inp, tg = head_case(name)
for dt in ['fp32', 'bf16']:
    cur[0] = dt
    args.compute_dtype = dt
    m = build_svanet(args); m.load_state_dict(sd); m = m.cuda().eval()
    with torch.no_grad():
        out = m(**{k: v.cuda() for k, v in inp.items()})
a, b = rec['fp32'], rec['bf16']
for (n1, o1), (n2, o2) in zip(a, b):
    for i, (x, y) in enumerate(zip(o1, o2)):
        d = (x - y).abs()
        idx = int(d.reshape(-1).argmax())
        print(f'{n1:16s} out{i} shape={tuple(x.shape)} max|ref|={float(x.abs().max()):.3f} maxdiff={float(d.max()):.4f} mean={float(d.mean()):.5f} argmax_flat={idx}')
