import sys, torch, numpy as np
sys.path.insert(0, '.')
from tests import gpu_checks as G
from tests.helpers import encdec_case, encdec_stack
from svol_amd import synthetic as syn
name, dts = sys.argv[1], sys.argv[2]
dtype = torch.float32 if dts == 'fp32' else torch.bfloat16
z, meta, args, sd, inp = encdec_case(name)
args.compute_dtype = dts
model = G.build_encdec_model(args, meta['head'])
model.load_state_dict(sd); model.cuda().eval()
out = model(*(inp[k].cuda() for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')))
if isinstance(out, tuple): out = out[0]
logits, boxes = encdec_stack(out, meta['head'])
loss = (logits * syn.synth_probe(logits.shape, 'logits').cuda()).sum() + (boxes * syn.synth_probe(boxes.shape, 'boxes').cuda()).sum()
loss.backward()
print('logits err', float((logits.cpu() - torch.from_numpy(z['logits'])).abs().max()))
rows = []
for k, p in model.named_parameters():
    if f'gnone/{k}' in z.files or p.grad is None: continue
    if f'g/{k}' in z.files:
        ref = torch.from_numpy(z[f'g/{k}']).double(); got = p.grad.double().cpu()
    else:
        flat = p.grad.double().cpu().reshape(-1); step = max(1, flat.numel() // 256)
        got = flat[::step][:256]; ref = torch.from_numpy(z[f'gsample/{k}']).double()
    rows.append((float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30), float(ref.abs().max()), k))
for r in sorted(rows, reverse=True)[:12]: print('%.3e  refmax %.3e  %s' % r)
num = den = 0.0
for k, p in model.named_parameters():
    if f'gnone/{k}' in z.files or p.grad is None: continue
    if f'g/{k}' in z.files:
        ref = torch.from_numpy(z[f'g/{k}']).double(); got = p.grad.double().cpu()
    else:
        flat = p.grad.double().cpu().reshape(-1); step = max(1, flat.numel() // 256)
        got = flat[::step][:256]; ref = torch.from_numpy(z[f'gsample/{k}']).double()
    num += float((got - ref).pow(2).sum()); den += float(ref.pow(2).sum())
print('GLOBAL L2 rel', (num / den) ** 0.5, ' max|logits|', float(np.abs(z['logits']).max()),
      ' logits err last', float((logits[-1].cpu() - torch.from_numpy(z['logits'])[-1]).abs().max()),
      ' boxes err', float((boxes.cpu() - torch.from_numpy(z['boxes'])).abs().max()))
