"""Shared helpers for the test-suite (loading goldens, unpacking indices)."""
import json
import os

import numpy as np
import torch

from svol_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta'])) if 'meta' in z.files else {}
    return z, meta


def head_case(name, dtype=torch.float32):
    """Regenerate (args, state_dict, inputs, targets) of a golden head case."""
    z, meta = load_golden('head_' + name)
    args = syn.head_args(**meta['args'])
    sd = syn.synth_state_dict(args, seed=1)
    inp = syn.synth_inputs(args, meta['B'], meta['T'], meta['P'], seed=1, pad_frames=meta['pad_frames'],
                           pad_all=meta.get('pad_all', False))
    tg = syn.synth_targets(meta['B'], meta['T'], seed=1)
    if dtype != torch.float32:
        sd = {k: v.to(dtype) for k, v in sd.items()}
        inp = {k: v.to(dtype) for k, v in inp.items()}
    return z, meta, args, sd, inp, tg


def recorded_dropout_masks(z, p):
    """the keep masks a train-mode golden recorded from the reference's nn.Dropout modules (make_golden.py::run_train_case), as
    the oracle takes them: {'video': [m_0, ...], 'sketch': [...]}, float32, already scaled by 1 / (1 - p)."""
    out = {}
    for which in ('video', 'sketch'):
        ms = []
        j = 0
        while f'mask/{which}/{j}/bits' in z.files:
            shape = tuple(int(v) for v in z[f'mask/{which}/{j}/shape'])
            n = int(np.prod(shape))
            keep = np.unpackbits(z[f'mask/{which}/{j}/bits'])[:n].reshape(shape).astype(np.float32)
            ms.append(torch.from_numpy(keep) * np.float32(1.0 / (1.0 - p)))
            j += 1
        out[which] = ms
    return out


def unpack_indices(z, prefix):
    p, t, o = z[prefix + '/pred'], z[prefix + '/tgt'], z[prefix + '/offs']
    return [(p[o[i]:o[i + 1]], t[o[i]:o[i + 1]]) for i in range(len(o) - 1)]


def check_grad(z, key, g, rtol, atol):
    """Compare a gradient against the golden record (full / stats+sample / None)."""
    if f'gnone/{key}' in z.files:
        assert g is None or float(g.abs().max()) == 0.0, f'{key}: expected grad None'
        return
    assert g is not None, f'{key}: grad missing'
    g = g.detach().cpu().double().numpy()
    if f'g/{key}' in z.files:
        ref = z[f'g/{key}'].astype(np.float64)
        np.testing.assert_allclose(g, ref, rtol=rtol, atol=atol, err_msg=key)
    else:
        flat = g.reshape(-1)
        step = max(1, flat.size // 256)
        ref_s = z[f'gsample/{key}'].astype(np.float64)
        np.testing.assert_allclose(flat[::step][:256], ref_s, rtol=rtol, atol=atol, err_msg=key + ' (sample)')
        st = z[f'gstat/{key}']
        l2 = np.sqrt((flat ** 2).sum())
        assert abs(l2 - st[2]) <= rtol * st[2] + atol * np.sqrt(flat.size), f'{key}: l2 {l2} vs {st[2]}'


ENCDEC_CASES = ['encdec_concat_to_seq_post', 'encdec_concat_to_seq_pre', 'encdec_append_to_seq_post', 'encdec_append_to_seq_pre',
                'encdec_concat_to_qry_post', 'encdec_concat_to_qry_pre', 'encdec_mid_append_post', 'encdec_mid_qry_pre',
                'encdec_sketch_detr_post']


def encdec_case(name):
    """Regenerate (args, state_dict, inputs) of an enc/dec golden case (tests/golden/make_golden_encdec.py)."""
    from collections import OrderedDict
    z, meta = load_golden(name)
    args = syn.encdec_args(**meta['args'])
    shapes = OrderedDict(zip(str(z['keys']).split('\n'), [tuple(s) for s in json.loads(str(z['shapes']))]))
    sd = syn.synth_like(shapes, seed=1)
    inp = syn.synth_encdec_inputs(args, meta['B'], meta['L'], meta['Ls'], seed=1, pad=meta['pad'])
    return z, meta, args, sd, inp


def encdec_stack(out, head):
    """model output -> (logits, boxes) stacked like the fixtures: [n_dec,B,N,.] or, for sketch_detr, [n_dec,T,B,N,.]"""
    def st(o):
        layers = list(o.get('aux_outputs', [])) + [o]
        return torch.stack([l['pred_logits'] for l in layers]), torch.stack([l['pred_boxes'] for l in layers])
    if head == 'sketch_detr':
        per = [st(o) for o in out]
        return torch.stack([p[0] for p in per], dim=1), torch.stack([p[1] for p in per], dim=1)
    return st(out)


def encdec_att_view(att, z):
    """the slice of the attention-weight stack a fixture stores (make_golden_encdec.py)."""
    return att if att.numel() <= 65536 else att[:, :, ::7, ::37]
