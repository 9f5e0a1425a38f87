#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  It imports the reference's head + criterion
(lib.modeling.svanet / lib.modeling.loss — SURVEY.md §8c), injects the single
missing third-party symbol ``torchvision.ops.boxes.box_area`` (exact
torchvision definition), loads the deterministic synthetic weights from
``svol_amd.synthetic`` into the reference modules, runs forward + criterion +
backward on CPU fp32, and stores ONLY DATA (outputs / indices / losses /
gradients) as small .npz files.  Inputs and weights are not stored: they are
regenerated bit-for-bit from ``svol_amd.synthetic`` by the tests.

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = os.environ.get('SVOL_REFERENCE', '/root/reference')
sys.path.insert(0, REF)

# --- the one missing third-party symbol (box_utils.py:6) --------------------
_tv = types.ModuleType('torchvision')
_ops = types.ModuleType('torchvision.ops')
_boxes = types.ModuleType('torchvision.ops.boxes')
_boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
_tv.ops = _ops
_ops.boxes = _boxes
sys.modules.setdefault('torchvision', _tv)
sys.modules.setdefault('torchvision.ops', _ops)
sys.modules.setdefault('torchvision.ops.boxes', _boxes)

from lib.modeling.svanet import build_svanet  # noqa: E402  (reference)
from lib.modeling.loss import build_loss  # noqa: E402  (reference)
from lib.modeling.matcher import build_matcher  # noqa: E402  (reference)

from svol_amd import synthetic as syn  # noqa: E402

FULL_GRAD_MAX = 4096
SAMPLE = 256

HEAD_CASES = {
    # name: (args overrides, B, T, P, pad_frames[, pad_all])
    'tiny_video': (dict(hidden_dim=32, nheads=4, num_layers=2, num_queries=8, num_queries_per_frame=2,
                        num_frames=4, input_vid_dim=32, input_skch_dim=32, matcher='video_matcher'), 2, 4, 6, 1),
    'tiny_frame': (dict(hidden_dim=32, nheads=4, num_layers=2, num_queries=8, num_queries_per_frame=2,
                        num_frames=4, input_vid_dim=32, input_skch_dim=32, matcher='per_frame_matcher'), 2, 4, 6, 1),
    'cfg1_video': (dict(hidden_dim=64, nheads=8, num_layers=1, num_queries=10, num_queries_per_frame=10,
                        num_frames=4, matcher='video_matcher'), 1, 4, 49, 0),
    'cfg1_frame': (dict(hidden_dim=64, nheads=8, num_layers=1, num_queries=40, num_queries_per_frame=10,
                        num_frames=4, matcher='per_frame_matcher'), 1, 4, 49, 0),
    'mid_video': (dict(hidden_dim=128, nheads=8, num_layers=3, num_queries=16, num_queries_per_frame=2,
                       num_frames=8, input_vid_dim=64, input_skch_dim=48, matcher='video_matcher'), 2, 8, 16, 2),
    'mid32_video': (dict(hidden_dim=256, nheads=8, num_layers=2, num_queries=100, num_queries_per_frame=10,
                         num_frames=8, input_vid_dim=64, input_skch_dim=64, matcher='video_matcher'), 2, 8, 24, 2),
    # BASELINE configs[1] at full depth / width / sequence length, one video (the reference materialises 1.26 GB of attention
    # weights per layer for it: ~15 GB resident, a minute on 8 cores): value-level parity at the shapes the bench launches
    'cfg2_b1_video': (dict(hidden_dim=256, nheads=8, num_layers=6, num_queries=100, num_queries_per_frame=10,
                           num_frames=32, matcher='video_matcher'), 1, 32, 196, 0),
    # the same with the last 8 of the 32 frames padded: 4704 valid keys = 36.75 key tiles, the key-padding mask at L = 6272
    'cfg2_b1_video_pad': (dict(hidden_dim=256, nheads=8, num_layers=6, num_queries=100, num_queries_per_frame=10,
                               num_frames=32, matcher='video_matcher'), 1, 32, 196, 8, True),
    # the shipped recipe's matcher scale THROUGH THE HEAD (train_sketchy.sh:21-23: 32 frames x 10 queries per frame = 320 queries,
    # per_frame_matcher) at the benchmark depth / width; P = 49 (the ResNet-34 map) keeps the reference's L x L tensors small
    'cfg2w_b1_frame': (dict(hidden_dim=256, nheads=8, num_layers=6, num_queries=320, num_queries_per_frame=10,
                            num_frames=32, matcher='per_frame_matcher'), 1, 32, 49, 4, True),
}

CRIT_CASES = {
    # name: (args overrides, B, T, N, max_per_frame)
    'crit_video_B8_N100_T32': (dict(num_layers=1, num_queries=100, num_frames=32, matcher='video_matcher'), 8, 32, 100, 2),
    'crit_frame_B8_N320_T32': (dict(num_layers=1, num_queries=320, num_queries_per_frame=10, num_frames=32,
                                    matcher='per_frame_matcher'), 8, 32, 320, 2),
    'crit_frame_B3_N8_T4_over': (dict(num_layers=1, num_queries=8, num_queries_per_frame=2, num_frames=4,
                                      matcher='per_frame_matcher'), 3, 4, 8, 3),
    'crit_video_B2_N4_T4_tall': (dict(num_layers=1, num_queries=4, num_frames=4, matcher='video_matcher'), 2, 4, 4, 3),
}


def pack_indices(indices):
    """list[(pred_idx, tgt_idx)] per video -> flat int64 arrays + offsets."""
    offs = [0]
    pi, ti = [], []
    for p, t in indices:
        pi.append(np.asarray(p, dtype=np.int64))
        ti.append(np.asarray(t, dtype=np.int64))
        offs.append(offs[-1] + len(pi[-1]))
    return (np.concatenate(pi) if pi else np.zeros(0, np.int64),
            np.concatenate(ti) if ti else np.zeros(0, np.int64),
            np.asarray(offs, dtype=np.int64))


def grad_record(out, key, g):
    if g is None:
        out[f'gnone/{key}'] = np.asarray(1, np.int8)
        return
    g = g.detach().cpu().numpy().astype(np.float32)
    if g.size <= FULL_GRAD_MAX:
        out[f'g/{key}'] = g
    else:
        flat = g.reshape(-1).astype(np.float64)
        step = max(1, flat.size // SAMPLE)
        out[f'gstat/{key}'] = np.asarray([flat.sum(), np.abs(flat).sum(), np.sqrt((flat ** 2).sum())], np.float64)
        out[f'gsample/{key}'] = flat[::step][:SAMPLE].astype(np.float32)


def run_head_case(name, over, B, T, P, pad, pad_all=False):
    args = syn.head_args(**over)
    torch.manual_seed(1)
    model = build_svanet(args)
    ref_keys = list(model.state_dict().keys())
    sd = syn.synth_state_dict(args, seed=1)
    assert ref_keys == list(sd.keys()), 'synthetic key order != reference state_dict order'
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    model.load_state_dict(sd, strict=True)
    model.eval()  # input_dropout off
    criterion = build_loss(args)
    criterion.eval()

    inp = syn.synth_inputs(args, B, T, P, seed=1, pad_frames=pad, pad_all=pad_all)
    targets = syn.synth_targets(B, T, seed=1)
    outputs = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
    loss_dict = criterion(outputs, targets)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)  # train.py:227-228
    total.backward()

    rec = {}
    rec['meta'] = np.asarray(json.dumps(dict(args=vars(args), B=B, T=T, P=P, pad_frames=pad, pad_all=pad_all,
                                             torch=torch.__version__)))
    rec['keys'] = np.asarray('\n'.join(ref_keys))
    rec['pred_logits'] = outputs['pred_logits'].detach().numpy()
    rec['pred_boxes'] = outputs['pred_boxes'].detach().numpy()
    if 'aux_outputs' in outputs and len(outputs['aux_outputs']):
        rec['aux_logits'] = np.stack([a['pred_logits'].detach().numpy() for a in outputs['aux_outputs']])
        rec['aux_boxes'] = np.stack([a['pred_boxes'].detach().numpy() for a in outputs['aux_outputs']])
    names = sorted(loss_dict.keys())
    rec['loss_names'] = np.asarray('\n'.join(names))
    rec['loss_values'] = np.asarray([float(loss_dict[k]) for k in names], np.float64)
    rec['weight_dict'] = np.asarray(json.dumps(wd))
    rec['loss_total'] = np.asarray(float(total), np.float64)
    # matcher indices per layer (last layer first, then aux 0..)
    with torch.no_grad():
        layers = [{'pred_logits': outputs['pred_logits'], 'pred_boxes': outputs['pred_boxes']}]
        layers += list(outputs.get('aux_outputs', []))
        for li, lo in enumerate(layers):
            p, t, o = pack_indices(criterion.matcher(lo, targets))
            tag = 'last' if li == 0 else f'aux{li - 1}'
            rec[f'idx/{tag}/pred'] = p
            rec[f'idx/{tag}/tgt'] = t
            rec[f'idx/{tag}/offs'] = o
    for k, p in model.named_parameters():
        grad_record(rec, k, p.grad)
    path = os.path.join(HERE, f'head_{name}.npz')
    np.savez_compressed(path, **rec)
    print(f'{name}: loss_total={float(total):.6f}  -> {os.path.getsize(path) / 1024:.1f} KiB')


# TRAINING mode of the primary path (svanet.py:168-171: input_dropout inside LinearLayer is the only dropout).  The
# keep masks nn.Dropout drew are RECORDED (forward hooks on the reference's own Dropout modules) and stored as packed bits, so
# that the oracle's placement of the dropout can be pinned against the reference whatever RNG produced the masks.
TRAIN_CASES = {
    # name: (args overrides, B, T, P, pad_frames)
    'train_cfg1_video': (dict(hidden_dim=64, nheads=8, num_layers=1, num_queries=10, num_queries_per_frame=10,
                              num_frames=4, matcher='video_matcher', input_dropout=0.4), 1, 4, 49, 0),
    'train_mid32_video': (dict(hidden_dim=256, nheads=8, num_layers=2, num_queries=100, num_queries_per_frame=10,
                               num_frames=8, input_vid_dim=64, input_skch_dim=64, matcher='video_matcher',
                               input_dropout=0.4), 2, 8, 24, 2),
}


def run_train_case(name, over, B, T, P, pad):
    args = syn.head_args(**over)
    torch.manual_seed(1)
    model = build_svanet(args)
    sd = syn.synth_state_dict(args, seed=1)
    model.load_state_dict(sd, strict=True)
    model.train()
    criterion = build_loss(args)
    criterion.train()
    masks = {}

    def hook(tag):
        def h(_m, a, out):
            assert bool((a[0] != 0).all()), 'a LayerNorm output is exactly 0: the keep mask cannot be read off the output'
            masks[tag] = (out != 0).numpy()
        return h

    for which, seq in (('video', model.input_video_proj), ('sketch', model.input_sketch_proj)):
        for j, layer in enumerate(seq):
            layer.net[0].register_forward_hook(hook(f'{which}/{j}'))
    inp = syn.synth_inputs(args, B, T, P, seed=1, pad_frames=pad)
    targets = syn.synth_targets(B, T, seed=1)
    torch.manual_seed(7)
    outputs = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
    loss_dict = criterion(outputs, targets)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)
    total.backward()
    rec = {}
    rec['meta'] = np.asarray(json.dumps(dict(args=vars(args), B=B, T=T, P=P, pad_frames=pad, pad_all=False, torch=torch.__version__)))
    rec['keys'] = np.asarray('\n'.join(model.state_dict().keys()))
    for tag, m in masks.items():
        rec[f'mask/{tag}/shape'] = np.asarray(m.shape, np.int64)
        rec[f'mask/{tag}/bits'] = np.packbits(m.reshape(-1))
        keep = float(m.mean())
        assert abs(keep - (1 - args.input_dropout)) < 0.1, (tag, keep)
    rec['pred_logits'] = outputs['pred_logits'].detach().numpy()
    rec['pred_boxes'] = outputs['pred_boxes'].detach().numpy()
    if 'aux_outputs' in outputs and len(outputs['aux_outputs']):
        rec['aux_logits'] = np.stack([a['pred_logits'].detach().numpy() for a in outputs['aux_outputs']])
        rec['aux_boxes'] = np.stack([a['pred_boxes'].detach().numpy() for a in outputs['aux_outputs']])
    names = sorted(loss_dict.keys())
    rec['loss_names'] = np.asarray('\n'.join(names))
    rec['loss_values'] = np.asarray([float(loss_dict[k]) for k in names], np.float64)
    rec['loss_total'] = np.asarray(float(total), np.float64)
    with torch.no_grad():
        layers = [{'pred_logits': outputs['pred_logits'], 'pred_boxes': outputs['pred_boxes']}]
        layers += list(outputs.get('aux_outputs', []))
        for li, lo in enumerate(layers):
            p, t, o = pack_indices(criterion.matcher(lo, targets))
            tag = 'last' if li == 0 else f'aux{li - 1}'
            rec[f'idx/{tag}/pred'], rec[f'idx/{tag}/tgt'], rec[f'idx/{tag}/offs'] = p, t, o
    for k, p in model.named_parameters():
        grad_record(rec, k, p.grad)
    path = os.path.join(HERE, f'head_{name}.npz')
    np.savez_compressed(path, **rec)
    print(f'{name}: loss_total={float(total):.6f}  -> {os.path.getsize(path) / 1024:.1f} KiB')


def run_crit_case(name, over, B, T, N, mpf):
    args = syn.head_args(**over)
    criterion = build_loss(args)
    criterion.eval()
    logits, boxes = syn.synth_head_outputs(B, N, seed=1)
    logits.requires_grad_(True)
    boxes.requires_grad_(True)
    targets = syn.synth_targets(B, T, seed=1, max_per_frame=mpf)
    outputs = {'pred_logits': logits, 'pred_boxes': boxes}
    loss_dict = criterion(outputs, targets)
    wd = criterion.weight_dict
    total = sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)
    total.backward()
    rec = {}
    rec['meta'] = np.asarray(json.dumps(dict(args=vars(args), B=B, T=T, N=N, max_per_frame=mpf,
                                             torch=torch.__version__)))
    names = sorted(loss_dict.keys())
    rec['loss_names'] = np.asarray('\n'.join(names))
    rec['loss_values'] = np.asarray([float(loss_dict[k]) for k in names], np.float64)
    rec['loss_total'] = np.asarray(float(total), np.float64)
    p, t, o = pack_indices(criterion.matcher(outputs, targets))
    rec['idx/pred'], rec['idx/tgt'], rec['idx/offs'] = p, t, o
    rec['g_logits'] = logits.grad.numpy()
    rec['g_boxes'] = boxes.grad.numpy()
    path = os.path.join(HERE, f'{name}.npz')
    np.savez_compressed(path, **rec)
    print(f'{name}: loss_total={float(total):.6f} matched={len(p)} -> {os.path.getsize(path) / 1024:.1f} KiB')


def run_lsap():
    """Known-answer vectors from scipy.optimize.linear_sum_assignment (the
    third-party solver the reference calls at matcher.py:93,158)."""
    import scipy
    from scipy.optimize import linear_sum_assignment
    rng = np.random.RandomState(7)
    rec = {'scipy_version': np.asarray(scipy.__version__)}
    shapes = [(4, 2), (2, 4), (3, 3), (10, 3), (3, 10), (100, 37), (37, 100), (10, 1), (1, 10), (10, 0),
              (64, 64), (100, 64), (5, 2), (320, 45)]
    cases = []
    for (nr, nc) in shapes:
        cases.append(('rand32', rng.random_sample((nr, nc)).astype(np.float32)))
        cases.append(('ties', rng.randint(0, 3, size=(nr, nc)).astype(np.float32)))
        cases.append(('zeros', np.zeros((nr, nc), np.float32)))
    cases.append(('neg', -rng.random_sample((12, 5)).astype(np.float32)))
    cases.append(('inf', np.array([[np.inf, 1.0], [2.0, np.inf], [0.5, 0.25]], np.float32)))
    cases.append(('f64', rng.standard_normal((20, 9))))
    for i, (kind, c) in enumerate(cases):
        r, cc = linear_sum_assignment(c)
        rec[f'c{i}/kind'] = np.asarray(kind)
        rec[f'c{i}/cost'] = c
        rec[f'c{i}/rows'] = r.astype(np.int64)
        rec[f'c{i}/cols'] = cc.astype(np.int64)
    rec['n'] = np.asarray(len(cases))
    path = os.path.join(HERE, 'lsap_known_answers.npz')
    np.savez_compressed(path, **rec)
    print(f'lsap: {len(cases)} cases -> {os.path.getsize(path) / 1024:.1f} KiB')


def run_configs():
    """Default option surface of lib/configs.py (parsed at import, configs.py:179)."""
    import io
    import contextlib
    argv = sys.argv
    sys.argv = ['x']
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import lib.configs as rc
    finally:
        sys.argv = argv
    d = vars(rc.args)
    with open(os.path.join(HERE, 'configs_defaults.json'), 'w') as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print(f'configs: {len(d)} options')


def run_posenc():
    from lib.modeling.position_encoding import PositionEmbeddingSine
    pe = PositionEmbeddingSine(32, normalize=True)
    mask = torch.ones(2, 12, dtype=torch.bool)
    mask[1, 8:] = False
    out = pe(torch.zeros(2, 12, 32), mask)
    np.savez_compressed(os.path.join(HERE, 'posenc_sine.npz'), mask=mask.numpy(), pos=out.numpy())
    print('posenc: done')


if __name__ == '__main__':
    torch.set_num_threads(int(os.environ.get('SVOL_GOLDEN_THREADS', '4')))
    torch.use_deterministic_algorithms(False)
    only = sys.argv[1:]   # optional: names of head cases to (re)generate; default = everything
    for n, c in HEAD_CASES.items():
        if not only or n in only:
            run_head_case(n, *c)
    for n, c in TRAIN_CASES.items():
        if not only or n in only:
            run_train_case(n, *c)
    if not only:
        for n, c in CRIT_CASES.items():
            run_crit_case(n, *c)
        run_lsap()
        run_posenc()
        run_configs()
