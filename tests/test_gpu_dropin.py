"""The drop-in boundary as a reference-style driver sees it: after ``svol_amd.install_as_lib()`` the
reference's own import lines resolve to this build, ``lib.configs`` parses a reference command line, ``build_model`` /
``build_loss`` take that namespace, the model is called with the reference's keyword names (svol_dataset.py:322-329) and the
train-step glue of train.py:208-234 runs unchanged."""
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_reference_style_train_step(monkeypatch):
    import svol_amd
    svol_amd.install_as_lib()
    # ---- what train.py does, with the reference's own import paths ----
    from lib import configs
    from lib.modeling.loss import build_loss
    from lib.modeling.model import build_model
    args = configs.parse_args(['--backbone', 'features', '--matcher', 'video_matcher', '--num_layers', '2', '--num_queries', '20',
                               '--num_frames', '4', '--hidden_dim', '64', '--nheads', '8', '--input_dropout', '0.4'])
    assert args.sketch_head == 'svanet' and args.aux_loss is True and args.set_cost_bbox == 5
    args.input_vid_dim = args.input_skch_dim = 32
    torch.manual_seed(args.seed)
    model = build_model(args).to('cuda')
    criterion = build_loss(args).to('cuda')
    assert hasattr(model, 'backbone') and hasattr(model, 'head')
    assert all(k.startswith(('head.', 'backbone.')) for k in model.state_dict())
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4)
    from svol_amd import synthetic as syn
    B, T, P = 2, 4, 16
    feats = syn.synth_inputs(args, B, T, P, seed=1)
    model_inputs = dict(src_sketch=feats['src_sketch'].cuda(), src_sketch_mask=torch.ones(B, 1).cuda(),
                        src_video=feats['src_video'].view(B, T, P, -1).cuda(), src_video_mask=torch.ones(B, T).cuda())
    targets = syn.synth_targets(B, T, seed=1)
    model.train()
    criterion.train()
    optimizer.zero_grad()
    outputs = model(**model_inputs)                                   # train.py:222
    loss_dict = criterion(outputs, targets)                           # train.py:223
    weight_dict = criterion.weight_dict
    losses = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)  # train.py:227-228
    losses.backward()
    optimizer.step()
    assert set(outputs.keys()) >= {'pred_logits', 'pred_boxes', 'aux_outputs'}
    assert outputs['pred_logits'].shape == (B, 20, 2) and outputs['pred_boxes'].shape == (B, 20, 4)
    assert sorted(k for k in loss_dict if not k[-1].isdigit()) == ['class_error', 'loss_bbox', 'loss_giou', 'loss_label']
    assert bool(torch.isfinite(losses)) and float(losses) > 0
    g = [p.grad for n, p in model.named_parameters() if 'class_head' not in n and 'sketch_video_cross_attn.out_proj' not in n]
    assert all(x is not None and bool(torch.isfinite(x).all()) for x in g)


def test_unknown_head_and_cpu_tensors_are_refused():
    import svol_amd
    svol_amd.install_as_lib()
    from lib import configs
    from lib.modeling.model import build_model
    args = configs.parse_args(['--backbone', 'features'])
    args.sketch_head = 'nope'
    with pytest.raises(NotImplementedError):   # model.py:38
        build_model(args)
    args.sketch_head = 'svanet'
    args.input_vid_dim = args.input_skch_dim = 32
    m = build_model(args)
    with pytest.raises(RuntimeError):          # no CPU path in the product
        m(src_sketch=torch.zeros(1, 1, 32), src_sketch_mask=torch.ones(1, 1), src_video=torch.zeros(1, 32, 4, 32),
          src_video_mask=torch.ones(1, 32))


def _crit_case(N=12, T=4, B=2, NL=2):
    from svol_amd import synthetic as syn
    from svol_amd.modeling.loss import build_loss
    args = syn.head_args(num_layers=NL, num_queries=N, num_frames=T, matcher='video_matcher')
    crit = build_loss(args).cuda()
    lg, bx = syn.synth_head_outputs(NL * B, N, seed=3)
    tg = syn.synth_targets(B, T, seed=2)
    return crit, lg.view(NL, B, N, 2).cuda(), bx.view(NL, B, N, 4).cuda(), tg


def _outputs(l_, b_):
    return {'pred_logits': l_[-1], 'pred_boxes': b_[-1],
            'aux_outputs': [{'pred_logits': l_[i], 'pred_boxes': b_[i]} for i in range(l_.shape[0] - 1)]}


def test_error_conventions_of_the_reference_are_kept():
    """where the reference raises on the host, the criterion stays asynchronous but (a) the layer's
    losses turn NaN and (b) the reference's exception is raised by the first call that synchronises anyway
    (matcher.forward / criterion.last_indices): AssertionError for degenerate boxes (box_utils.py:51-52), scipy's
    ValueError for NaN costs (matcher.py:93,158)."""
    crit, lg, bx, tg = _crit_case()
    ld = crit(_outputs(lg, bx), tg)
    assert all(bool(torch.isfinite(v)) for v in ld.values())
    crit.last_indices()                                         # healthy: nothing raised

    # (1) a prediction box with negative width in layer 0 only -> AssertionError; layer 0's losses NaN, layer 1 untouched
    bad = bx.clone()
    bad[0, 1, 3, 2] = -0.1
    lgr, badr = lg.clone().requires_grad_(True), bad.clone().requires_grad_(True)
    ld = crit(_outputs(lgr, badr), tg)
    assert all(not bool(torch.isfinite(ld[k + '_0'])) for k in ('loss_label', 'loss_bbox', 'loss_giou'))
    assert all(bool(torch.isfinite(ld[k])) for k in ('loss_label', 'loss_bbox', 'loss_giou'))
    # ... and the flagged layer's GRADIENTS are NaN too: backward() / optimizer.step() cannot proceed on a matching scipy
    # would have refused; the healthy layer's gradients stay finite
    sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict).backward()
    assert bool(torch.isnan(lgr.grad[0]).all()) and bool(torch.isnan(badr.grad[0]).all())
    assert bool(torch.isfinite(lgr.grad[1]).all()) and bool(torch.isfinite(badr.grad[1]).all())
    with pytest.raises(AssertionError):
        crit.last_indices()
    with pytest.raises(AssertionError):
        crit.matcher({'pred_logits': lg[0], 'pred_boxes': bad[0]}, tg)

    # (2) a degenerate TARGET box: every layer is flagged
    tg2 = [dict(t) for t in tg]
    fr = next(k for k, v in tg2[0]['bboxes'].items() if len(v))
    boxes = [dict(b) for b in tg2[0]['bboxes'][fr]]
    boxes[0]['bbox'] = boxes[0]['bbox'].clone()
    boxes[0]['bbox'][3] = -0.2
    tg2[0]['bboxes'] = dict(tg2[0]['bboxes'])
    tg2[0]['bboxes'][fr] = boxes
    ld = crit(_outputs(lg, bx), tg2)
    assert not bool(torch.isfinite(ld['loss_giou'])) and not bool(torch.isfinite(ld['loss_giou_0']))
    with pytest.raises(AssertionError):
        crit.last_indices()

    # (3) NaN logits (boxes fine) -> scipy's ValueError, NaN losses for that layer
    nl = lg.clone()
    nl[1, 0, 2, 0] = float('nan')
    ld = crit(_outputs(nl, bx), tg)
    assert not bool(torch.isfinite(ld['loss_label']))
    with pytest.raises(ValueError, match='invalid numeric entries'):
        crit.last_indices()

    # (4) NaN box coordinates: the reference trips over the box check first (AssertionError), not scipy
    nb = bx.clone()
    nb[1, 0, 0, 0] = float('nan')
    crit(_outputs(lg, nb), tg)
    with pytest.raises(AssertionError):
        crit.last_indices()
