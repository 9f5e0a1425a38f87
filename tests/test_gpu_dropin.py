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
