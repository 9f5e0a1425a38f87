"""Checkpoint files in the reference's schema (train.py:267-284) round-trip through the build's modules, including the apex-DDP
``module.`` prefix test.py strips (test.py:76-86) and a ResNet-backbone checkpoint loaded into a features-backbone model."""
import argparse

import torch

from svol_amd import synthetic as syn
from svol_amd.modeling.model import build_model
from svol_amd.utils import checkpoint as C


def _args():
    a = syn.head_args(hidden_dim=32, nheads=4, num_layers=2, num_queries=8, num_frames=4, input_vid_dim=32, input_skch_dim=32,
                      backbone='features', video_dataset='imagenet_vid', sketch_dataset='sketchy')
    return a


def test_name_pattern():
    assert C.checkpoint_name(_args(), 41) == '0041_model_imagenet_vid_sketchy_svanet_features_2l_4f_8q_5_1_2.ckpt'


def test_round_trip_with_ddp_prefix_and_foreign_backbone(tmp_path):
    args = _args()
    torch.manual_seed(1)
    m1 = build_model(args)
    opt = torch.optim.AdamW(m1.parameters(), lr=1e-4, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, 10)
    path = C.save_checkpoint(str(tmp_path / C.checkpoint_name(args, 7)), m1, opt, sched, 7, args, ddp_prefix=True)
    raw = torch.load(path, weights_only=False)
    assert sorted(raw.keys()) == ['amp', 'args', 'iter', 'lr_scheduler', 'model', 'optimizer']
    assert all(k.startswith('module.head.') for k in raw['model'])
    assert isinstance(raw['args'], argparse.Namespace) and raw['args'].num_queries == 8
    # a checkpoint of the reference additionally holds torchvision backbone weights
    raw['model']['module.backbone.video_backbone.0.weight'] = torch.zeros(64, 3, 7, 7)
    torch.save(raw, path)
    torch.manual_seed(2)
    m2 = build_model(args)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=5e-4)
    sched2 = torch.optim.lr_scheduler.StepLR(opt2, 10)
    ckpt, info = C.load_checkpoint(path, m2, opt2, sched2, resume_all=True)
    assert info['start_iter'] == 8 and info['set_aside'] == ['backbone.video_backbone.0.weight']
    assert info['amp'] == C.NEUTRAL_AMP_STATE
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert opt2.param_groups[0]['lr'] == 1e-4


def test_dropout_stream_position_survives_a_resume(tmp_path):
    """the enc/dec Transformer's stateless dropout masks are a function of (base seed + rank, training step, ...); the step
    counter is host state and used to restart from 0 after a resume.  It travels in the checkpoint (an entry the reference's drivers
    ignore) and the rank is folded into the seed."""
    from svol_amd import synthetic as syn
    from svol_amd.modeling import transformer as T
    from svol_amd.modeling.svanet_variants import build_svanet
    args = syn.encdec_args(dropout=0.1, input_dropout=0.0)
    torch.manual_seed(1)
    m1 = build_svanet(args)
    tr = [m for m in m1.modules() if isinstance(m, T.Transformer)][0]
    tr._drop_step, tr.drop_base_seed = 1234, 7
    path = C.save_checkpoint(str(tmp_path / 'x.ckpt'), m1, None, None, 3, args)
    raw = torch.load(path, weights_only=False)
    assert list(raw['svol_dropout'].values()) == [{'drop_base_seed': 7, 'drop_step': 1234}]
    m2 = build_svanet(args)
    C.load_checkpoint(path, m2, resume_all=True)
    tr2 = [m for m in m2.modules() if isinstance(m, T.Transformer)][0]
    assert (tr2._drop_step, tr2.drop_base_seed) == (1234, 7)
    import os
    old = os.environ.get('RANK')
    try:
        os.environ['RANK'] = '3'
        assert T._rank() == 3
    finally:
        if old is None:
            os.environ.pop('RANK')
        else:
            os.environ['RANK'] = old
