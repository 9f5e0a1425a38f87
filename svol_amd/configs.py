"""Option surface of the SVOL drivers (mirror of the reference's lib/configs.py:8-177).

Same flag names, destinations, types and defaults (pinned by
tests/golden/configs_defaults.json, which was dumped from the reference), so a
reference command line parses unchanged.  Differences, all additive:

* the table below is data, parsed on demand — ``configs.args`` is resolved lazily
  on first attribute access (the reference parses at import time, configs.py:179);
* ``--compute_dtype {bf16,fp16,fp32}`` selects the kernel element type (the
  reference's only precision knob is apex ``--opt-level``; ``O0`` = fp32; fp16: the SVANet head only, BASELINE configs[4]);
* ``--backbone features`` feeds pre-extracted frame / sketch features straight
  to the head (the measured boundary);
* ``--enc_layers --dec_layers --mode --feat_dim`` and ``--sketch_head svanet_variants``: what the reference's enc/dec
  heads read but its parser never defines;
* ``--train_backbone {0,1}`` (``--backbone resnet``): default = the reference's behaviour (its training step optimises the
  backbone, train.py:72), with the reference's dead ``--freeze_backbone`` honoured as the opt-out; ``--sync_bn`` is honoured
  by the trainable extractors (global-batch BatchNorm statistics across ranks, train.py:65-68).
"""
from __future__ import annotations

import argparse
import sys

_T, _F = True, False
# (flags, kwargs) — grouped as in the reference file
_OPTIONS = [
    # meta
    (('--root',), dict(type=str, default='/mnt/server15_hard2/sangmin/data/svol/', help='dataset root')),
    (('--anno_root',), dict(type=str, default='/mnt/server15_hard2/sangmin/data/svol/annos/', help='annotation root')),
    (('--video_dataset',), dict(type=str, default='imagenet_vid')),
    (('--sketch_dataset',), dict(type=str, default='sketchy', choices=['sketchy', 'tu_berlin', 'quickdraw'])),
    (('--results_dir',), dict(type=str, default='results')),
    (('--seed',), dict(type=int, default=1, help='0 = do not fix the seed')),
    (('--log_interval',), dict(type=int, default=100)),
    (('--val_interval',), dict(type=int, default=1000)),
    (('--save_interval',), dict(type=int, default=-1, help='-1 disables periodic checkpoints')),
    (('--no_gpu',), dict(dest='use_gpu', action='store_false')),
    (('--debug',), dict(action='store_true')),
    (('--eval_untrained',), dict(action='store_true')),
    (('--log_dir',), dict(type=str, default='logs')),
    (('--checkpoint',), dict(type=str, default='./save')),
    (('--resume',), dict(type=str, default=None)),
    (('--resume_all',), dict(action='store_true')),
    (('--use_neptune',), dict(action='store_true', help='accepted for CLI compatibility; no Neptune integration')),
    # distributed
    (('--dist-backend',), dict(type=str, default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL on ROCm')),
    (('--use_amp',), dict(type=bool, default=True)),
    (('--sync_bn',), dict(action='store_true')),
    (('--channels-last',), dict(type=bool, default=False)),
    (('--opt-level',), dict(type=str, default='O0')),
    (('--keep-batchnorm-fp32',), dict(type=str, default=None)),
    (('--loss-scale',), dict(type=str, default=None)),
    # training
    (('--start_iter',), dict(type=int, default=None)),
    (('--num_iters',), dict(type=int, default=50000)),
    (('--early_stop_patience',), dict(type=int, default=10)),
    (('--lr',), dict(type=float, default=1e-4)),
    (('--lr_drop_step',), dict(type=int, default=20000)),
    (('--wd',), dict(type=float, default=1e-4)),
    (('--optimizer',), dict(type=str, default='adamw')),
    (('--scheduler',), dict(type=str, default='steplr')),
    (('--freeze_backbone',), dict(action='store_true')),
    (('--zeroshot_dataset_eval',), dict(action='store_true')),
    (('--zeroshot_category_eval',), dict(action='store_true')),
    (('--unified_sketch_dataset',), dict(action='store_true')),
    # data
    (('--bs',), dict(type=int, default=16)),
    (('--eval_bs',), dict(type=int, default=16)),
    (('--num_workers',), dict(type=int, default=4)),
    (('--no_pin_memory',), dict(dest='pin_memory', action='store_false')),
    (('--num_frames',), dict(type=int, default=32)),
    (('--num_input_sketches',), dict(type=int, default=1)),
    (('--tight_frame_sampling',), dict(action='store_true')),
    (('--aspect_ratio_grouping',), dict(type=bool, default=False)),
    # model
    (('--sketch_head',), dict(type=str, default='svanet', choices=['svanet', 'sketch_detr', 'svanet_variants'])),
    (('--backbone',), dict(type=str, default='vit', choices=['vit', 'resnet', 's3d', 'features'])),
    (('--hidden_dim',), dict(type=int, default=256)),
    (('--nheads',), dict(type=int, default=8)),
    (('--num_layers',), dict(type=int, default=4)),
    (('--num_queries',), dict(type=int, default=320)),
    (('--num_queries_per_frame',), dict(type=int, default=10)),
    (('--input_dropout',), dict(type=float, default=0.4)),
    (('--use_sketch_pos',), dict(type=bool, default=True)),
    (('--n_input_proj',), dict(type=int, default=2)),
    (('--dropout',), dict(type=float, default=0.1)),
    (('--dim_feedforward',), dict(type=int, default=1024, help='ignored by svanet (d_ff fixed to 2048)')),
    (('--pre_norm',), dict(action='store_true')),
    (('--sketch_position_embedding',), dict(type=str, default='sine', choices=['trainable', 'sine', 'learned'])),
    (('--video_position_embedding',), dict(type=str, default='sine', choices=['trainable', 'sine', 'learned'])),
    # loss
    (('--matcher',), dict(type=str, default='per_frame_matcher', choices=['per_frame_matcher', 'video_matcher'])),
    (('--set_cost_bbox',), dict(type=int, default=5)),
    (('--set_cost_giou',), dict(type=int, default=1)),
    (('--set_cost_class',), dict(type=int, default=2)),
    (('--no_aux_loss',), dict(dest='aux_loss', action='store_false')),
    (('--eos_coef',), dict(type=float, default=0.1)),
    # evaluation
    (('--bbox_type',), dict(type=str, default='cxcywh', choices=['cxcywh', 'xyxy'])),
    (('--no_sort_results',), dict(action='store_true')),
    # feature plots
    (('--vis_mode',), dict(type=str, default=None)),
    (('--use_vis_mean',), dict(action='store_true')),
    (('--n_neighbor',), dict(type=int, default=15)),
]

# additive, build-specific options (not part of the reference surface)
_EXTRA = [
    (('--compute_dtype',), dict(type=str, default='bf16', choices=['bf16', 'fp16', 'fp32'],
                                help='element type of the HIP kernels (fp32 accumulate either way)')),
    (('--input_vid_dim',), dict(type=int, default=512, help='feature width with --backbone features')),
    (('--input_skch_dim',), dict(type=int, default=512, help='feature width with --backbone features')),
    # the enc/dec Transformer heads (sketch_detr, svanet_variants) read these; the reference never defines them, which is
    # why its own --sketch_head sketch_detr cannot be built from its option surface
    (('--enc_layers',), dict(type=int, default=6)),
    (('--dec_layers',), dict(type=int, default=6)),
    (('--mode',), dict(type=str, default='append_to_seq', choices=['concat_to_seq', 'append_to_seq', 'concat_to_qry'],
                       help='how svanet_variants presents the sketch to the enc/dec Transformer')),
    (('--feat_dim',), dict(type=int, default=512, help='feature width of the enc/dec heads (one width for sketch and video)')),
    # --backbone resnet: the reference optimises the backbone too (train.py:72); None = "as the reference": trainable unless
    # --freeze_backbone (a flag the reference parses and never reads) is given; 0 / 1 force the frozen / trainable extractors
    (('--train_backbone',), dict(type=int, default=None, choices=[0, 1],
                                 help='ResNet extractors in training mode and in the optimiser (default: 1 unless --freeze_backbone)')),
]


def get_parser(extra: bool = True) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description='Sketch Localization Transformer (MI355X build)')
    for flags, kw in _OPTIONS + (_EXTRA if extra else []):
        p.add_argument(*flags, **kw)
    return p


def parse_args(argv=None, extra: bool = True) -> argparse.Namespace:
    return get_parser(extra).parse_args(argv)


def reference_defaults() -> dict:
    """Defaults of the reference's option surface only (no build extras)."""
    return vars(parse_args([], extra=False))


_ARGS = None


def __getattr__(name):  # lazy module-level ``args`` like the reference's configs.args
    global _ARGS
    if name == 'args':
        if _ARGS is None:
            _ARGS = parse_args(sys.argv[1:])
        return _ARGS
    raise AttributeError(name)
