"""Model assembly (reference lib/modeling/model.py:8-44): backbone -> mask expansion -> head.

``build_model(args)`` / ``SketchLocalizationModel.forward(src_sketch, src_video, src_sketch_mask,
src_video_mask)`` keep the reference's signatures and the ``backbone.`` / ``head.`` state-dict
prefixes.  ``--backbone features`` plugs pre-extracted features in at the measured boundary;
``--backbone vit`` runs the ViT-B/16 extractor of ``backbone.py`` on the device for every frame and the sketch
(frozen); ``--backbone resnet`` the ResNet-34 / ResNet-18 extractors of ``resnet.py`` (f4), frozen by default,
trained with the head when ``args.train_backbone`` is set (what the reference's train.py does).
"""
from __future__ import annotations

import torch
from torch import nn

from .svanet import build_svanet


class FeatureBackbone(nn.Module):
    """Identity feature 'backbone': src_sketch [B,Ls,Dskch], src_video [B,T,P,Dvid] (or [B,T*P,Dvid])
    are already features.  Returns (sketch [B,Ls,D], video [B,T*P,D]) like backbone.py:72-89."""

    def forward(self, src_sketch, src_video):
        if src_video.dim() == 4:
            src_video = src_video.flatten(1, 2)
        return src_sketch, src_video


def build_backbone(args):
    """Like backbone.py:116-152 this MUTATES args (input_vid_dim / input_skch_dim)."""
    if args.backbone == 'features':
        args.input_vid_dim = getattr(args, 'input_vid_dim', 512)
        args.input_skch_dim = getattr(args, 'input_skch_dim', 512)
        return FeatureBackbone()
    if 'vit' in args.backbone:  # backbone.py:117-132: ViT-B/16 for frames and sketch, 768-wide features
        from .backbone import ViTBackbone, ViTExtractor, vit_base_config
        args.input_vid_dim = 768
        args.input_skch_dim = 768
        # the pretrained google/vit-base-patch16-224-in21k weights are loaded by the caller
        # (ViTExtractor.load_hf_state_dict); nothing is downloaded here
        return ViTBackbone(ViTExtractor(vit_base_config()), ViTExtractor(vit_base_config()))
    if 'resnet' in args.backbone:  # backbone.py:133-152: ResNet-34 on the frames (7x7 tokens), ResNet-18 + avgpool on the sketch
        from .resnet import ResNetBackbone, resnet18, resnet34
        args.input_vid_dim = 512
        args.input_skch_dim = 512
        cd = getattr(args, 'compute_dtype', 'bf16')
        # the torchvision IMAGENET1K_V1 weights are loaded by the caller (load_state_dict with the reference's backbone.* keys) —
        # nothing is downloaded here.  The reference TRAINS its backbone: train.py:72 hands every parameter of build_model(args) to
        # the optimiser and model.train() puts BatchNorm into batch-statistics mode (its --freeze_backbone flag is parsed and never
        # read).  So a reference-style run gets the trainable extractors (csrc/resnet_train.hip) by default; --freeze_backbone — dead
        # in the reference, honoured here — or --train_backbone 0 selects the frozen, BatchNorm-folded extractors (inference, or the
        # features-are-fixed setting bench.py --workload resnet measures without --train-backbone).
        tb = getattr(args, 'train_backbone', None)
        tb = (not bool(getattr(args, 'freeze_backbone', False))) if tb is None else bool(tb)
        sync = bool(getattr(args, 'sync_bn', False))   # train.py:65-68: apex convert_syncbn_model
        return ResNetBackbone(resnet34(compute_dtype=cd, trainable=tb, sync_bn=sync),
                              resnet18(avgpool=True, compute_dtype=cd, trainable=tb, sync_bn=sync))
    raise NotImplementedError(f"backbone '{args.backbone}' is not part of the MI355X build (the reference has it commented out)")


class SketchLocalizationModel(nn.Module):
    def __init__(self, backbone, head):
        super().__init__()
        self.backbone = backbone
        self.head = head

    def forward(self, src_sketch, src_video, src_sketch_mask=None, src_video_mask=None):
        T = src_video.shape[1]
        src_sketch, src_video = self.backbone(src_sketch, src_video)
        # per-token masks: one sketch-mask entry per sketch token, one frame-mask entry per patch (model.py:21-22)
        src_sketch_mask = src_sketch_mask.repeat_interleave(src_sketch.shape[1], dim=1)
        src_video_mask = src_video_mask.repeat_interleave(src_video.shape[1] // T, dim=1)
        return self.head(src_sketch, src_sketch_mask, src_video, src_video_mask)


def build_model(args):
    backbone = build_backbone(args)
    if args.sketch_head == 'svanet':
        head = build_svanet(args)
    elif args.sketch_head == 'sketch_detr':  # model.py:35-36; needs the build's --enc_layers/--dec_layers/--mode/--feat_dim
        from .sketch_detr import build_sketchdetr
        head = build_sketchdetr(args)
    elif args.sketch_head == 'svanet_variants':  # the reference's svanet_variants.py has no dispatch entry of its own
        from .svanet_variants import build_svanet as build_svanet_variants
        head = build_svanet_variants(args)
    else:
        raise NotImplementedError
    return SketchLocalizationModel(backbone, head)
