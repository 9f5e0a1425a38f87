"""Deterministic synthetic workload generators for the SVOL hot path.

Everything here is numpy ``RandomState`` based (stable across numpy/torch
versions and across machines) so that the golden fixtures under
``tests/golden/`` — which were produced by feeding exactly these tensors to the
reference implementation — can be regenerated bit-for-bit on the GPU box
without shipping weights.

Workload definition follows SURVEY.md §8(d) / BASELINE.md §3: features ~ N(0,1)
at the head boundary ``SVANet.forward(src_sketch[B,1,Din], src_sketch_mask[B,1],
src_video[B,T*P,Din], src_video_mask[B,T*P])`` (reference svanet.py:65), targets
per frame m ~ U{0,1,2} boxes with at least one box in frame 0
(svol_dataset.py:272 guarantees >=1 box per video), cx,cy ~ U(.25,.75),
w,h ~ U(.05,.35), cxcywh.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

D_FF = 2048  # hard-coded in the reference (cross_modal_transformer.py:196-202)


def head_args(**over) -> SimpleNamespace:
    """Namespace with every field the head + criterion read."""
    a = dict(
        hidden_dim=256, nheads=8, num_layers=6, num_queries=100,
        num_queries_per_frame=10, num_frames=32,
        input_vid_dim=512, input_skch_dim=512, input_dropout=0.4,
        n_input_proj=2, aux_loss=True, use_sketch_pos=True, vis_mode=None,
        sketch_position_embedding='sine', video_position_embedding='sine',
        matcher='video_matcher', set_cost_bbox=5, set_cost_giou=1,
        set_cost_class=2, eos_coef=0.1, bbox_type='cxcywh',
        sketch_head='svanet',
    )
    a.update(over)
    return SimpleNamespace(**a)


# named workload configurations (BASELINE.json "configs")
def cfg1_args(matcher='video_matcher'):
    n = 10 if matcher == 'video_matcher' else 40
    return head_args(hidden_dim=64, nheads=8, num_layers=1, num_queries=n,
                     num_queries_per_frame=10, num_frames=4, matcher=matcher)


def cfg2_args(matcher='video_matcher'):
    n = 100 if matcher == 'video_matcher' else 320
    return head_args(hidden_dim=256, nheads=8, num_layers=6, num_queries=n,
                     num_queries_per_frame=10, num_frames=32, matcher=matcher)


CFG_SHAPES = {
    'cfg1': dict(B=1, T=4, P=49),
    'cfg2': dict(B=8, T=32, P=196),
    'cfg5': dict(B=1, T=128, P=256),
}


def head_param_shapes(args) -> "OrderedDict[str, tuple]":
    """Parameter names/shapes of the SVANet head, in the reference's
    ``state_dict()`` order (probed from the reference; pinned by
    tests/golden/*.npz ``keys``)."""
    d, nl = args.hidden_dim, args.num_layers
    out = OrderedDict()
    attn = ['sketch_video_cross_attn', 'content_self_attn', 'token_self_attn',
            'content_token_cross_attn']
    for i in range(nl):
        p = f'transformer.layers.{i}.'

        def mha(name):
            out[p + name + '.in_proj_weight'] = (3 * d, d)
            out[p + name + '.in_proj_bias'] = (3 * d,)
            out[p + name + '.out_proj.weight'] = (d, d)
            out[p + name + '.out_proj.bias'] = (d,)

        def norm(k):
            out[p + f'norm{k}.weight'] = (d,)
            out[p + f'norm{k}.bias'] = (d,)

        def mlp(k):
            out[p + f'mlp{k}.fc1.weight'] = (D_FF, d)
            out[p + f'mlp{k}.fc1.bias'] = (D_FF,)
            out[p + f'mlp{k}.fc2.weight'] = (d, D_FF)
            out[p + f'mlp{k}.fc2.bias'] = (d,)

        mha(attn[0]); norm(1)
        mha(attn[1]); norm(2)
        mlp(1); norm(3)
        mha(attn[2]); norm(4)
        mha(attn[3]); norm(5)
        mlp(2); norm(6)
    for j, (o, i_) in enumerate([(d, d), (d, d), (4, d)]):
        out[f'bbox_embed.layers.{j}.weight'] = (o, i_)
        out[f'bbox_embed.layers.{j}.bias'] = (o,)
    out['class_embed.weight'] = (2, d); out['class_embed.bias'] = (2,)
    out['class_head.weight'] = (2, d); out['class_head.bias'] = (2,)
    out['query_embed.weight'] = (args.num_queries, d)
    for name, din in (('input_video_proj', args.input_vid_dim),
                      ('input_sketch_proj', args.input_skch_dim)):
        for j in range(args.n_input_proj):
            i_ = din if j == 0 else d
            out[f'{name}.{j}.LayerNorm.weight'] = (i_,)
            out[f'{name}.{j}.LayerNorm.bias'] = (i_,)
            out[f'{name}.{j}.net.1.weight'] = (d, i_)
            out[f'{name}.{j}.net.1.bias'] = (d,)
    return out


def _rs(key: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_state_dict(args, seed: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic non-trivial fp32 weights for every head parameter.

    Matrices: Xavier-uniform range (what the reference re-initialises its
    transformer with, cross_modal_transformer.py:22-25); biases U(-.1,.1);
    LayerNorm weight 1+U(-.1,.1) so that every gradient path is exercised.
    """
    sd = OrderedDict()
    for k, shp in head_param_shapes(args).items():
        r = _rs(k, seed)
        if k == 'query_embed.weight':
            v = r.standard_normal(shp)
        elif len(shp) == 2:
            a = float(np.sqrt(6.0 / (shp[0] + shp[1])))
            v = r.uniform(-a, a, size=shp)
        elif ('norm' in k or 'LayerNorm' in k) and k.endswith('weight'):
            v = 1.0 + r.uniform(-0.1, 0.1, size=shp)
        else:
            v = r.uniform(-0.1, 0.1, size=shp)
        sd[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def synth_inputs(args, B: int, T: int, P: int, seed: int = 1, pad_frames: int = 0, pad_all: bool = False):
    """Head-boundary inputs. ``pad_frames``: number of trailing frames marked
    padding (mask 0) in every odd batch element — in EVERY element with ``pad_all`` — (exercises key_padding_mask)."""
    r = _rs('inputs', seed)
    L = T * P
    src_video = r.standard_normal((B, L, args.input_vid_dim)).astype(np.float32)
    src_sketch = r.standard_normal((B, 1, args.input_skch_dim)).astype(np.float32)
    vmask = np.ones((B, L), np.float32)
    if pad_frames:
        for b in range(B):
            if b % 2 == 1 or pad_all:
                vmask[b, (T - pad_frames) * P:] = 0.0
    smask = np.ones((B, 1), np.float32)
    return dict(src_sketch=torch.from_numpy(src_sketch),
                src_sketch_mask=torch.from_numpy(smask),
                src_video=torch.from_numpy(src_video),
                src_video_mask=torch.from_numpy(vmask))


def synth_targets(B: int, T: int, seed: int = 1, max_per_frame: int = 2):
    """``targets`` list in the reference's schema (svol_dataset.py:18-44):
    ``'bboxes': {frame_idx: [{'track_id', 'bbox': Tensor[4] cxcywh}]}``,
    ``'num_boxes_per_frame': list[T]``."""
    r = _rs('targets', seed)
    targets = []
    for b in range(B):
        bboxes = OrderedDict()
        nbf = []
        for t in range(T):
            m = int(r.randint(0, max_per_frame + 1))
            if t == 0 and m == 0:
                m = 1
            frame = []
            for j in range(m):
                cxcy = r.uniform(0.25, 0.75, size=2)
                wh = r.uniform(0.05, 0.35, size=2)
                box = np.concatenate([cxcy, wh]).astype(np.float32)
                frame.append({'track_id': j, 'bbox': torch.from_numpy(box)})
            bboxes[t * 3] = frame  # frame ids are arbitrary sampled indices
            nbf.append(m)
        targets.append({'video': f'synthetic_{b:04d}', 'size': [480, 360],
                        'sketch': f'sketch_{b:04d}', 'category': 'synthetic',
                        'total_boxes': int(sum(nbf)),
                        'num_boxes_per_frame': nbf, 'bboxes': bboxes})
    return targets


def synth_head_outputs(B: int, N: int, seed: int = 1):
    """Random (pred_logits, pred_boxes) for matcher/criterion-only goldens."""
    r = _rs('head_outputs', seed)
    logits = r.standard_normal((B, N, 2)).astype(np.float32)
    cxcy = r.uniform(0.1, 0.9, size=(B, N, 2))
    wh = r.uniform(0.02, 0.5, size=(B, N, 2))
    boxes = np.concatenate([cxcy, wh], -1).astype(np.float32)
    return torch.from_numpy(logits), torch.from_numpy(boxes)


def synth_eval_outputs(targets, N: int, T: int, seed: int = 1, noise: float = 0.08, tie_every: int = 0):
    """Head outputs that make the evaluation metrics non-trivial: the queries of chunk t (``chunk(num_frames)``
    order, test.py:148) sit near the ground-truth boxes of the t-th annotated frame (when it has any) with
    ``noise``-sized perturbations, the rest are random; ``tie_every`` > 0 duplicates every k-th logit pair so that
    equal scores exercise the stable sort.  Returns (pred_logits [B,N,2], pred_boxes [B,N,4]) fp32."""
    r = _rs('eval_outputs', seed)
    B = len(targets)
    chunk = -(-N // T)
    logits = r.standard_normal((B, N, 2)).astype(np.float32)
    cxcy = r.uniform(0.1, 0.9, size=(B, N, 2))
    wh = r.uniform(0.02, 0.5, size=(B, N, 2))
    boxes = np.concatenate([cxcy, wh], -1).astype(np.float32)
    for b, tg in enumerate(targets):
        frames = list(tg['bboxes'].keys())
        for c in range(-(-N // chunk)):
            if c >= len(frames):
                break
            gts = tg['bboxes'][frames[c]]
            for j in range(chunk):
                n = c * chunk + j
                if n >= N or not gts:
                    continue
                if r.uniform() < 0.7:
                    g = gts[int(r.randint(0, len(gts)))]['bbox'].numpy()
                    boxes[b, n] = np.clip(g + r.uniform(-noise, noise, size=4) * np.array([1, 1, 0.5, 0.5]), 0.01, 0.99)
                    logits[b, n, 0] += 1.5
    if tie_every:
        for n in range(tie_every, N, tie_every):
            logits[:, n] = logits[:, n - 1]
    return torch.from_numpy(logits), torch.from_numpy(boxes.astype(np.float32))


# ---- ViT feature extractor: HF ``ViTModel`` semantics, random weights ---------------------
def vit_config(**over) -> SimpleNamespace:
    """google/vit-base-patch16-224-in21k shape by default (backbone.py:118-122)."""
    a = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072, image_size=224,
             patch_size=16, num_channels=3, layer_norm_eps=1e-12)
    a.update(over)
    return SimpleNamespace(**a)


def vit_param_shapes(cfg) -> "OrderedDict[str, tuple]":
    """state-dict names / shapes of ``transformers.ViTModel(config, add_pooling_layer=False)`` (transformers 5.x)."""
    d, f, p, c = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_channels
    n_tok = (cfg.image_size // p) ** 2 + 1
    sh = OrderedDict()
    sh['embeddings.cls_token'] = (1, 1, d)
    sh['embeddings.position_embeddings'] = (1, n_tok, d)
    sh['embeddings.patch_embeddings.projection.weight'] = (d, c, p, p)
    sh['embeddings.patch_embeddings.projection.bias'] = (d,)
    for i in range(cfg.num_hidden_layers):
        pre = f'layers.{i}.'
        for nm in ('q_proj', 'k_proj', 'v_proj', 'o_proj'):
            sh[pre + f'attention.{nm}.weight'] = (d, d)
            sh[pre + f'attention.{nm}.bias'] = (d,)
        for nm in ('layernorm_before', 'layernorm_after'):
            sh[pre + nm + '.weight'] = (d,)
            sh[pre + nm + '.bias'] = (d,)
        sh[pre + 'mlp.fc1.weight'] = (f, d)
        sh[pre + 'mlp.fc1.bias'] = (f,)
        sh[pre + 'mlp.fc2.weight'] = (d, f)
        sh[pre + 'mlp.fc2.bias'] = (d,)
    sh['layernorm.weight'] = (d,)
    sh['layernorm.bias'] = (d,)
    return sh


def synth_vit_state_dict(cfg, seed: int = 1) -> "OrderedDict[str, torch.Tensor]":
    r = _rs('vit_weights', seed)
    sd = OrderedDict()
    for k, shp in vit_param_shapes(cfg).items():
        if 'layernorm' in k and k.endswith('weight'):
            v = 1.0 + 0.1 * r.standard_normal(shp)
        elif k.endswith('bias'):
            v = 0.05 * r.standard_normal(shp)
        elif k.endswith('weight') and len(shp) in (2, 4):
            fan_in = int(np.prod(shp[1:]))
            v = r.standard_normal(shp) / np.sqrt(fan_in)
        else:  # cls token, position embeddings
            v = 0.5 * r.standard_normal(shp)
        sd[k] = torch.from_numpy(v.astype(np.float32))
    return sd


def synth_images(n: int, cfg, seed: int = 1) -> torch.Tensor:
    """pixel_values [n, C, H, W] fp32 as the HF feature extractor hands them over (already normalised)."""
    r = _rs('images', seed)
    return torch.from_numpy(r.standard_normal((n, cfg.num_channels, cfg.image_size, cfg.image_size)).astype(np.float32))


# ----------------------------------------------------------------------------
# enc/dec Transformer heads: svanet_variants / sketch_detr
# ----------------------------------------------------------------------------
def encdec_args(**over) -> SimpleNamespace:
    """head_args + the fields build_transformer / the variant builders read (transformer.py:312-322,
    svanet_variants.py:289-306, sketch_detr.py:118-132) — absent from the reference's own option surface."""
    a = dict(enc_layers=2, dec_layers=2, dim_feedforward=64, dropout=0.1, pre_norm=False, mode='append_to_seq', feat_dim=32,
             backbone='features', hidden_dim=32, nheads=4, num_queries=8)
    a.update(over)
    return head_args(**a)


def synth_like(shapes, seed: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """synth_state_dict's recipe for an arbitrary ordered {key: shape} table (taken from a module's state_dict)."""
    sd = OrderedDict()
    for k, shp in shapes.items():
        shp = tuple(shp)
        r = _rs(k, seed)
        if k.endswith('query_embed.weight'):
            v = r.standard_normal(shp)
        elif len(shp) == 2:
            a = float(np.sqrt(6.0 / (shp[0] + shp[1])))
            v = r.uniform(-a, a, size=shp)
        elif ('norm' in k or 'LayerNorm' in k) and k.endswith('weight'):
            v = 1.0 + r.uniform(-0.1, 0.1, size=shp)
        else:
            v = r.uniform(-0.1, 0.1, size=shp)
        sd[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def synth_encdec_inputs(args, B: int, L: int, Ls: int = 1, seed: int = 1, pad: int = 0):
    """features at the variant heads' boundary: src_sketch [B,Ls,D], src_video [B,L,D]; the last `pad` video tokens of
    every odd batch element are padding (mask 0)."""
    r = _rs('encdec_inputs', seed)
    D = args.feat_dim
    src_video = r.standard_normal((B, L, D)).astype(np.float32)
    src_sketch = r.standard_normal((B, Ls, D)).astype(np.float32)
    vmask = np.ones((B, L), np.float32)
    for b in range(B):
        if pad and b % 2 == 1:
            vmask[b, L - pad:] = 0.0
    return dict(src_sketch=torch.from_numpy(src_sketch), src_sketch_mask=torch.from_numpy(np.ones((B, Ls), np.float32)),
                src_video=torch.from_numpy(src_video), src_video_mask=torch.from_numpy(vmask))


def synth_probe(shape, key: str, seed: int = 1) -> torch.Tensor:
    """fixed N(0,1) weights of the linear functional  sum(output * probe)  the gradient fixtures differentiate."""
    return torch.from_numpy(_rs('probe/' + key, seed).standard_normal(tuple(shape)).astype(np.float32))


# ----------------------------------------------------------------------------
# ResNet extractor
# ----------------------------------------------------------------------------
def resnet_param_shapes(depths=(2, 2, 2, 2), widths=(64, 128, 256, 512), stem=64) -> "OrderedDict[str, tuple]":
    """state-dict keys / shapes of ``nn.Sequential(*list(torchvision resnet.children())[:-2])`` (BasicBlock nets)."""
    sh = OrderedDict()

    def bn(p, c):
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            sh[f'{p}.{k}'] = (c,)
        sh[f'{p}.num_batches_tracked'] = ()

    sh['0.weight'] = (stem, 3, 7, 7)
    bn('1', stem)
    inp = stem
    for li, (n, planes) in enumerate(zip(depths, widths)):
        for bi in range(n):
            p = f'{4 + li}.{bi}'
            stride = (1 if li == 0 else 2) if bi == 0 else 1
            sh[p + '.conv1.weight'] = (planes, inp, 3, 3)
            bn(p + '.bn1', planes)
            sh[p + '.conv2.weight'] = (planes, planes, 3, 3)
            bn(p + '.bn2', planes)
            if stride != 1 or inp != planes:
                sh[p + '.downsample.0.weight'] = (planes, inp, 1, 1)
                bn(p + '.downsample.1', planes)
            inp = planes
    return sh


def synth_resnet_state_dict(shapes, seed: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """He-uniform convolutions, BatchNorm weight ~ 1 (0.5 on a block's second norm, which keeps the residual stream O(1)
    through 16 blocks), small shifts / running means, running variances in [0.8, 1.2]."""
    sd = OrderedDict()
    for k, shp in shapes.items():
        r = _rs('resnet/' + k, seed)
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.int64)
            continue
        if len(shp) == 4:
            a = float(np.sqrt(6.0 / (shp[1] * shp[2] * shp[3])))
            v = r.uniform(-a, a, size=shp)
        elif k.endswith('running_var'):
            v = r.uniform(0.8, 1.2, size=shp)
        elif k.endswith('running_mean') or k.endswith('bias'):
            v = r.uniform(-0.1, 0.1, size=shp)
        else:  # BatchNorm weight
            v = (0.5 if '.bn2.' in k else 1.0) + r.uniform(-0.1, 0.1, size=shp)
        sd[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd
