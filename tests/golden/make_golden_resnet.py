#!/usr/bin/env python3
"""ResNet extractor fixtures from an INDEPENDENT implementation: ``transformers.ResNetModel`` (basic layers =
torchvision's BasicBlock network; torchvision itself is not installed here).  The deterministic synthetic weights of
``svol_amd.synthetic.synth_resnet_state_dict`` (torchvision key names) are renamed to the Hugging Face names, the model runs on
CPU fp32 in eval mode, and ONLY outputs are stored (tokens in the (h, w) order backbone.py:85-87 flattens to, pooled features).

    python tests/golden/make_golden_resnet.py
"""
import json
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from transformers import ResNetConfig, ResNetModel  # noqa: E402

from svol_amd import synthetic as syn  # noqa: E402

CASES = {
    'resnet_tiny': dict(depths=(1, 2), widths=(16, 32), stem=16, n=2, size=32),
    'resnet_tiny3': dict(depths=(2, 1, 1), widths=(16, 32, 64), stem=16, n=3, size=64),
    'resnet18_1img': dict(depths=(2, 2, 2, 2), widths=(64, 128, 256, 512), stem=64, n=1, size=224),
    'resnet34_1img': dict(depths=(3, 4, 6, 3), widths=(64, 128, 256, 512), stem=64, n=1, size=224),
}


def hf_key(k):
    if k == '0.weight':
        return 'embedder.embedder.convolution.weight'
    if k.startswith('1.'):
        return 'embedder.embedder.normalization.' + k[2:]
    m = re.match(r'(\d+)\.(\d+)\.(conv|bn)(\d)\.(.+)', k)
    if m:
        s, b, kind, j, rest = int(m.group(1)) - 4, m.group(2), m.group(3), int(m.group(4)) - 1, m.group(5)
        return f'encoder.stages.{s}.layers.{b}.layer.{j}.' + ('convolution.' if kind == 'conv' else 'normalization.') + rest
    m = re.match(r'(\d+)\.(\d+)\.downsample\.(\d)\.(.+)', k)
    s, b, j, rest = int(m.group(1)) - 4, m.group(2), m.group(3), m.group(4)
    return f'encoder.stages.{s}.layers.{b}.shortcut.' + ('convolution.' if j == '0' else 'normalization.') + rest


def run(name, c):
    shapes = syn.resnet_param_shapes(c['depths'], c['widths'], c['stem'])
    sd = syn.synth_resnet_state_dict(shapes, seed=1)
    cfg = ResNetConfig(num_channels=3, embedding_size=c['stem'], hidden_sizes=list(c['widths']), depths=list(c['depths']),
                       layer_type='basic', hidden_act='relu', downsample_in_first_stage=False)
    model = ResNetModel(cfg).eval()
    model.load_state_dict({hf_key(k): v for k, v in sd.items()}, strict=True)
    x = syn.synth_images(c['n'], syn.vit_config(image_size=c['size']), seed=1)
    with torch.no_grad():
        o = model(x)
    fmap = o.last_hidden_state
    rec = {'meta': np.asarray(json.dumps(dict(depths=list(c['depths']), widths=list(c['widths']), stem=c['stem'], n=c['n'],
                                              size=c['size'], torch=torch.__version__))),
           'tokens': fmap.flatten(2).transpose(1, 2).contiguous().numpy(), 'pooled': o.pooler_output.flatten(1).numpy()}
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **rec)
    print(f'{name}: tokens {tuple(rec["tokens"].shape)} |max| {np.abs(rec["tokens"]).max():.3f} -> {os.path.getsize(path) / 1024:.1f} KiB')


TRAIN_CASES = {   # training mode: tokens of the batch-statistics forward, parameter gradients of loss = sum(features * probe), running statistics
    'resnet_tiny_train': dict(depths=(1, 2), widths=(16, 32), stem=16, n=3, size=32),
    'resnet_tiny3_train': dict(depths=(2, 1, 1), widths=(16, 32, 64), stem=16, n=2, size=64),
}


def run_train(name, c):
    shapes = syn.resnet_param_shapes(c['depths'], c['widths'], c['stem'])
    sd = syn.synth_resnet_state_dict(shapes, seed=1)
    cfg = ResNetConfig(num_channels=3, embedding_size=c['stem'], hidden_sizes=list(c['widths']), depths=list(c['depths']),
                       layer_type='basic', hidden_act='relu', downsample_in_first_stage=False)
    model = ResNetModel(cfg).double().train()
    model.load_state_dict({hf_key(k): v.double() if v.is_floating_point() else v for k, v in sd.items()}, strict=True)
    x = syn.synth_images(c['n'], syn.vit_config(image_size=c['size']), seed=1).double()
    o = model(x)
    fmap = o.last_hidden_state
    g = torch.Generator().manual_seed(7)
    probe = torch.randn(fmap.shape, generator=g, dtype=torch.float64)
    (fmap * probe).sum().backward()
    inv = {hf_key(k): k for k in sd}
    rec = {'meta': np.asarray(json.dumps(dict(depths=list(c['depths']), widths=list(c['widths']), stem=c['stem'], n=c['n'],
                                              size=c['size'], torch=torch.__version__))),
           'features': fmap.detach().numpy(), 'probe': probe.numpy()}
    for hk, p_ in model.named_parameters():
        rec['grad/' + inv[hk]] = p_.grad.numpy()
    for hk, b_ in model.named_buffers():
        if hk in inv and 'running' in hk:
            rec['stat/' + inv[hk]] = b_.detach().numpy()
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **rec)
    print(f'{name}: features {tuple(fmap.shape)}, {sum(1 for k in rec if k.startswith("grad/"))} gradients -> {os.path.getsize(path) / 1024:.1f} KiB')


if __name__ == '__main__':
    for name, c in CASES.items():
        run(name, c)
    for name, c in TRAIN_CASES.items():
        run_train(name, c)
