"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch functional ops on state-dict tensors) of the torchvision BasicBlock ResNet feature extractor
the reference builds at lib/modeling/backbone.py:133-152 (``nn.Sequential(*list(resnet34(...).children())[:-2])`` for frames,
``[:-1]`` — with the global average pool — for the sketch) and of ``ResNetBackbone.forward`` (backbone.py:72-89), eval-mode
BatchNorm.  torchvision is a third-party dependency that is NOT installed in this image (the reference's backbone module cannot
be imported); the published architecture (He et al. 2015; torchvision/models/resnet.py ``BasicBlock``:
conv3x3-bn-relu-conv3x3-bn, + identity or conv1x1(stride)-bn, relu; stem conv7x7 s2 p3 - bn - relu - maxpool 3x3 s2 p1) is
restated here and PINNED against an independent implementation of the same network, ``transformers.ResNetModel``
(layer_type='basic'), by tests/golden/make_golden_resnet.py.  Only ``tests/`` may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _bn(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], False, 0.0, eps)


def basic_block(x, sd, p, stride):
    out = F.relu(_bn(F.conv2d(x, sd[p + '.conv1.weight'], None, stride, 1), sd, p + '.bn1'))
    out = _bn(F.conv2d(out, sd[p + '.conv2.weight'], None, 1, 1), sd, p + '.bn2')
    if (p + '.downsample.0.weight') in sd:
        x = _bn(F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride, 0), sd, p + '.downsample.1')
    return F.relu(out + x)


def resnet_features(sd, depths, pixel_values, avgpool=False):
    """[n,3,H,W] -> feature map [n,C,h,w] (children[:-2]) or [n,C,1,1] (children[:-1])."""
    x = F.relu(_bn(F.conv2d(pixel_values, sd['0.weight'], None, 2, 3), sd, '1'))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(depths):
        for bi in range(n):
            x = basic_block(x, sd, f'{4 + li}.{bi}', (1 if li == 0 else 2) if bi == 0 else 1)
    return F.adaptive_avg_pool2d(x, 1) if avgpool else x


def resnet_backbone_forward(sd_video, depths_video, sd_sketch, depths_sketch, sketch_batch, video_batch):
    """ResNetBackbone.forward, backbone.py:72-89 -> (src_sketch [N,1,C], src_video [N,T*h*w,C])."""
    src_sketch = resnet_features(sd_sketch, depths_sketch, sketch_batch.squeeze(1), avgpool=True).squeeze(-1).squeeze(-1).unsqueeze(1)
    N, T = video_batch.shape[:2]
    v = resnet_features(sd_video, depths_video, video_batch.flatten(0, 1))
    v = v.reshape(N, T, *v.shape[1:]).transpose(1, 2).flatten(2, -1).transpose(1, 2)
    return src_sketch, v
