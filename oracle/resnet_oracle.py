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


# ---- TRAINING mode (the reference's train.py:72 optimises every parameter of build_model(args); model.train() puts torchvision's
# BatchNorm2d into batch-statistics mode, backbone.py:133-152).  Same network, F.batch_norm(training=True) on CLONES of the running
# statistics (returned updated), gradients by torch autograd; pinned against transformers.ResNetModel.train() by
# tests/golden/make_golden_resnet.py (the *_train fixtures).
def _bn_train(x, sd, p, stats, eps=1e-5, momentum=0.1):
    rm, rv = sd[p + '.running_mean'].clone(), sd[p + '.running_var'].clone()
    y = F.batch_norm(x, rm, rv, sd[p + '.weight'], sd[p + '.bias'], True, momentum, eps)
    stats[p + '.running_mean'], stats[p + '.running_var'] = rm, rv
    return y


def resnet_features_train(sd, depths, pixel_values, avgpool=False):
    """-> (feature map [n,C,h,w] or pooled [n,C,1,1], {key: updated running statistic}); differentiable w.r.t. the tensors of sd."""
    stats = {}
    x = F.relu(_bn_train(F.conv2d(pixel_values, sd['0.weight'], None, 2, 3), sd, '1', stats))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(depths):
        for bi in range(n):
            p, stride = f'{4 + li}.{bi}', (1 if li == 0 else 2) if bi == 0 else 1
            out = F.relu(_bn_train(F.conv2d(x, sd[p + '.conv1.weight'], None, stride, 1), sd, p + '.bn1', stats))
            out = _bn_train(F.conv2d(out, sd[p + '.conv2.weight'], None, 1, 1), sd, p + '.bn2', stats)
            if (p + '.downsample.0.weight') in sd:
                x = _bn_train(F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride, 0), sd, p + '.downsample.1', stats)
            x = F.relu(out + x)
    return (F.adaptive_avg_pool2d(x, 1) if avgpool else x), stats


def resnet_train_grads(sd, depths, pixel_values, probe, avgpool=False, dtype=torch.float64):
    """loss = sum(features * probe) in `dtype`: -> (features, {parameter key: gradient}, updated running statistics)."""
    # (detach + clone: with dtype = the tensors' own, .to() would return the caller's tensors and requires_grad_ would mutate them)
    sdd = {k: (v.detach().clone().to(dtype).requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    f, stats = resnet_features_train(sdd, depths, pixel_values.to(dtype), avgpool)
    (f * probe.to(dtype)).sum().backward()
    return f.detach(), {k: v.grad for k, v in sdd.items() if torch.is_tensor(v) and v.requires_grad}, stats


def resnet_backbone_forward(sd_video, depths_video, sd_sketch, depths_sketch, sketch_batch, video_batch):
    """ResNetBackbone.forward, backbone.py:72-89 -> (src_sketch [N,1,C], src_video [N,T*h*w,C])."""
    src_sketch = resnet_features(sd_sketch, depths_sketch, sketch_batch.squeeze(1), avgpool=True).squeeze(-1).squeeze(-1).unsqueeze(1)
    N, T = video_batch.shape[:2]
    v = resnet_features(sd_video, depths_video, video_batch.flatten(0, 1))
    v = v.reshape(N, T, *v.shape[1:]).transpose(1, 2).flatten(2, -1).transpose(1, 2)
    return src_sketch, v
