"""torchvision-style BasicBlock ResNet feature extractor (ResNet-18 / ResNet-34) on the MI355X kernels — SURVEY.md §8 f4.

The reference builds ``nn.Sequential(*list(resnet34(...).children())[:-2])`` for the frames and
``nn.Sequential(*list(resnet18(...).children())[:-1])`` for the sketch (backbone.py:133-152) and turns the frame feature
maps into tokens (backbone.py:72-89).  This module keeps that Sequential's child indices — ``0`` conv1, ``1`` bn1, ``4``-``7``
layer1-4 — so the ``backbone.{video,sketch}_backbone.*`` keys of a reference checkpoint load unchanged, and uses
``nn.Conv2d`` / ``nn.BatchNorm2d`` purely as parameter containers.  Arithmetic (svol_amd/csrc/resnet.hip + the GEMMs):

  conv + eval-mode BatchNorm (+ identity) (+ ReLU)  =  svol_conv_nhwc (implicit GEMM: the tile kernel gathers its A operand from
  the NHWC activation, no im2col matrix; the stem and odd widths: svol_im2col -> svol_gemm_nt) with the BatchNorm scale folded into the
  weights ([Cout, kh*kw*Cin], (ky,kx,c) order, K padded to 32), the BatchNorm shift as bias and the ReLU — for a block's second
  convolution the ReLU AFTER the identity add (SVOL_ACT_RELU_RES) — in the GEMM epilogue; 3x3 s2 max pooling and the sketch
  branch's global average pooling are their own kernels; activations are NHWC from the stem to the tokens.

Inference / frozen only (BatchNorm uses its running statistics, no backward): the pretrained torchvision weights cannot be
downloaded here and a trainable backbone needs batch-statistics BatchNorm and convolution gradients, neither of which is built.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops

_DTYPES = {'bf16': torch.bfloat16, 'fp32': torch.float32, torch.bfloat16: torch.bfloat16, torch.float32: torch.float32}


class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock parameter layout: conv1, bn1, conv2, bn2, downsample.{0,1}."""

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d, dtype):
    """(W' [Cout, Kp] in `dtype`, shift [Cout] fp32): eval-mode BatchNorm folded into the convolution, K = kh*kw*Cin in
    (ky, kx, c) order, zero-padded to a multiple of 32."""
    w = conv.weight.detach().float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    w = (w * scale[:, None, None, None]).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    K = w.shape[1]
    Kp = (K + 31) // 32 * 32
    if Kp != K:
        w = torch.cat([w, torch.zeros((w.shape[0], Kp - K), dtype=w.dtype, device=w.device)], dim=1)
    return w.to(dtype).contiguous(), shift.contiguous()


class ResNetExtractor(nn.Module):
    def __init__(self, depths=(2, 2, 2, 2), widths=(64, 128, 256, 512), stem=64, avgpool=False, compute_dtype='bf16'):
        super().__init__()
        self.add_module('0', nn.Conv2d(3, stem, 7, 2, 3, bias=False))
        self.add_module('1', nn.BatchNorm2d(stem))
        inplanes = stem
        for li, (n, planes) in enumerate(zip(depths, widths)):
            blocks = []
            for bi in range(n):
                blocks.append(BasicBlock(inplanes, planes, (1 if li == 0 else 2) if bi == 0 else 1))
                inplanes = planes
            self.add_module(str(4 + li), nn.Sequential(*blocks))
        self.n_layers = len(depths)
        self.out_channels = inplanes
        self.avgpool = avgpool
        self.compute_dtype = _DTYPES[compute_dtype]
        self._folded = None
        self.register_load_state_dict_post_hook(lambda m, _k: setattr(m, '_folded', None))
        for p in self.parameters():
            p.requires_grad_(False)

    def _apply(self, fn, *a, **kw):  # .to() / .cuda() move the parameters: fold again on the new device
        self._folded = None
        return super()._apply(fn, *a, **kw)

    def refold(self):
        """(re)build the folded compute-dtype weights; call after editing parameters in place."""
        dt = self.compute_dtype
        f = {'stem': _fold(getattr(self, '0'), getattr(self, '1'), dt), 'blocks': []}
        for li in range(self.n_layers):
            for blk in getattr(self, str(4 + li)):
                f['blocks'].append((blk.stride, blk.conv1.out_channels, _fold(blk.conv1, blk.bn1, dt), _fold(blk.conv2, blk.bn2, dt),
                                    _fold(blk.downsample[0], blk.downsample[1], dt) if blk.downsample is not None else None))
        self._folded = f
        return f

    @torch.no_grad()
    def forward(self, pixel_values):
        """pixel_values [n,3,H,W] fp32 (normalised as for torchvision's ImageNet weights) -> tokens [n, h*w, C] in the compute
        dtype (row-major over (h, w): the order backbone.py:85-87 flattens to), or [n, C] fp32 with ``avgpool``."""
        if not pixel_values.is_cuda:
            raise RuntimeError('svol_amd ResNetExtractor runs on the MI355X HIP kernels only (no CPU path)')
        if self.training and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('the ResNet extractor is inference / frozen only (no BatchNorm batch statistics, no backward)')
        dt = self.compute_dtype
        f = self._folded or self.refold()
        x = pixel_values.float().contiguous()
        n, c, H, W = x.shape
        w0, b0 = f['stem']
        cols, H, W = ops.im2col(x, n, H, W, c, 7, 7, 2, 3, dt, strides=(c * H * W, W, 1, H * W), ldcols=w0.shape[1])
        y = ops.gemm_nt(cols, w0, b0, ops.ACT_RELU)
        del cols
        C = w0.shape[0]
        y, H, W = ops.maxpool_nhwc(y, n, H, W, C, 3, 2, 1)
        for stride, planes, (w1, b1), (w2, b2), down in f['blocks']:
            t, Ho, Wo = ops.conv_nhwc(y, w1, b1, ops.ACT_RELU, n, H, W, C, 3, 3, stride, 1)
            idt = ops.conv_nhwc(y, down[0], down[1], ops.ACT_NONE, n, H, W, C, 1, 1, stride, 0)[0] if down is not None else y
            y, _, _ = ops.conv_nhwc(t, w2, b2, ops.ACT_RELU_RES, n, Ho, Wo, planes, 3, 3, 1, 1, residual=idt)
            H, W, C = Ho, Wo, planes
        if self.avgpool:
            return ops.avgpool_nhwc(y, n, H * W, C)
        return y.view(n, H * W, C)


def resnet18(avgpool=False, compute_dtype='bf16'):
    return ResNetExtractor((2, 2, 2, 2), avgpool=avgpool, compute_dtype=compute_dtype)


def resnet34(avgpool=False, compute_dtype='bf16'):
    return ResNetExtractor((3, 4, 6, 3), avgpool=avgpool, compute_dtype=compute_dtype)


class ResNetBackbone(nn.Module):
    """backbone.py:65-89: sketch [N,1,3,H,W] -> [N,1,C] (global average pooled), frames [N,T,3,H,W] -> [N, T*h*w, C]."""

    def __init__(self, video_backbone, sketch_backbone):
        super().__init__()
        self.video_backbone = video_backbone
        self.sketch_backbone = sketch_backbone

    def forward(self, sketch_batch, video_batch):
        N, T = video_batch.shape[:2]
        src_sketch = self.sketch_backbone(sketch_batch.flatten(0, 1)).view(N, -1, self.sketch_backbone.out_channels)
        tok = self.video_backbone(video_batch.flatten(0, 1))                  # [N*T, h*w, C]
        return src_sketch, tok.reshape(N, T * tok.shape[1], tok.shape[2])
