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

Two modes.  Frozen / eval (the default; BatchNorm uses its running statistics, folded into the weights as above, no backward).
``trainable=True`` and ``.train()``: what the reference's training step runs (train.py:72 puts EVERY parameter of build_model(args)
into the optimiser, and ``model.train()`` puts torchvision's BatchNorm into batch-statistics mode): every conv -> BatchNorm (-> +identity)
(-> ReLU) unit is one autograd Function (``_ConvBnFn``) over csrc/resnet_train.hip —

  forward   z = conv(x) (svol_conv_nhwc / im2col + GEMM, no bias) ; batch mean / biased variance per channel (two column passes) ;
            y = relu?(z * gamma * rstd + (beta - mean * gamma * rstd) + identity) ; running statistics updated with the UNBIASED variance
  backward  g = dy * [y > 0] ; dbeta = sum g, dgamma = sum g * xhat ; dz = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) ; d identity = g ;
            dW = dz^T im2col(x) (svol_gemm_tn) ; dx = col2im(dz W) (svol_gemm_nt + svol_col2im_nhwc)

— plus max pooling with its argmax and the sketch branch's average pooling as Functions.  Activations bf16 / fp16 NHWC, statistics,
parameters and parameter gradients fp32.
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


def _w16(weight, dtype):
    """[Cout, Cin, kh, kw] fp32 -> [Cout, Kp] compute dtype, (ky, kx, c) order, K padded to a multiple of 32 (the GEMMs' layout)."""
    return ops.conv_weight_pack(weight, dtype)


def _bn_forward(z, bn, gamma, beta, residual, relu, sync=False):
    """train-mode BatchNorm of the conv output z [M, C] (+ identity) (+ ReLU); updates bn's running statistics as nn.BatchNorm2d does."""
    if bn.momentum is None:
        raise NotImplementedError('BatchNorm2d(momentum=None) (cumulative average) is not on the reference path')
    track = bn.track_running_stats and bn.running_mean is not None
    mean, rstd, scale, shift = ops.bn_train_stats(z, gamma.detach(), beta.detach(), bn.running_mean if track else None,
                                                  bn.running_var if track else None, bn.momentum, bn.eps, sync)
    if track:
        with torch.no_grad():
            bn.num_batches_tracked += 1
    return ops.bn_apply(z, scale, shift, residual, relu), mean, rstd


def _weight_grad(dz, make_dwp, weight, sink, keep):
    """dW of a convolution in the parameter's own layout: make_dwp() -> dz^T im2col(x) as [Cout, Kp] fp32 (svol_conv_wgrad_nhwc: the im2col
    view gathered inside the weight-gradient GEMM; the stem: svol_gemm_tn on its kept im2col matrix), folded back by one kernel.
    With a gradient sink (the parameter's .grad is a view of a BucketedGradAllReduce bucket) the three launches go to the
    weight-gradient stream (ops.py "weight gradients off the critical path": nobody needs dW before the optimiser; the dx chain behind
    it is the backward's critical path) and add straight into the bucket: returns None.  `keep`: tensors the side stream reads."""
    if sink is not None and ops.WGRAD_ASYNC and not torch.cuda.is_current_stream_capturing():
        cur = torch.cuda.current_stream(dz.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        ws = ops._wgrad_stream(dz.device)
        ws.wait_event(ev)
        with torch.cuda.stream(ws):
            ops.conv_weight_unpack_add(make_dwp(), sink.view.view(weight.shape))
        for t in keep:
            t.record_stream(ws)
        return None
    dW = torch.zeros_like(weight, dtype=torch.float32, memory_format=torch.contiguous_format)
    ops.conv_weight_unpack_add(make_dwp(), dW)
    if sink is not None:
        sink.view.view(weight.shape).add_(dW)
        return None
    return dW


class _ConvBnFn(torch.autograd.Function):
    """y [n*ho*wo, Cout] = relu?(BatchNorm_train(conv(x)) + identity) on NHWC activations; see the module docstring."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, identity, geom, relu, bn, dt, sync=False):
        n, H, W, C, kh, kw, stride, pad = geom
        w16 = _w16(weight, dt)
        z, Ho, Wo = ops.conv_nhwc(x, w16, None, ops.ACT_NONE, n, H, W, C, kh, kw, stride, pad)
        y, mean, rstd = _bn_forward(z, bn, gamma, beta, identity, relu, sync)
        ctx.geom, ctx.relu, ctx.has_id, ctx.out_hw, ctx.sync = geom, relu, identity is not None, (Ho, Wo), sync
        ctx.sink = ops._claim(weight, ctx.needs_input_grad[1])
        ctx.save_for_backward(x, weight, w16, z, y if relu else None, mean, rstd, gamma.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, w16, z, y, mean, rstd, gamma = ctx.saved_tensors
        n, H, W, C, kh, kw, stride, pad = ctx.geom
        Ho, Wo = ctx.out_hw
        dz, did, dgamma, dbeta = ops.bn_bwd(dy.contiguous(), y, z, mean, rstd, gamma, ctx.has_id and ctx.needs_input_grad[4], ctx.sync)
        Cout = weight.shape[0]
        dW = None
        if ctx.needs_input_grad[1]:
            def dwp():
                d = ops.conv_wgrad_nhwc(dz, x, n, H, W, C, kh, kw, stride, pad, w16.shape[1])
                if d is None:   # a shape the gathering kernel does not take: the explicit matrix
                    d = ops.gemm_tn(dz, ops.im2col(x, n, H, W, C, kh, kw, stride, pad, x.dtype, ldcols=w16.shape[1])[0])
                return d
            dW = _weight_grad(dz, dwp, weight, ctx.sink, (x, dz))
        dx = None
        if ctx.needs_input_grad[0]:
            if stride == 1 and 2 * pad == kh - 1 and kh == kw:
                # the data gradient of a "same" stride-1 convolution is the convolution of dz with the reversed taps and swapped channels:
                # the implicit-GEMM kernel again, no [M, 9 C] matrix
                dx = ops.conv_nhwc(dz, ops.conv_weight_pack(weight, x.dtype, flip=True), None, ops.ACT_NONE, n, Ho, Wo, Cout, kh, kw, 1, pad)[0]
            else:
                dx = ops.col2im_nhwc(ops.gemm_nt(dz, w16.t().contiguous()), n, H, W, C, kh, kw, stride, pad)
        return dx, dW, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None, did, None, None, None, None, None


class _StemFn(torch.autograd.Function):
    """pixels [n,3,H,W] fp32 -> relu(BatchNorm_train(conv7x7 s2 p3)) as NHWC [n*ho*wo, C]; no gradient for the pixels."""

    @staticmethod
    def forward(ctx, pix, weight, gamma, beta, bn, dt, sync=False):
        n, c, H, W = pix.shape
        w16 = _w16(weight, dt)
        kh, kw = weight.shape[2], weight.shape[3]
        cols, Ho, Wo = ops.im2col(pix, n, H, W, c, kh, kw, 2, 3, dt, strides=(c * H * W, W, 1, H * W), ldcols=w16.shape[1])
        z = ops.gemm_nt(cols, w16)
        y, mean, rstd = _bn_forward(z, bn, gamma, beta, None, True, sync)
        ctx.sync = sync
        ctx.sink = ops._claim(weight, ctx.needs_input_grad[1])
        # the stem's im2col matrix is KEPT for the weight gradient (1 GB at 256 frames of 224 x 224 — 0.4 % of this part's HBM; building
        # it again from the NCHW fp32 pixels costs 0.6 ms) — unless the stem's weight takes no gradient
        ctx.save_for_backward(cols if ctx.needs_input_grad[1] else None, weight, z, y, mean, rstd, gamma.detach())
        return y

    @staticmethod
    def backward(ctx, dy):
        cols, weight, z, y, mean, rstd, gamma = ctx.saved_tensors
        if not any(ctx.needs_input_grad[1:4]):   # a frozen stem: the pixels take no gradient either, nothing to compute
            return None, None, None, None, None, None, None
        dz, _, dgamma, dbeta = ops.bn_bwd(dy.contiguous(), y, z, mean, rstd, gamma, False, ctx.sync)
        dW = _weight_grad(dz, lambda: ops.gemm_tn(dz, cols), weight, ctx.sink, (cols, dz)) if ctx.needs_input_grad[1] else None
        return None, dW, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None, None, None, None


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geom):
        n, H, W, C, k, stride, pad = geom
        y, idx, Ho, Wo = ops.maxpool_idx_nhwc(x, n, H, W, C, k, stride, pad)
        ctx.geom = geom
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, H, W, C, k, stride, pad = ctx.geom
        return ops.maxpool_bwd_nhwc(dy.contiguous(), ctx.saved_tensors[0], n, H, W, C, k, stride, pad), None


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n, HW, C):
        ctx.shape, ctx.dt = (n, HW, C), x.dtype
        return ops.avgpool_nhwc(x, n, HW, C)

    @staticmethod
    def backward(ctx, dy):
        n, HW, C = ctx.shape
        return (dy.float() / HW).to(ctx.dt).view(n, 1, C).expand(n, HW, C).reshape(n * HW, C).contiguous(), None, None, None


class ResNetExtractor(nn.Module):
    def __init__(self, depths=(2, 2, 2, 2), widths=(64, 128, 256, 512), stem=64, avgpool=False, compute_dtype='bf16', trainable=False,
                 sync_bn=False):
        super().__init__()
        # --sync_bn (apex.parallel.convert_syncbn_model, train.py:65-68 / test.py:61): train-mode BatchNorm statistics over the
        # GLOBAL batch of all ranks of the default process group; no effect at world size 1 or on the frozen extractor
        self.sync_bn = bool(sync_bn)
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
        self.trainable = bool(trainable)
        for p in self.parameters():
            p.requires_grad_(self.trainable)

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

    def forward(self, pixel_values):
        """pixel_values [n,3,H,W] fp32 (normalised as for torchvision's ImageNet weights) -> tokens [n, h*w, C] in the compute
        dtype (row-major over (h, w): the order backbone.py:85-87 flattens to), or [n, C] fp32 with ``avgpool``."""
        if not pixel_values.is_cuda:
            raise RuntimeError('svol_amd ResNetExtractor runs on the MI355X HIP kernels only (no CPU path)')
        if self.training and self.trainable:
            return self._forward_train(pixel_values)
        if self.training and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('parameters that require gradients need ResNetExtractor(trainable=True): the frozen path folds '
                                      'the running statistics into the weights and has no backward')
        with torch.no_grad():
            return self._forward_frozen(pixel_values)

    def _forward_train(self, pixel_values):
        """batch-statistics BatchNorm, autograd through every unit (module docstring); folded weights are invalidated."""
        dt = self.compute_dtype
        if dt != torch.bfloat16:
            raise NotImplementedError('the training path of the ResNet extractor keeps its activations in bf16 (compute_dtype="bf16")')
        self._folded = None
        x = pixel_values.float().contiguous()
        n, c, H, W = x.shape
        conv0, bn0 = getattr(self, '0'), getattr(self, '1')
        sync = self.sync_bn
        y = _StemFn.apply(x, conv0.weight, bn0.weight, bn0.bias, bn0, dt, sync)
        H, W, C = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1, conv0.out_channels
        y = _MaxPoolFn.apply(y, (n, H, W, C, 3, 2, 1))
        H, W = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        for li in range(self.n_layers):
            for blk in getattr(self, str(4 + li)):
                s_, planes = blk.stride, blk.conv1.out_channels
                Ho, Wo = (H + 2 - 3) // s_ + 1, (W + 2 - 3) // s_ + 1
                t = _ConvBnFn.apply(y, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, None, (n, H, W, C, 3, 3, s_, 1), True, blk.bn1, dt, sync)
                idt = y
                if blk.downsample is not None:
                    idt = _ConvBnFn.apply(y, blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias, None,
                                          (n, H, W, C, 1, 1, s_, 0), False, blk.downsample[1], dt, sync)
                y = _ConvBnFn.apply(t, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias, idt, (n, Ho, Wo, planes, 3, 3, 1, 1), True, blk.bn2, dt, sync)
                H, W, C = Ho, Wo, planes
        if self.avgpool:
            return _AvgPoolFn.apply(y, n, H * W, C)
        return y.view(n, H * W, C)

    def _forward_frozen(self, pixel_values):
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


def resnet18(avgpool=False, compute_dtype='bf16', trainable=False, sync_bn=False):
    return ResNetExtractor((2, 2, 2, 2), avgpool=avgpool, compute_dtype=compute_dtype, trainable=trainable, sync_bn=sync_bn)


def resnet34(avgpool=False, compute_dtype='bf16', trainable=False, sync_bn=False):
    return ResNetExtractor((3, 4, 6, 3), avgpool=avgpool, compute_dtype=compute_dtype, trainable=trainable, sync_bn=sync_bn)


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
