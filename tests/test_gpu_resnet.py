"""GPU parity of the ResNet extractor: im2col / pooling kernels against torch's unfold / pooling, the fused
convolution epilogues, and ResNet-18 / ResNet-34 / toy networks through the product module against transformers.ResNetModel
goldens and the CPU oracle (fp32 1e-3, bf16 relative to the feature scale)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from svol_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_im2col_and_pooling(dtype):
    from svol_amd import ops
    g = torch.Generator().manual_seed(3)
    for (N, H, W, C, k, s, p) in [(2, 9, 11, 16, 3, 1, 1), (3, 12, 12, 8, 3, 2, 1), (2, 8, 8, 32, 1, 2, 0), (1, 14, 10, 24, 7, 2, 3)]:
        x = torch.randn(N, C, H, W, generator=g).to(dtype)
        nhwc = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        K = k * k * C
        ld = (K + 31) // 32 * 32
        cols, Ho, Wo = ops.im2col(nhwc, N, H, W, C, k, k, s, p, dtype, ldcols=ld)
        ref = F.unfold(x.float(), k, padding=p, stride=s)                       # [N, C*k*k, Ho*Wo], (c, ky, kx) order
        ref = ref.view(N, C, k, k, Ho * Wo).permute(0, 4, 2, 3, 1).reshape(N * Ho * Wo, K)
        assert torch.equal(cols[:, :K].float().cpu(), ref) and float(cols[:, K:].abs().max() if ld > K else 0) == 0.0
        y, Hp, Wp = ops.maxpool_nhwc(nhwc, N, H, W, C, 3, 2, 1)
        refp = F.max_pool2d(x.float(), 3, 2, 1).permute(0, 2, 3, 1).reshape(-1, C)
        assert torch.equal(y.float().cpu(), refp)
        a = ops.avgpool_nhwc(nhwc, N, H * W, C)
        assert float((a.cpu() - x.float().mean((2, 3))).abs().max()) < 1e-5
    # the stem: NCHW fp32 pixels -> (ky, kx, c) columns in the compute dtype, K = 147 padded to 160
    x = torch.randn(2, 3, 20, 18, generator=g)
    cols, Ho, Wo = ops.im2col(x.to(DEV), 2, 20, 18, 3, 7, 7, 2, 3, dtype, strides=(3 * 20 * 18, 18, 1, 20 * 18), ldcols=160)
    ref = F.unfold(x, 7, padding=3, stride=2).view(2, 3, 7, 7, Ho * Wo).permute(0, 4, 2, 3, 1).reshape(-1, 147).to(dtype)
    assert torch.equal(cols[:, :147].cpu(), ref) and float(cols[:, 147:].abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_gemm_relu_after_residual(dtype):
    from svol_amd import ops
    g = torch.Generator().manual_seed(5)
    for (M, N, K) in [(300, 64, 576), (5000, 256, 1152), (130, 512, 64), (2000, 128, 160)]:
        A = torch.randn(M, K, generator=g).to(dtype)
        Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
        b = torch.randn(N, generator=g)
        R = torch.randn(M, N, generator=g).to(dtype)
        out = ops.gemm_nt(A.to(DEV), Wt.to(DEV), b.to(DEV), ops.ACT_RELU_RES, residual=R.to(DEV))
        ref = torch.relu(A.double() @ Wt.double().t() + b.double() + R.double())
        err = float((out.double().cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err < (2e-5 if dtype == torch.float32 else 1.2e-2), (M, N, K, err)
        out = ops.gemm_nt(A.to(DEV), Wt.to(DEV), b.to(DEV), ops.ACT_RELU)
        ref = torch.relu(A.double() @ Wt.double().t() + b.double())
        err = float((out.double().cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err < (2e-5 if dtype == torch.float32 else 1.2e-2), (M, N, K, err)


@pytest.mark.parametrize('name', ['resnet_tiny', 'resnet_tiny3', 'resnet18_1img', 'resnet34_1img'])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_resnet_extractor_golden(name, dtype):
    from svol_amd.modeling.resnet import ResNetExtractor
    from tests.test_oracle_resnet import resnet_case
    z, meta, sd, x = resnet_case(name)
    scale = float(np.abs(z['tokens']).max())
    # bf16: ~17 (ResNet-18) / 33 (ResNet-34) chained bf16 GEMMs with bf16 activations in between
    tol = 1e-3 if dtype == 'fp32' else 3e-2
    for avg in (False, True):
        m = ResNetExtractor(tuple(meta['depths']), tuple(meta['widths']), meta['stem'], avgpool=avg, compute_dtype=dtype)
        m.load_state_dict(sd, strict=True)
        m.to(DEV).eval()
        out = m(x.to(DEV)).float().cpu()
        ref = torch.from_numpy(z['pooled'] if avg else z['tokens'])
        assert out.shape == ref.shape
        err = float((out - ref).abs().max()) / scale
        assert err <= tol, f'{name} {dtype} avgpool={avg}: {err:.3e} > {tol:.1e}'
        l2 = float((out - ref).norm() / ref.norm())
        assert l2 <= (1e-4 if dtype == 'fp32' else 1e-2), f'{name} {dtype} avgpool={avg}: L2 {l2:.3e}'


def test_resnet_backbone_through_build_model():
    """--backbone resnet: frames -> 49 tokens each, sketch -> one pooled token, into the SVANet head (bf16)."""
    from oracle import resnet_oracle as R
    from svol_amd.modeling.model import build_model
    args = syn.head_args(hidden_dim=64, nheads=8, num_layers=1, num_queries=10, num_frames=2, backbone='resnet')
    torch.manual_seed(1)
    model = build_model(args)
    assert args.input_vid_dim == 512 and args.input_skch_dim == 512
    sdv = syn.synth_resnet_state_dict(syn.resnet_param_shapes((3, 4, 6, 3)), seed=1)
    sds = syn.synth_resnet_state_dict(syn.resnet_param_shapes((2, 2, 2, 2)), seed=2)
    model.backbone.video_backbone.load_state_dict(sdv)
    model.backbone.sketch_backbone.load_state_dict(sds)
    model.to(DEV).eval()
    vid = syn.synth_images(2, syn.vit_config(image_size=224), seed=4).view(1, 2, 3, 224, 224)
    sk = syn.synth_images(1, syn.vit_config(image_size=224), seed=5).view(1, 1, 3, 224, 224)
    s, v = model.backbone(sk.to(DEV), vid.to(DEV))
    rs, rv = R.resnet_backbone_forward(sdv, (3, 4, 6, 3), sds, (2, 2, 2, 2), sk, vid)
    assert s.shape == (1, 1, 512) and v.shape == (1, 98, 512)
    assert float((v.float().cpu() - rv).abs().max()) / float(rv.abs().max()) < 3e-2
    assert float((s.float().cpu() - rs).abs().max()) / float(rs.abs().max()) < 3e-2
    out = model(sk.to(DEV), vid.to(DEV), torch.ones(1, 1, device=DEV), torch.ones(1, 2, device=DEV))
    assert out['pred_boxes'].shape == (1, 10, 4) and bool(torch.isfinite(out['pred_logits']).all())


def test_implicit_conv_matches_im2col_path():
    """svol_conv_nhwc (A operand gathered inside the GEMM's LDS-DMA loads) against torch's conv2d: 3x3 s1 / s2 with padding,
    1x1 s2, ragged M (rows past the last tile), residual + ReLU-after-residual, K step 32 and 64."""
    from svol_amd import ops
    g = torch.Generator().manual_seed(11)
    for (N, H, W, C, Cout, k, s, p, act, res) in [(2, 14, 14, 64, 64, 3, 1, 1, 'relu', False), (3, 13, 9, 64, 128, 3, 2, 1, 'relu', False),
                                                  (2, 12, 12, 128, 256, 1, 2, 0, 'none', False), (2, 10, 10, 128, 128, 3, 1, 1, 'relu_res', True),
                                                  (1, 7, 7, 256, 512, 3, 1, 1, 'relu_res', True), (5, 9, 11, 32, 36, 3, 1, 1, 'none', False)]:
        x = torch.randn(N, C, H, W, generator=g)
        w = torch.randn(Cout, C, k, k, generator=g) / (C * k * k) ** 0.5
        b = torch.randn(Cout, generator=g)
        xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
        ref = F.conv2d(xb.double(), wb.double(), b.double(), s, p)
        Ho, Wo = ref.shape[2], ref.shape[3]
        r = torch.randn(N, Cout, Ho, Wo, generator=g).to(torch.bfloat16) if res else None
        if res:
            ref = ref + r.double()
        if act != 'none':
            ref = torch.relu(ref)
        nhwc = xb.permute(0, 2, 3, 1).contiguous().view(-1, C).to(DEV)
        wf = wb.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
        rr = r.permute(0, 2, 3, 1).contiguous().view(-1, Cout).to(DEV) if res else None
        code = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'relu_res': ops.ACT_RELU_RES}[act]
        y, ho, wo = ops.conv_nhwc(nhwc, wf, b.to(DEV), code, N, H, W, C, k, k, s, p, residual=rr)
        assert (ho, wo) == (Ho, Wo)
        got = y.float().cpu().view(N, Ho, Wo, Cout).permute(0, 3, 1, 2).double()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        assert err < 1.2e-2, (N, H, W, C, Cout, k, s, p, act, err)
