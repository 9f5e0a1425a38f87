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


class _RoundBf16(torch.autograd.Function):
    """x -> bf16(x) in the forward AND the backward: marks the places where the HIP path keeps an activation / its gradient in bf16."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _emulated_bf16_reference(sd, depths, x, probe_fmap, avg):
    """the same network in torch fp32 on the GPU with the HIP path's roundings (conv output, unit output, their gradients; bf16 conv
    weights): a reference whose ReLU masks and BatchNorm cancellations see bf16 activations like the product does."""
    import torch.nn.functional as F
    r = _RoundBf16.apply
    P = {k: (v.detach().to(DEV).float().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    w = lambda k: r(P[k])

    def bn(z, p):
        return F.batch_norm(z, None, None, P[p + '.weight'], P[p + '.bias'], True, 0.1, 1e-5)
    h = r(F.relu(bn(r(F.conv2d(r(x.to(DEV).float()), w('0.weight'), None, 2, 3)), '1')))
    h = F.max_pool2d(h, 3, 2, 1)
    for li, nb in enumerate(depths):
        for bi in range(nb):
            p, stride = f'{4 + li}.{bi}', (1 if li == 0 else 2) if bi == 0 else 1
            t = r(F.relu(bn(r(F.conv2d(h, w(p + '.conv1.weight'), None, stride, 1)), p + '.bn1')))
            idt = h
            if (p + '.downsample.0.weight') in P:
                idt = r(bn(r(F.conv2d(h, w(p + '.downsample.0.weight'), None, stride, 0)), p + '.downsample.1'))
            h = r(F.relu(bn(r(F.conv2d(t, w(p + '.conv2.weight'), None, 1, 1)), p + '.bn2') + idt))
    f = F.adaptive_avg_pool2d(h, 1) if avg else h
    (f * probe_fmap.to(DEV).float()).sum().backward()
    return f.detach(), {k: v.grad for k, v in P.items() if torch.is_tensor(v) and v.requires_grad}


@pytest.mark.parametrize('name', ['resnet_tiny_train', 'resnet_tiny3_train'])
@pytest.mark.parametrize('avg', [False, True], ids=['tokens', 'pooled'])
@pytest.mark.parametrize('two_pass', [True, False], ids=['two-pass-stats', 'one-pass-stats'])
def test_resnet_training_mode_gradients(name, avg, two_pass, monkeypatch):
    """ResNetExtractor(trainable=True).train(): batch-statistics BatchNorm forward, every parameter gradient and the running statistics
    — VERDICT r4 "Next round 8" (the reference optimises the backbone too: train.py:72, backbone.py:133-152).  Two references on the
    same weights, pixels and probe: (a) the fp64 oracle, itself pinned to transformers.ResNetModel.train() by the *_train fixtures:
    features and running statistics tightly, gradients loosely — these nets normalise over 32 ... 768 samples, where ONE ReLU mask
    that flips under bf16 rounding moves a bias gradient by several per cent; (b) the same network in torch fp32 with the product's
    bf16 roundings (activations, their gradients, conv weights): gradients per tensor to a few per cent.  The kernels themselves are
    checked one by one against fp64 in test_resnet_training_kernels."""
    from oracle import resnet_oracle as R
    from svol_amd import ops
    from svol_amd.modeling.resnet import ResNetExtractor
    from tests.test_oracle_resnet import resnet_case
    # two_pass: the statistics in torch's own order (mean, then the moment about it), equal to the references' to the last bits — the
    # tight bounds; the product's default (one pass about a pivot, 1e-6 from them) flips a few bf16 roundings and ReLU masks of these
    # 32 ... 2048-sample normalisations: the loose bounds
    monkeypatch.setattr(ops, 'BN_STATS_TWO_PASS', two_pass)
    z, meta, sd, x = resnet_case(name)
    depths = tuple(meta['depths'])
    g = torch.Generator().manual_seed(11)
    m = ResNetExtractor(depths, tuple(meta['widths']), meta['stem'], avgpool=avg, compute_dtype='bf16', trainable=True)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).train()
    out = m(x.to(DEV))
    probe = torch.randn(out.shape, generator=g)
    (out.float() * probe.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    side = int(round(out.shape[1] ** 0.5))
    # the oracle's features are [n, C, h, w]; tokens are (h, w)-major rows of C
    probe_fmap = probe.view(out.shape[0], -1, 1, 1) if avg else probe.transpose(1, 2).reshape(out.shape[0], -1, side, side)
    f, grads, stats = R.resnet_train_grads(sd, depths, x, probe_fmap, avgpool=avg)
    ref_out = f.flatten(1) if avg else f.flatten(2).transpose(1, 2)
    got = out.detach().float().cpu().double()
    l2 = float((got - ref_out).norm() / ref_out.norm())
    assert l2 <= 2.5e-2, f'features vs fp64: L2 {l2:.3e}'
    for k, v in stats.items():
        have = dict(m.named_buffers())[k].double().cpu()
        assert float((have - v.double()).abs().max()) <= 1e-2 * max(1.0, float(v.abs().max())), k
    assert int(dict(m.named_buffers())['1.num_batches_tracked']) == 1
    fe, ge = _emulated_bf16_reference(sd, depths, x, probe_fmap, avg)
    e64, eem = {}, {}
    for k, p in m.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
        e64[k] = float((p.grad.double().cpu() - grads[k]).norm() / max(float(grads[k].norm()), 1e-12))
        eem[k] = float((p.grad.double().cpu() - ge[k].double().cpu()).norm() / max(float(ge[k].norm()), 1e-12))
    print(f'{name} avg={avg}: feature L2 {l2:.2e}; gradient L2 vs fp64 max {max(e64.values()):.2e} median {sorted(e64.values())[len(e64) // 2]:.2e}; '
          f'vs the bf16-rounding reference max {max(eem.values()):.2e} median {sorted(eem.values())[len(eem) // 2]:.2e}')
    # measured (round 5): vs fp64 max 0.17 - 0.33, median 0.06 - 0.24; vs the bf16-rounding reference tiny (48 samples in the last
    # stage) max 0.06 - 0.12, median 0.04 - 0.08, tiny3 (64 x 64 pixels) max 0.016, median 0.007
    assert max(e64.values()) <= 0.6, {k: round(v, 3) for k, v in e64.items() if v > 0.6}
    med = sorted(eem.values())[len(eem) // 2]
    bad = {k: round(v, 4) for k, v in eem.items() if not v <= 0.2}
    assert not bad and med <= (0.03 if (name == 'resnet_tiny3_train' and two_pass) else 0.12), (bad, med)


def test_resnet_training_kernels():
    """csrc/resnet_train.hip one kernel at a time against torch fp64 on the same bf16 values: conv forward, batch statistics, the BatchNorm
    apply, its backward (with the kernel's own ReLU mask), the weight gradient (svol_gemm_tn of im2col), the data gradient (svol_gemm_nt +
    col2im), max pooling with ties (post-ReLU zeros) and its backward — 3x3 s1 / s2, 1x1 s2, C = 16 ... 64 (both convolution paths)."""
    import torch.nn.functional as F
    from svol_amd import ops
    from svol_amd.modeling.resnet import _w16

    def rel(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).norm() / max(float(b.norm()), 1e-30))
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    res = {}
    for (n, H, W, C, Cout, k, s, p) in [(3, 8, 8, 16, 32, 3, 2, 1), (2, 16, 16, 32, 32, 3, 1, 1), (2, 8, 8, 64, 128, 1, 2, 0), (2, 14, 14, 64, 64, 3, 1, 1),
                                       (5, 28, 28, 128, 256, 3, 2, 1), (3, 7, 7, 512, 512, 3, 1, 1), (9, 56, 56, 64, 64, 3, 1, 1)]:
        tag = f'{n}x{H}x{W}x{C}->{Cout} k{k}s{s}'
        x = torch.randn(n, C, H, W, generator=g)
        w = torch.randn(Cout, C, k, k, generator=g) * 0.1
        x16 = x.permute(0, 2, 3, 1).reshape(n * H * W, C).to(dt).to(DEV).contiguous()
        xr = x16.double().cpu().view(n, H, W, C).permute(0, 3, 1, 2).requires_grad_(True)
        w16 = _w16(w.to(DEV), dt)
        wr = w16[:, :k * k * C].double().cpu().view(Cout, k, k, C).permute(0, 3, 1, 2).requires_grad_(True)
        z, Ho, Wo = ops.conv_nhwc(x16, w16, None, ops.ACT_NONE, n, H, W, C, k, k, s, p)
        zr = F.conv2d(xr, wr, None, s, p)
        res[tag + ' conv'] = (rel(z, zr.detach().permute(0, 2, 3, 1).reshape(-1, Cout)), 4e-3)
        M = z.shape[0]
        gamma = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV)
        beta = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
        rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
        mean, rstd, scale, shift = ops.bn_train_stats(z, gamma, beta, rm, rv, 0.1, 1e-5)
        zd = z.double().cpu()
        var_ref = zd.var(0, unbiased=False)
        res[tag + ' mean'] = (float(((mean.double().cpu() - zd.mean(0)).abs() / zd.std(0)).max()), 1e-6)   # (in standard deviations)
        res[tag + ' rstd'] = (rel(rstd, 1.0 / torch.sqrt(var_ref + 1e-5)), 1e-5)
        res[tag + ' running_mean'] = (float(((rm.double().cpu() - 0.1 * zd.mean(0)).abs() / zd.std(0)).max()), 1e-6)
        res[tag + ' running_var'] = (rel(rv, 0.9 + 0.1 * zd.var(0, unbiased=True)), 1e-5)
        idt = torch.randn(M, Cout, generator=g).to(dt).to(DEV)
        y = ops.bn_apply(z, scale, shift, idt, True)
        zz, gr, br, rr = zd.clone().requires_grad_(True), gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True), idt.double().cpu().requires_grad_(True)
        pre = F.batch_norm(zz, None, None, gr, br, True, 0.1, 1e-5) + rr
        res[tag + ' bn_apply'] = (rel(y, torch.relu(pre).detach()), 4e-3)
        dy = torch.randn(M, Cout, generator=g).to(dt).to(DEV)
        dz, dres, dgm, dbt = ops.bn_bwd(dy, y, z, mean, rstd, gamma, True)
        (pre * (dy.double().cpu() * (y.double().cpu() > 0).double())).sum().backward()
        res[tag + ' dz'] = (rel(dz, zz.grad), 4e-3)
        res[tag + ' didentity'] = (rel(dres, rr.grad), 1e-6)
        res[tag + ' dgamma'] = (rel(dgm, gr.grad), 1e-5)
        res[tag + ' dbeta'] = (rel(dbt, br.grad), 1e-5)
        zr.backward(dz.double().cpu().view(n, Ho, Wo, Cout).permute(0, 3, 1, 2))
        cols, _, _ = ops.im2col(x16, n, H, W, C, k, k, s, p, dt, ldcols=w16.shape[1])
        dW = ops.gemm_tn(dz, cols)[:, :k * k * C].reshape(Cout, k, k, C).permute(0, 3, 1, 2)
        res[tag + ' dW'] = (rel(dW, wr.grad), 1e-5)
        dwg = ops.conv_wgrad_nhwc(dz, x16, n, H, W, C, k, k, s, p, w16.shape[1])
        assert dwg is not None
        res[tag + ' dW (gathered in the GEMM)'] = (rel(dwg[:, :k * k * C].reshape(Cout, k, k, C).permute(0, 3, 1, 2), wr.grad), 1e-5)
        assert float(dwg[:, k * k * C:].abs().max()) == 0.0 if w16.shape[1] > k * k * C else True
        w32 = wr.detach().float().to(DEV).contiguous()
        res[tag + ' weight pack'] = (rel(ops.conv_weight_pack(w32, dt), w16), 0.0)
        acc = torch.ones_like(w32)
        ops.conv_weight_unpack_add(ops.gemm_tn(dz, cols), acc)
        res[tag + ' weight unpack-add'] = (rel(acc - 1.0, wr.grad), 1e-5)
        dx = ops.col2im_nhwc(ops.gemm_nt(dz, w16.t().contiguous()), n, H, W, C, k, k, s, p)
        res[tag + ' dx'] = (rel(dx.view(n, H, W, C).permute(0, 3, 1, 2), xr.grad), 5e-3)
        if s == 1:
            dx2 = ops.conv_nhwc(dz, ops.conv_weight_pack(w32, dt, flip=True), None, ops.ACT_NONE, n, Ho, Wo, Cout, k, k, 1, p)[0]
            res[tag + ' dx (as a convolution of dz)'] = (rel(dx2.view(n, H, W, C).permute(0, 3, 1, 2), xr.grad), 5e-3)
    # one-pass statistics about a pivot: channels whose mean is far from zero (60 standard deviations) must not lose their variance
    zo = (torch.randn(70001, 64, generator=g) * 0.5 + torch.linspace(-30, 30, 64)).to(dt).to(DEV)
    mean, rstd, _, _ = ops.bn_train_stats(zo, torch.ones(64, device=DEV), torch.zeros(64, device=DEV), None, None, 0.1, 1e-5)
    res['offset channels mean'] = (rel(mean, zo.double().cpu().mean(0)), 1e-5)
    res['offset channels rstd'] = (rel(rstd, 1.0 / torch.sqrt(zo.double().cpu().var(0, unbiased=False) + 1e-5)), 1e-4)
    n, H, W, C = 2, 9, 9, 16
    x16 = torch.relu(torch.randn(n, C, H, W, generator=g)).permute(0, 2, 3, 1).reshape(-1, C).to(dt).to(DEV).contiguous()
    xr = x16.double().cpu().view(n, H, W, C).permute(0, 3, 1, 2).requires_grad_(True)
    y, idx, Ho, Wo = ops.maxpool_idx_nhwc(x16, n, H, W, C, 3, 2, 1)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(n * Ho * Wo, C, generator=g).to(dt).to(DEV)
    yr.backward(dy.double().cpu().view(n, Ho, Wo, C).permute(0, 3, 1, 2))
    res['maxpool fwd'] = (rel(y, yr.detach().permute(0, 2, 3, 1).reshape(-1, C)), 0.0)
    res['maxpool bwd (ties)'] = (rel(ops.maxpool_bwd_nhwc(dy, idx, n, H, W, C, 3, 2, 1).view(n, H, W, C).permute(0, 3, 1, 2), xr.grad), 3e-3)
    bad = {k_: v for k_, v in res.items() if not v[0] <= v[1]}
    assert not bad, bad


def test_resnet_trainable_backbone_in_the_training_step():
    """--backbone resnet with the extractors in the optimiser: one step of the whole model (frames -> ResNet-34 -> SVANet head ->
    criterion -> backward -> AdamW) moves backbone weights, and eval() afterwards runs the frozen path on the UPDATED running statistics."""
    from svol_amd import parallel
    from svol_amd import synthetic as syn
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.resnet import ResNetBackbone, ResNetExtractor
    from svol_amd.modeling.svanet import build_svanet
    args = syn.head_args(hidden_dim=64, nheads=8, num_layers=1, num_queries=10, num_frames=2, input_vid_dim=32, input_skch_dim=32,
                         input_dropout=0.0)
    args.compute_dtype = 'bf16'
    torch.manual_seed(0)
    vb = ResNetExtractor((1, 1), (16, 32), 16, compute_dtype='bf16', trainable=True)
    sb = ResNetExtractor((1, 1), (16, 32), 16, avgpool=True, compute_dtype='bf16', trainable=True)
    vb.load_state_dict(syn.synth_resnet_state_dict(syn.resnet_param_shapes((1, 1), (16, 32), 16), seed=1))
    sb.load_state_dict(syn.synth_resnet_state_dict(syn.resnet_param_shapes((1, 1), (16, 32), 16), seed=2))
    backbone = ResNetBackbone(vb, sb).to(DEV).train()
    head = build_svanet(args).to(DEV).train()
    crit = build_loss(args).to(DEV).train()
    B, T = 2, 2
    vid = syn.synth_images(B * T, syn.vit_config(image_size=64), seed=3).view(B, T, 3, 64, 64).to(DEV)
    sk = syn.synth_images(B, syn.vit_config(image_size=64), seed=4).view(B, 1, 3, 64, 64).to(DEV)
    params = [p for p in list(backbone.parameters()) + list(head.parameters()) if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3)
    before = {k: v.detach().clone() for k, v in backbone.named_parameters()}
    s, v = backbone(sk, vid)
    P = v.shape[1] // T
    out = head(s, torch.ones(B, 1, device=DEV), v, torch.ones(B, T * P, device=DEV))
    crit(out, syn.synth_targets(B, T, seed=1))
    loss = crit.weighted_total()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    moved = [k for k, p in backbone.named_parameters() if not torch.equal(p.detach(), before[k])]
    assert len(moved) == len(before), f'{len(before) - len(moved)} backbone parameters did not move'
    backbone.eval()
    with torch.no_grad():
        s2, v2 = backbone(sk, vid)
    assert bool(torch.isfinite(v2.float()).all()) and bool(torch.isfinite(s2.float()).all())


def test_resnet_weight_gradients_into_gradient_sinks():
    """With BucketedGradAllReduce owning .grad, the convolutions' weight gradients are issued on the weight-gradient stream and added
    straight into the bucket views (resnet.py::_weight_grad): the same numbers as the plain autograd path, after finish() joined the
    side stream."""
    from svol_amd import parallel
    from svol_amd import synthetic as syn
    from svol_amd.modeling.resnet import ResNetExtractor
    sd = syn.synth_resnet_state_dict(syn.resnet_param_shapes((1, 2), (32, 64), 32), seed=1)
    x = syn.synth_images(4, syn.vit_config(image_size=64), seed=5).to(DEV)
    g = torch.Generator().manual_seed(3)

    def run(sinks):
        m = ResNetExtractor((1, 2), (32, 64), 32, compute_dtype='bf16', trainable=True)
        m.load_state_dict(sd)
        m.to(DEV).train()
        red = None
        if sinks:
            red = parallel.BucketedGradAllReduce(list(reversed([p for p in m.parameters()])), ordered=True)
            red.zero_grad()
        out = m(x)
        probe = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
        (out.float() * probe).sum().backward()
        if red is not None:
            red.finish()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    a, b = run(False), run(True)
    for k in a:
        err = float((a[k] - b[k]).norm() / max(float(a[k].norm()), 1e-12))
        assert err <= 2e-3, (k, err)


def test_build_model_with_a_trained_backbone_and_checkpoint_round_trip(tmp_path):
    """args.train_backbone = True through build_model (model.py:31-44 + the reference's train.py:72): the full ResNet-34 / ResNet-18
    extractors in the parameter list, one training step with BucketedGradAllReduce + FlatAdamW (the bench's pair), every backbone
    parameter moves; the state dict (BatchNorm running statistics and counters included) survives a save / load round trip bit for bit."""
    from svol_amd import parallel
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.model import build_model
    args = syn.head_args(hidden_dim=64, nheads=8, num_layers=1, num_queries=10, num_frames=2, backbone='resnet', input_dropout=0.0)
    args.train_backbone = True
    torch.manual_seed(1)
    model = build_model(args)
    model.backbone.video_backbone.load_state_dict(syn.synth_resnet_state_dict(syn.resnet_param_shapes((3, 4, 6, 3)), seed=1))
    model.backbone.sketch_backbone.load_state_dict(syn.synth_resnet_state_dict(syn.resnet_param_shapes((2, 2, 2, 2)), seed=2))
    model.to(DEV).train()
    crit = build_loss(args).to(DEV).train()
    n_bb = sum(1 for _ in model.backbone.parameters())
    assert n_bb == 108 + 60 and all(p.requires_grad for p in model.backbone.parameters())   # 36 + 20 convolutions, 2 BatchNorm rows each
    params = [p for p in model.parameters() if p.requires_grad]
    red = parallel.BucketedGradAllReduce(parallel.arrival_order(model), skip=parallel.unused_parameters(model), ordered=True)
    opt = parallel.FlatAdamW(red, lr=1e-3, weight_decay=1e-4, params=params)
    B, T = 1, 2
    vid = syn.synth_images(B * T, syn.vit_config(image_size=224), seed=4).view(B, T, 3, 224, 224).to(DEV)
    sk = syn.synth_images(B, syn.vit_config(image_size=224), seed=5).view(B, 1, 3, 224, 224).to(DEV)
    before = {k: v.detach().clone() for k, v in model.backbone.named_parameters()}
    red.zero_grad()
    out = model(sk, vid, torch.ones(B, 1, device=DEV), torch.ones(B, T, device=DEV))
    crit(out, syn.synth_targets(B, T, seed=1))
    loss = crit.weighted_total()
    loss.backward()
    red.finish()
    opt.step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
    still = [k for k, p in model.backbone.named_parameters() if torch.equal(p.detach(), before[k])]
    assert not still, f'{len(still)} backbone parameters did not move: {still[:5]}'
    assert int(model.backbone.video_backbone.state_dict()['1.num_batches_tracked']) == 1
    path = tmp_path / 'm.pt'
    torch.save(model.state_dict(), path)
    args2 = syn.head_args(hidden_dim=64, nheads=8, num_layers=1, num_queries=10, num_frames=2, backbone='resnet', input_dropout=0.0)
    args2.train_backbone = True
    m2 = build_model(args2)
    m2.load_state_dict(torch.load(path, map_location='cpu'), strict=True)
    m2.to(DEV)
    sd1, sd2 = model.state_dict(), m2.state_dict()
    assert list(sd1.keys()) == list(sd2.keys()) and all(torch.equal(sd1[k], sd2[k]) for k in sd1)
    model.eval(); m2.eval()
    with torch.no_grad():
        o1 = model(sk, vid, torch.ones(B, 1, device=DEV), torch.ones(B, T, device=DEV))
        o2 = m2(sk, vid, torch.ones(B, 1, device=DEV), torch.ones(B, T, device=DEV))
    assert torch.equal(o1['pred_boxes'], o2['pred_boxes'])


@pytest.mark.parametrize('depths,name', [((2, 2, 2, 2), 'resnet18'), ((3, 4, 6, 3), 'resnet34')])
def test_full_size_resnet_training_gradients(depths, name):
    """The real networks at the real resolution (2 images of 224 x 224: BatchNorm over 25 088 ... 98 samples) in training mode: features,
    running statistics and all 60 / 108 parameter gradients against the fp32 oracle on the CPU (the oracle's training mode is pinned to
    transformers.ResNetModel.train() in fp64 by the *_train fixtures) AND against a torch evaluation with the product's bf16 roundings
    on the GPU.  See the comment at the assertions for what can and cannot be concluded at this depth."""
    from oracle import resnet_oracle as R
    from svol_amd.modeling.resnet import ResNetExtractor
    sd = syn.synth_resnet_state_dict(syn.resnet_param_shapes(depths), seed=1)
    x = syn.synth_images(2, syn.vit_config(image_size=224), seed=7)
    m = ResNetExtractor(depths, compute_dtype='bf16', trainable=True)
    m.load_state_dict(sd)
    m.to(DEV).train()
    out = m(x.to(DEV))
    probe = torch.randn(out.shape, generator=torch.Generator().manual_seed(5))
    (out.float() * probe.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    probe_fmap = probe.transpose(1, 2).reshape(2, -1, 7, 7)
    f, grads, stats = R.resnet_train_grads(sd, depths, x, probe_fmap, dtype=torch.float32)
    ref_out = f.flatten(2).transpose(1, 2)
    l2 = float((out.detach().float().cpu() - ref_out).norm() / ref_out.norm())
    errs = {k: float((p.grad.float().cpu() - grads[k]).norm() / max(float(grads[k].norm()), 1e-20)) for k, p in m.named_parameters()}
    fe, ge = _emulated_bf16_reference(sd, depths, x, probe_fmap, False)
    assert not [k for k, v in ge.items() if v is None], [k for k, v in ge.items() if v is None][:8]
    l2e = float((out.detach().float() - fe.flatten(2).transpose(1, 2)).norm() / fe.norm())
    erre = {k: float((p.grad.float() - ge[k]).norm() / max(float(ge[k].norm()), 1e-20)) for k, p in m.named_parameters()}
    eme = {k: float((ge[k].float().cpu() - grads[k]).norm() / max(float(grads[k].norm()), 1e-20)) for k in grads}
    srt, srte, srtm = sorted(errs.values()), sorted(erre.values()), sorted(eme.values())
    print(f'{name}: vs fp32 oracle: feature L2 {l2:.2e}, gradient L2 max {srt[-1]:.2e} median {srt[len(srt) // 2]:.2e}; '
          f'vs the bf16-rounding reference: feature L2 {l2e:.2e}, gradient L2 max {srte[-1]:.2e} median {srte[len(srte) // 2]:.2e} (worst: {max(erre, key=erre.get)}); '
          f'the bf16-rounding reference vs fp32: gradient L2 max {srtm[-1]:.2e} median {srtm[len(srtm) // 2]:.2e}')
    for k, v in stats.items():
        have = dict(m.named_buffers())[k].float().cpu()
        assert float((have - v).abs().max()) <= 3e-2 * max(1.0, float(v.abs().max())), k
    # measured (round 5): ResNet-18 vs fp32 0.26 median / vs the rounding reference 0.20; ResNet-34 0.42 / 0.36 — and the rounding
    # reference itself is 0.43 from fp32: with random weights, 17 / 33 BatchNorm + ReLU units deep, two bf16 evaluations of the SAME
    # network disagree with each other as much as with fp32 (ReLU masks flip, BatchNorm re-normalises the difference).  So the full-size
    # check is relative: the product must be no further from fp32 than a plain torch evaluation with the same roundings is (a wiring
    # error — a missing branch, a wrong stride — gives errors of order 1); the tight checks are the kernel-by-kernel test and the toy nets.
    assert l2 <= 8e-2 and l2e <= 6e-2, (l2, l2e)
    assert srt[len(srt) // 2] <= 1.2 * srtm[len(srtm) // 2] + 0.02, (srt[len(srt) // 2], srtm[len(srtm) // 2])
    assert srt[-1] <= 1.2 * srtm[-1] + 0.05, (srt[-1], srtm[-1])


def test_resnet_training_kernels_in_deterministic_mode():
    """SVOL_DETERMINISTIC=1 re-routes the column reductions of csrc/resnet_train.hip through partial rows folded in index order and
    leaves the gathered weight-gradient GEMM unsplit (one adder per output element): the kernel-by-kernel fp64 checks and the gradient-sink
    check again, in a child process (the switch is read once), and two backward passes of one extractor must agree bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SVOL_DETERMINISTIC='1', PYTHONPATH=root)
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_resnet.py', '-q', '-m', 'gpu', '-x', '-k',
                        'test_resnet_training_kernels or weight_gradients_into_gradient_sinks', '-p', 'no:cacheprovider',
                        '--deselect', 'tests/test_gpu_resnet.py::test_resnet_training_kernels_in_deterministic_mode'],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    code = r'''
import torch
from svol_amd import synthetic as syn
from svol_amd.modeling.resnet import ResNetExtractor
sd = syn.synth_resnet_state_dict(syn.resnet_param_shapes((1, 2), (32, 64), 32), seed=1)
x = syn.synth_images(4, syn.vit_config(image_size=64), seed=5).cuda()
outs = []
for rep in range(2):
    m = ResNetExtractor((1, 2), (32, 64), 32, compute_dtype='bf16', trainable=True)
    m.load_state_dict(sd); m.cuda().train()
    out = m(x)
    (out.float() * torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).cuda()).sum().backward()
    torch.cuda.synchronize()
    outs.append([p.grad.clone() for p in m.parameters()] + [b.clone() for b in m.buffers()])
print('RESNET', 'identical' if all(torch.equal(a, b) for a, b in zip(*outs)) else 'DIFFERENT')
'''
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and 'RESNET identical' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
