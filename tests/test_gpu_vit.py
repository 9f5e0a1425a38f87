"""ViT extractor on the MI355X against the oracle / the Hugging Face goldens:
short-sequence attention kernel vs fp64 math, patchify + embedding exact, whole extractor within bf16 tolerance of the
fp32 transformers output, backbone output shapes of BASELINE configs[3]."""
import ast
import glob
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, 'golden', 'vit_*.npz')))


@pytest.mark.parametrize('n,H,L,dh', [(3, 2, 5, 32), (2, 2, 17, 64), (2, 12, 197, 64), (1, 4, 256, 32), (2, 3, 33, 64)])
def test_attn_small(n, H, L, dh):
    from svol_amd import _lib
    from svol_amd.ops import _ptr, _stream
    torch.manual_seed(L)
    d = H * dh
    qkv = (torch.randn(n * L, 3 * d) * 1.2).bfloat16()
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    sp = lambda t: t.double().view(n, L, H, dh).transpose(1, 2)  # noqa: E731
    ref = (torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(dh), -1) @ sp(v)).transpose(1, 2).reshape(n * L, d)
    g = qkv.cuda()
    o = torch.empty((n * L, d), dtype=torch.bfloat16, device='cuda')
    rc = _lib.lib().svol_attn_small_fwd(_ptr(g[:, :d]), 3 * d, _ptr(g[:, d:2 * d]), 3 * d, _ptr(g[:, 2 * d:]), 3 * d, _ptr(o), d,
                                        n, H, L, dh, 1.0 / math.sqrt(dh), 1, _stream())
    assert rc == 0
    err = float((o.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-2, err


def test_patchify_and_embed_exact():
    from oracle import vit_oracle  # noqa: F401
    from svol_amd import _lib
    from svol_amd import synthetic as syn
    from svol_amd.ops import _ptr, _stream
    cfg = syn.vit_config(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128, image_size=48)
    x = syn.synth_images(2, cfg, 4)
    n, C, H, W = x.shape
    p = cfg.patch_size
    P = (H // p) * (W // p)
    out = torch.empty((n * P, C * p * p), dtype=torch.float32, device='cuda')
    assert _lib.lib().svol_patchify(_ptr(x.cuda()), _ptr(out), n, C, H, W, p, 0, _stream()) == 0
    ref = torch.nn.functional.unfold(x, kernel_size=p, stride=p).transpose(1, 2).reshape(n * P, -1)  # (c, ky, kx) order
    assert torch.equal(out.cpu(), ref)
    d = cfg.hidden_size
    proj, cls, pos = torch.randn(n * P, d), torch.randn(1, 1, d), torch.randn(1, P + 1, d)
    x32 = torch.empty((n * (P + 1), d), device='cuda')
    assert _lib.lib().svol_vit_embed(_ptr(proj.cuda()), _ptr(cls.cuda()), _ptr(pos.cuda()), _ptr(x32), 0, n, P, d, 1, _stream()) == 0
    ref = torch.cat([cls.expand(n, -1, -1), proj.view(n, P, d)], 1) + pos
    assert torch.equal(x32.cpu().view(n, P + 1, d), ref)


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(p)[4:-4] for p in FIXTURES])
def test_extractor_vs_transformers(path):
    from svol_amd import synthetic as syn
    from svol_amd.modeling.backbone import ViTExtractor
    z = np.load(path)
    cfg = syn.vit_config(**ast.literal_eval(str(z['over'])))
    st = int(z['stride'])
    m = ViTExtractor(cfg)
    m.load_state_dict(syn.synth_vit_state_dict(cfg, int(z['seed'])))
    m = m.cuda().eval()
    x = syn.synth_images(int(z['n']), cfg, int(z['seed'])).cuda()
    last, pre = m(x, return_pre_norm=True)
    ref = torch.from_numpy(z['last_hidden_state'])
    # bf16 operands, fp32 residual stream: 1e-2 of the output scale (LayerNorm outputs reach |x| ~ 4-5 with these
    # synthetic gains; bf16 operand rounding is 2^-9 per GEMM input)
    e_last = float((last.cpu()[:, ::st] - ref).abs().max()) / float(ref.abs().max())
    refp = torch.from_numpy(z['pre_norm'])
    e_pre = float((pre.cpu()[:, ::st] - refp).abs().max()) / float(refp.abs().max())
    print(f'vit extractor rel err: last {e_last:.2e} pre-norm {e_pre:.2e}')
    assert e_last < 1e-2 and e_pre < 1e-2


def test_backbone_shapes_and_model_path():
    """BASELINE configs[3] wiring: frames + sketch -> ViT features -> head, through build_model(--backbone vit)."""
    from svol_amd import synthetic as syn
    from svol_amd.modeling.backbone import ViTBackbone, ViTExtractor
    from svol_amd.modeling.model import SketchLocalizationModel
    from svol_amd.modeling.svanet import build_svanet
    cfg = syn.vit_config(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128, image_size=32)
    B, T = 2, 4
    bb = ViTBackbone(ViTExtractor(cfg), ViTExtractor(cfg)).cuda()
    sk, vd = bb(torch.randn(B, 1, 3, 32, 32, device='cuda'), torch.randn(B, T, 3, 32, 32, device='cuda'))
    assert sk.shape == (B, 1, 64) and vd.shape == (B, T * 4, 64)
    args = syn.head_args(hidden_dim=64, nheads=8, num_layers=1, num_queries=10, num_frames=T, input_vid_dim=64, input_skch_dim=64,
                         compute_dtype='bf16')
    model = SketchLocalizationModel(bb, build_svanet(args)).cuda().eval()
    out = model(torch.randn(B, 1, 3, 32, 32, device='cuda'), torch.randn(B, T, 3, 32, 32, device='cuda'),
                torch.ones(B, 1, device='cuda'), torch.ones(B, T, device='cuda'))
    assert out['pred_boxes'].shape == (B, 10, 4) and torch.isfinite(out['pred_logits']).all()
