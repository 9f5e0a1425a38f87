"""GPU parity of the whole hot path (head forward + matcher + criterion + backward) through the
product modules against the reference's golden vectors: 1e-3 fp32 / 1e-2 bf16 on outputs and
losses (north_star), Hungarian assignment bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# cfg2_b1_video(_pad): BASELINE configs[1] at full depth / width / sequence length (L = 6272, 6 layers, d = 256, N = 100), one
# video, without and with 8 padded frames — reference-generated values at the shapes the bench launches
CASES = ['tiny_video', 'tiny_frame', 'cfg1_video', 'cfg1_frame', 'mid_video', 'mid32_video', 'cfg2_b1_video', 'cfg2_b1_video_pad']


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_head_golden(name, dtype):
    from tests import gpu_checks as G
    res = G.check_head_case(name, dtype)
    bad = {k: v for k, v in res.items() if not (v[0] <= v[1])}
    assert not bad, '\n'.join(f'{k}: err={e:.3e} tol={t:.1e}' for k, (e, t) in bad.items())


@pytest.mark.parametrize('name', ['cfg1_video', 'cfg1_frame'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_head_golden_gradient_sinks(name, dtype):
    """Same parity bar with the gradients owned by BucketedGradAllReduce: weight / bias / LayerNorm gradient
    kernels accumulate straight into the flat buckets and report completion to the reducer."""
    from tests import gpu_checks as G
    res = G.check_head_case(name, dtype, sinks=True)
    bad = {k: v for k, v in res.items() if not (v[0] <= v[1])}
    assert not bad, '\n'.join(f'{k}: err={e:.3e} tol={t:.1e}' for k, (e, t) in bad.items())


@pytest.mark.parametrize('name', ['tiny_video', 'mid_video', 'mid32_video'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_model_level_mask_expansion_matches_the_head_golden(name, dtype):
    """SURVEY §8 a1 (model.py:16-28) by VALUE: the reference-generated head goldens reached through
    ``build_model`` / ``SketchLocalizationModel.forward`` — features [B,T,P,Din], one mask entry per FRAME ([B,T], padded
    frames = 0) and per sketch ([B,1]), expanded per token inside the model (``repeat_interleave``) — must give the golden
    outputs of the head fed with per-token masks, to the same bar (1e-3 fp32 / 1e-2 bf16)."""
    from oracle import svol_oracle as O
    from svol_amd.modeling.model import build_model
    from tests.helpers import head_case
    z, meta, args, sd, inp, tg = head_case(name)
    B, T, P = meta['B'], meta['T'], meta['P']
    assert meta['pad_frames'] > 0
    args.backbone = 'features'
    args.compute_dtype = 'fp32' if dtype == torch.float32 else 'bf16'
    model = build_model(args)
    assert all(k.startswith(('head.', 'backbone.')) for k in model.state_dict())
    model.load_state_dict({'head.' + k: v for k, v in sd.items()}, strict=True)
    model = model.cuda().eval()
    frame_mask = inp['src_video_mask'].view(B, T, P)[:, :, 0].contiguous()          # [B,T]
    assert bool((frame_mask == 0).any())
    # the oracle's restatement of the expansion reproduces the per-token masks the golden was generated with
    sm, vm = O.expand_masks(inp['src_sketch_mask'], frame_mask, 1, P)
    assert torch.equal(vm, inp['src_video_mask']) and torch.equal(sm, inp['src_sketch_mask'])
    out = model(src_sketch=inp['src_sketch'].cuda(), src_video=inp['src_video'].view(B, T, P, -1).cuda(),
                src_sketch_mask=inp['src_sketch_mask'].cuda(), src_video_mask=frame_mask.cuda())
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    e_l = float((out['pred_logits'].cpu() - torch.from_numpy(z['pred_logits'])).abs().max())
    e_b = float((out['pred_boxes'].cpu() - torch.from_numpy(z['pred_boxes'])).abs().max())
    assert e_l <= tol and e_b <= tol, (e_l, e_b)
    # and the mask really is honoured: without it the padded videos' outputs move
    out2 = model(src_sketch=inp['src_sketch'].cuda(), src_video=inp['src_video'].view(B, T, P, -1).cuda(),
                 src_sketch_mask=inp['src_sketch_mask'].cuda(), src_video_mask=torch.ones(B, T).cuda())
    assert float((out2['pred_logits'] - out['pred_logits'])[1::2].abs().max()) > 1e-4
    assert float((out2['pred_logits'] - out['pred_logits'])[0::2].abs().max()) <= (1e-5 if dtype == torch.float32 else 1e-2)
