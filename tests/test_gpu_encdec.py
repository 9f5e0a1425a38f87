"""GPU parity of the enc/dec Transformer heads (svanet_variants' three fusion modes x post-/pre-norm, sketch_detr) through
the product modules against the reference's golden vectors (tests/golden/make_golden_encdec.py): 1e-3 fp32 / 1e-2 bf16."""
import pytest
import torch

from tests.helpers import ENCDEC_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ENCDEC_CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_encdec_golden(name, dtype):
    from tests import gpu_checks as G
    res = G.check_encdec_case(name, dtype)
    bad = {k: v for k, v in res.items() if not (v[0] <= v[1])}
    assert not bad, '\n'.join(f'{k}: err={e:.3e} tol={t:.1e}' for k, (e, t) in bad.items())


def test_training_mode_dropout_is_refused():
    from svol_amd import synthetic as syn
    from svol_amd.modeling.svanet_variants import build_svanet
    args = syn.encdec_args(dropout=0.1)
    model = build_svanet(args).cuda().train()
    inp = syn.synth_encdec_inputs(args, 2, 16, 2)
    with pytest.raises(NotImplementedError):
        model(*(inp[k].cuda() for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')))
    args = syn.encdec_args(dropout=0.0, input_dropout=0.0)
    model = build_svanet(args).cuda().train()
    out = model(*(inp[k].cuda() for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')))
    assert out['pred_logits'].shape == (2, args.num_queries, 2)


def test_sketch_detr_criterion_is_per_frame():
    """loss.py:159-190: with --sketch_head sketch_detr the criterion returns one loss dict per frame output, each matched
    against the video's targets; checked against the oracle criterion run on the same outputs."""
    from types import SimpleNamespace
    from oracle import svol_oracle as O
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.sketch_detr import build_sketchdetr
    args = syn_args = None
    from svol_amd import synthetic as syn
    args = syn.encdec_args(dropout=0.0, sketch_head='sketch_detr', matcher='video_matcher', num_layers=2)
    torch.manual_seed(1)
    model = build_sketchdetr(args).cuda().eval()
    crit = build_loss(args).cuda().eval()
    B, T = 2, 3
    inp = syn.synth_encdec_inputs(args, B, T, 1)
    tg = syn.synth_targets(B, T, seed=1)
    outs = model(*(inp[k].cuda() for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')))
    lds = crit(outs, tg)
    assert isinstance(lds, list) and len(lds) == T
    oargs = SimpleNamespace(**vars(args))
    for o, ld in zip(outs, lds):
        cpu = {'pred_logits': o['pred_logits'].detach().cpu(), 'pred_boxes': o['pred_boxes'].detach().cpu(),
               'aux_outputs': [{k: v.detach().cpu() for k, v in a.items()} for a in o['aux_outputs']]}
        ref = O.set_criterion(oargs, cpu, tg)
        assert sorted(ref.keys()) == sorted(ld.keys())
        for k, v in ref.items():
            assert abs(float(ld[k]) - float(v)) <= 2e-5 * max(1.0, abs(float(v))), k
    total = sum(sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict) for ld in lds)
    total.backward()
    assert model.query_embed.weight.grad is not None and bool(torch.isfinite(model.query_embed.weight.grad).all())


def test_weighted_total_equals_the_python_sum():
    from svol_amd import synthetic as syn
    from svol_amd.modeling.loss import build_loss
    args = syn.head_args(num_layers=3, num_queries=12, num_frames=4)
    crit = build_loss(args).cuda()
    lg, bx = syn.synth_head_outputs(2 * 3, 12, seed=3)
    tg = syn.synth_targets(2, 4, seed=2)
    res = []
    for fused in (False, True):
        l_ = lg.view(3, 2, 12, 2).cuda().requires_grad_(True)
        b_ = bx.view(3, 2, 12, 4).cuda().requires_grad_(True)
        out = {'pred_logits': l_[-1], 'pred_boxes': b_[-1], 'aux_outputs': [{'pred_logits': l_[i], 'pred_boxes': b_[i]} for i in range(2)]}
        ld = crit(out, tg)
        tot = crit.weighted_total() if fused else sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        tot.backward()
        res.append((float(tot), l_.grad.clone(), b_.grad.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[0][0])
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-6 and float((res[0][2] - res[1][2]).abs().max()) <= 1e-6
