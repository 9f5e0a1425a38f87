"""GPU parity of the whole hot path (head forward + matcher + criterion + backward) through the
product modules against the reference's golden vectors: 1e-3 fp32 / 1e-2 bf16 on outputs and
losses (north_star), Hungarian assignment bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# cfg2_b1_video(_pad): BASELINE configs[1] at full depth / width / sequence length (L = 6272, 6 layers, d = 256, N = 100), one
# video, without and with 8 padded frames — reference-generated values at the shapes the bench launches
# cfg2w_b1_frame: per_frame_matcher at the shipped recipe's scale (32 frames x 10 queries = 320) through the 6-layer d = 256 head
CASES = ['tiny_video', 'tiny_frame', 'cfg1_video', 'cfg1_frame', 'mid_video', 'mid32_video', 'cfg2_b1_video', 'cfg2_b1_video_pad',
         'cfg2w_b1_frame']


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_head_golden(name, dtype):
    from tests import gpu_checks as G
    res = G.check_head_case(name, dtype)
    bad = {k: v for k, v in res.items() if not (v[0] <= v[1])}
    assert not bad, '\n'.join(f'{k}: err={e:.3e} tol={t:.1e}' for k, (e, t) in bad.items())


@pytest.mark.parametrize('name', ['cfg1_video', 'cfg1_frame', 'mid32_video', 'cfg2_b1_video', 'cfg2_b1_video_pad', 'cfg2w_b1_frame'])
def test_head_golden_fp16(name):
    """fp16 operands (``--compute_dtype fp16``, BASELINE configs[4]'s stated dtype): the same goldens, outputs within 3e-3."""
    from tests import gpu_checks as G
    res = G.check_head_case(name, torch.float16)
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


def _device_input_dropout_masks(model, B, L, p):
    """the keep masks (scaled by 1 / (1 - p)) the product's LinearLayers drew in the training forward that just ran: stateless
    functions of (seed, row, column) — regenerated by the same LayerNorm kernel over gamma = 0, beta = 1 — and checked against the
    numpy twin of the generator."""
    import numpy as np
    from svol_amd import ops
    from tests.gpu_checks import dropout_keep_numpy
    out = {}
    for which, seq, salt, rows in (('video', model.input_video_proj, 0, B * L), ('sketch', model.input_sketch_proj, 1, B)):
        ms = []
        for j, layer in enumerate(seq):
            D = layer.LayerNorm.weight.numel()
            seed = (model.base_seed << 32) + (model._step << 8) + salt * 16 + j
            x = torch.randn(rows, D, device='cuda')
            _, y, _, _, _ = ops.layernorm_fwd(x, torch.zeros(D, device='cuda'), torch.ones(D, device='cuda'), torch.float32, None, p, seed)
            y = y.cpu()
            keep = dropout_keep_numpy((rows, D), p, seed)
            assert np.array_equal(y.numpy() > 0, keep), (which, j)
            assert float((y[y > 0] - 1 / (1 - p)).abs().max()) < 1e-6
            ms.append(y.view(B, rows // B, D))
        out[which] = ms
    return out


@pytest.mark.parametrize('name', ['train_cfg1_video', 'train_mid32_video'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['fp32', 'bf16', 'fp16'])
def test_training_mode_input_dropout_matches_the_oracle_under_shared_masks(name, dtype):
    """The mode bench.py TIMES: SVANet in ``.train()`` with ``input_dropout`` 0.4 (svanet.py:168-181; configs.py:127).  One training
    step (forward + criterion + backward) of the product against the oracle's training-mode forward under the SAME keep masks.
    The oracle's placement of the dropout is pinned against masks recorded from the reference's own nn.Dropout modules
    (tests/test_oracle_golden.py::test_training_mode_head_under_the_masks_the_reference_drew, same two configurations); the
    device's masks are stateless functions of (seed, row, column) regenerated here.  Bars: north_star's 1e-3 fp32 / 1e-2 bf16
    (3e-3 fp16) on logits and boxes of every layer, the matched loss, every parameter gradient (tests/gpu_checks.py::compare_param_grads)."""
    from types import SimpleNamespace
    from oracle import svol_oracle as O
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.svanet import build_svanet
    from tests import gpu_checks as G
    from tests.helpers import head_case
    z, meta, args, sd, inp, tg = head_case(name)
    p = args.input_dropout
    assert p == 0.4
    args.compute_dtype = {torch.float32: 'fp32', torch.bfloat16: 'bf16', torch.float16: 'fp16'}[dtype]
    model = build_svanet(args)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().train()
    crit = build_loss(args).cuda().train()
    a = [inp[k].cuda() for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')]
    out = model(*a)
    ld = crit(out, tg)
    tot = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
    if dtype == torch.float16:   # fp16 trains under a loss scale (gpu_checks.run_head_case)
        (tot * G.FP16_LOSS_SCALE).backward()
        with torch.no_grad():
            for p_ in model.parameters():
                if p_.grad is not None:
                    p_.grad.mul_(1.0 / G.FP16_LOSS_SCALE)
    else:
        tot.backward()
    torch.cuda.synchronize()
    masks = _device_input_dropout_masks(model, meta['B'], meta['T'] * meta['P'], p)
    keep = torch.cat([m.reshape(-1) for m in masks['video']])
    assert abs(float((keep > 0).float().mean()) - (1 - p)) < 0.02
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.svanet_forward(sdr, args, inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'],
                           dropout_masks=masks)
    dev_idx = [[(pi.cpu().numpy(), ti.cpu().numpy()) for pi, ti in lay] for lay in crit.last_indices()]   # [aux0.., last]
    r_ld, r_idx = O.set_criterion(SimpleNamespace(**vars(args)), ref, tg, return_indices=True)              # [last, aux0..]
    flips = sum(int(pi.tolist() != rp.tolist() or ti.tolist() != rt.tolist())
                for lay, rlay in zip(dev_idx[-1:] + dev_idx[:-1], r_idx) for (pi, ti), (rp, rt) in zip(lay, rlay))
    if dtype == torch.float32:
        assert flips == 0
    elif flips:   # a 16-bit near-tie flip: differentiate the loss the device differentiated
        r_ld = O.set_criterion(SimpleNamespace(**vars(args)), ref, tg, indices=dev_idx[-1:] + dev_idx[:-1])
    r_tot = O.total_loss(args, r_ld)
    r_tot.backward()
    tol = {torch.float32: 1e-3, torch.bfloat16: 1e-2, torch.float16: 3e-3}[dtype]
    lays = list(out.get('aux_outputs', [])) + [out]
    rlays = list(ref.get('aux_outputs', [])) + [ref]
    e_l = max(float((o['pred_logits'].detach().cpu() - r['pred_logits']).abs().max()) for o, r in zip(lays, rlays))
    e_b = max(float((o['pred_boxes'].detach().cpu() - r['pred_boxes']).abs().max()) for o, r in zip(lays, rlays))
    # an eval-mode forward of the same weights is far away: the masks really are applied
    model.eval()
    with torch.no_grad():
        ev = model(*a)
    assert float((ev['pred_logits'].cpu() - ref['pred_logits']).abs().max()) > 1e-2
    print(f'training-mode {name} {dtype}: |dlogits| {e_l:.2e} |dboxes| {e_b:.2e} |dloss| {abs(float(tot) - float(r_tot)):.2e} flips {flips}')
    assert e_l <= tol and e_b <= tol, (e_l, e_b)
    assert abs(float(tot) - float(r_tot)) <= tol * max(1.0, abs(float(r_tot)))
    res = G.compare_param_grads(f'train/{name}', model, {k: v.grad for k, v in sdr.items()}, dtype, G.grad_bars(dtype, args.hidden_dim))
    bad = {k: v for k, v in res.items() if not (v[0] <= v[1])}
    assert not bad, '\n'.join(f'{k}: err={e:.3e} tol={t:.1e}' for k, (e, t) in bad.items())


@pytest.mark.parametrize('lag', ['side', 'main'])
def test_forward_does_not_depend_on_which_stream_lags(lag):
    """The object-query half runs on a side stream.  Stall one of the two streams in front of the forward (a spin kernel) so
    that the other runs far ahead of it: the outputs must be bit-identical to an undisturbed forward.  (Regression: the
    initial query state was cloned on the side stream from main-stream zeros that the host dropped right away; with the side
    stream lagging, the video half reused and overwrote that memory before the clone had read it — garbage logits in about
    one run of the suite in fifteen.)"""
    from svol_amd.modeling import cross_modal_transformer as cmt
    from svol_amd.modeling.svanet import build_svanet
    from tests.helpers import head_case
    z, meta, args, sd, inp, tg = head_case('cfg1_video')
    args.compute_dtype = 'bf16'
    dev = torch.device('cuda', 0)
    model = build_svanet(args)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    x = [inp[k].to(dev) for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')]
    with torch.no_grad():
        ref = model(*x)
        torch.cuda.synchronize()
        side = cmt._side_stream(dev)
        for _ in range(3):
            stalled = side if lag == 'side' else torch.cuda.current_stream()
            with torch.cuda.stream(stalled):
                torch.cuda._sleep(200_000_000)   # ~0.1 s: the other stream gets through the whole forward meanwhile
            out = model(*x)
            torch.cuda.synchronize()
            assert torch.equal(out['pred_logits'], ref['pred_logits']) and torch.equal(out['pred_boxes'], ref['pred_boxes'])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_dq_image_zeroed_beside_the_previous_layers_attention_backward_changes_nothing(dtype):
    """Round 6: the video half's backward can alternate between TWO attention workspaces and zero the next layer's fp32 dQ image on a
    side stream beside its own single-pass kernel (blocks.DQ_PREZERO / SVOL_DQ_PREZERO=1, svol_attn_bwd_ex + svol_attn_bwd_zero_ws
    through the EV_CLEAN_IN / EV_PREP / ATTN_WS_NEXT / ZERO_STREAM / EV_CLEAN_OUT slots; off by default: +-0 in the step, see
    svol_amd/blocks.py).  At the benchmarked depth and length (cfg2, B = 1: six layers, L = 6272 — the launch the single pass serves):
    forward outputs and losses bit-identical with the switch on and off; gradients to 2e-2 of each parameter's largest entry (float
    atomics land in run-to-run order, 16-bit rounding behind them) over TWO backward passes, so that the second one starts from an
    image the first pass's last layer cleaned; the state must really have alternated and ended clean."""
    from svol_amd import blocks
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.svanet import build_svanet
    from tests.helpers import head_case
    assert blocks.ENABLED
    default = blocks.DQ_PREZERO
    z, meta, args, sd, inp, tg = head_case('cfg2_b1_video')
    args.compute_dtype = {torch.bfloat16: 'bf16', torch.float16: 'fp16'}[dtype]
    dev = torch.device('cuda', 0)
    model = build_svanet(args)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    crit = build_loss(args).to(dev).eval()
    x = [inp[k].to(dev) for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')]

    def run():
        model.zero_grad(set_to_none=True)
        out = model(*x)
        ld = crit(out, tg)
        tot = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        tot.backward()
        torch.cuda.synchronize()
        return out, float(tot), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    try:
        blocks.DQ_PREZERO = False
        off, tot_off, g_off = run()
        blocks.DQ_PREZERO = True
        blocks._DQ_PREZERO.clear()
        on1, tot_on1, g_on1 = run()
        states = [st for st in blocks._DQ_PREZERO.values() if st]
        assert len(states) == 1, 'the single pass did not serve the launch: nothing was tested'
        st = states[0]
        n = args.num_layers
        assert st.nxt == n % 2 and st.clean[st.nxt] and not st.clean[1 - st.nxt]   # n launches alternated; the next one's image is clean
        on2, tot_on2, g_on2 = run()                                                  # ... and this pass starts from it
        assert st.nxt == (2 * n) % 2 and st.clean[st.nxt]
    finally:
        blocks.DQ_PREZERO = default
        blocks._DQ_PREZERO.clear()
    for on, tot_on, g_on in ((on1, tot_on1, g_on1), (on2, tot_on2, g_on2)):
        for a, b in zip(list(on.get('aux_outputs', [])) + [on], list(off.get('aux_outputs', [])) + [off]):
            assert torch.equal(a['pred_logits'], b['pred_logits']) and torch.equal(a['pred_boxes'], b['pred_boxes'])
        assert tot_on == tot_off
        assert g_on.keys() == g_off.keys()
        # per parameter against its own largest entry — plus a floor of 1e-4 of the largest gradient entry of the model: at this depth
        # some parameters (layer 0's query self-attention in_proj: the object queries enter as zeros) have gradients of 1e-9 that are
        # rounding noise of two identical runs already; and norm-wise over the whole gradient
        gmax = max(float(g_off[k].abs().max()) for k in g_off)
        num = den = 0.0
        for k in g_on:
            scale = float(g_off[k].abs().max())
            assert float((g_on[k] - g_off[k]).abs().max()) <= 2e-2 * scale + 1e-4 * gmax, k
            num += float((g_on[k].double() - g_off[k].double()).pow(2).sum())
            den += float(g_off[k].double().pow(2).sum())
        assert (num / den) ** 0.5 <= 1e-2, (num / den) ** 0.5


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16], ids=['fp32', 'bf16', 'fp16'])
def test_gate_scores_from_the_previous_layers_layernorm_change_nothing(dtype):
    """Round 6: layer i's last LayerNorm can also write layer i + 1's gate scores (blocks.GATE_SCORES_FUSE / SVOL_GATE_SCORES_FUSE=1,
    svol_layernorm_gate_scores_fwd; off by default — slower inside the step, see svol_amd/blocks.py) and that layer then skips its
    own score pass.  Same arithmetic in the same order: the head's
    outputs are BIT-identical with the fused launch on and off, and so are the losses; the gradients (the backward kernels and
    what they read are unchanged, but float atomics land in run-to-run order and 16-bit modes round behind them) agree to
    1e-4 (fp32) / 2e-2 (16-bit) of each parameter's largest gradient entry.  The fused program must really have run (trace of the
    launches), and a video length that is not a multiple of 4 must quietly take the stand-alone pass."""
    from svol_amd import blocks
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.svanet import build_svanet
    from tests.helpers import head_case
    assert blocks.ENABLED
    default = blocks.GATE_SCORES_FUSE
    z, meta, args, sd, inp, tg = head_case('mid32_video')
    args.compute_dtype = {torch.float32: 'fp32', torch.bfloat16: 'bf16', torch.float16: 'fp16'}[dtype]
    dev = torch.device('cuda', 0)
    model = build_svanet(args)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    crit = build_loss(args).to(dev).eval()
    x = [inp[k].to(dev) for k in ('src_sketch', 'src_sketch_mask', 'src_video', 'src_video_mask')]
    L = x[2].shape[1] * (x[2].shape[2] if x[2].dim() == 4 else 1)

    def run():
        model.zero_grad(set_to_none=True)
        out = model(*x)
        ld = crit(out, tg)
        tot = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        tot.backward()
        torch.cuda.synchronize()
        return out, float(tot), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    calls = []
    real = blocks.VideoHalfFn.apply
    def spy(pl, x32, pos, u, u_next, sc_in, *params):
        calls.append((u_next is not None, sc_in is not None))
        return real(pl, x32, pos, u, u_next, sc_in, *params)
    blocks.VideoHalfFn.apply = spy
    try:
        blocks.GATE_SCORES_FUSE = True
        on, tot_on, g_on = run()
        fused_calls = list(calls)
        blocks.GATE_SCORES_FUSE = False
        calls.clear()
        off, tot_off, g_off = run()
        plain_calls = list(calls)
    finally:
        blocks.GATE_SCORES_FUSE = default
        blocks.VideoHalfFn.apply = real
    n = args.num_layers
    if L % 4 == 0:
        assert fused_calls == [(i + 1 < n, i > 0) for i in range(n)], fused_calls
    assert plain_calls == [(False, False)] * n, plain_calls
    for a, b in zip(list(on.get('aux_outputs', [])) + [on], list(off.get('aux_outputs', [])) + [off]):
        assert torch.equal(a['pred_logits'], b['pred_logits']) and torch.equal(a['pred_boxes'], b['pred_boxes'])
    assert tot_on == tot_off
    assert g_on.keys() == g_off.keys()
    for k in g_on:
        scale = float(g_off[k].abs().max())
        assert float((g_on[k] - g_off[k]).abs().max()) <= (1e-4 if dtype == torch.float32 else 2e-2) * scale + 1e-12, k
