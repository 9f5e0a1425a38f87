"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch tensor arithmetic, no nn.Module, no
nn.MultiheadAttention) of the SVOL hot path named by BASELINE.json
``north_star``: the SVANet head forward, the two matchers and the set
criterion.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product (``svol_amd``) never
does — it fails loudly when its HIP library is missing.

Parity is PINNED: tests/test_oracle_golden.py checks every function here
against fixtures produced by the reference itself
(tests/golden/make_golden.py imports /root/reference on CPU).

Each function cites the reference lines it follows (paths relative to the
reference root).  All functions are dtype-generic (fp32 = the reference's
precision, apex opt_level O0, configs.py:52; fp64 for arbitration).
Gradients come from torch autograd over these forward restatements, exactly
as in the reference (train.py:231-232).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lsap as _lsap

D_FF = 2048  # cross_modal_transformer.py:196-202 hard-codes dim_feedforward=2048


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def layer_norm(x, w, b, eps: float = 1e-5):
    """nn.LayerNorm over the last dim (biased variance, eps inside sqrt)."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * w + b


def linear(x, w, b=None):
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def gelu_erf(x):
    """F.gelu default (exact erf form) — cross_modal_transformer.py:189-190."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def input_proj(x, sd, prefix: str, n_input_proj: int, dropout_mask=None):
    """svanet.py:49-60 + LinearLayer.forward svanet.py:174-181:
    x -> [LN -> Dropout -> Linear -> (ReLU)] * n_input_proj, ReLU on every
    layer except index n_input_proj-1.  Dropout is identity in eval mode;
    ``dropout_mask[j]`` (already scaled by 1/(1-p)) may be supplied for train
    mode restatements."""
    relu_args = [True] * 3
    relu_args[n_input_proj - 1] = False
    for j in range(n_input_proj):
        x = layer_norm(x, sd[f'{prefix}.{j}.LayerNorm.weight'], sd[f'{prefix}.{j}.LayerNorm.bias'])
        if dropout_mask is not None:
            x = x * dropout_mask[j]
        x = linear(x, sd[f'{prefix}.{j}.net.1.weight'], sd[f'{prefix}.{j}.net.1.bias'])
        if relu_args[j]:
            x = torch.relu(x)
    return x


def position_embedding_sine(mask_bool, d: int, dtype=torch.float32, temperature: float = 10000.0):
    """position_encoding.py:51-71 with normalize=True, scale=2*pi,
    num_pos_feats = hidden_dim (position_encoding.py:101-129).  The reference
    always computes this in float32 (cumsum dtype=float32)."""
    x_embed = mask_bool.cumsum(1, dtype=torch.float32)
    x_embed = x_embed / (x_embed[:, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(d, dtype=torch.float32)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='trunc') / d)
    pos = x_embed[:, :, None] / dim_t
    pos = torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2)
    return pos.to(dtype)


def mha(q_in, k_in, v_in, in_w, in_b, out_w, out_b, h: int, key_padding_mask=None,
        need_output: bool = True, p_drop=None):
    """torch.nn.MultiheadAttention forward (batch-first restatement), as used
    4x per layer (cross_modal_transformer.py:86-97): packed in_proj, q scaled
    by d_h**-0.5 after projection, softmax over keys, out_proj.  Returns
    (output [B,Lq,d] or None, head-averaged weights [B,Lq,Lk])."""
    B, Lq, d = q_in.shape
    Lk = k_in.shape[1]
    dh = d // h
    q = linear(q_in, in_w[:d], in_b[:d])
    k = linear(k_in, in_w[d:2 * d], in_b[d:2 * d])
    q = q.view(B, Lq, h, dh).transpose(1, 2) * (1.0 / math.sqrt(dh))
    k = k.view(B, Lk, h, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)  # [B,h,Lq,Lk]
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float('-inf'))
    p = torch.softmax(s, dim=-1)
    if p_drop is not None:  # training mode: F.multi_head_attention_forward drops the softmax OUTPUT (and returns the dropped weights)
        p = p_drop(p)
    w_mean = p.mean(1)
    if not need_output:
        return None, w_mean
    v = linear(v_in, in_w[2 * d:], in_b[2 * d:]).view(B, Lk, h, dh).transpose(1, 2)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, d)
    return linear(o, out_w, out_b), w_mean


def mlp_block(x, sd, p):
    """MLP cross_modal_transformer.py:163-179: fc2(gelu(fc1(x)))."""
    return linear(gelu_erf(linear(x, sd[p + '.fc1.weight'], sd[p + '.fc1.bias'])),
                  sd[p + '.fc2.weight'], sd[p + '.fc2.bias'])


def _ln(x, sd, p):
    return layer_norm(x, sd[p + '.weight'], sd[p + '.bias'])


def _mha_p(sd, p):
    return (sd[p + '.in_proj_weight'], sd[p + '.in_proj_bias'], sd[p + '.out_proj.weight'], sd[p + '.out_proj.bias'])


def cross_modal_layer(sd, p: str, h: int, src_vid, src_skch, out, vid_pad_mask, vid_pos, query_pos):
    """CrossModalTransformerLayer.forward, cross_modal_transformer.py:105-160
    (batch-first).  Returns (mem, out)."""
    # gate (:122-127): only the head-mean weights of sketch->video attention are used
    _, att1 = mha(src_skch, src_vid + vid_pos, src_vid + vid_pos, *_mha_p(sd, p + 'sketch_video_cross_attn'), h,
                  need_output=False)  # [B,1,L]
    mem = src_vid + att1.transpose(1, 2) * src_vid
    mem = _ln(mem, sd, p + 'norm1')
    # video self-attention (:137-141) — no key padding mask
    qk = mem + vid_pos
    o, _ = mha(qk, qk, mem, *_mha_p(sd, p + 'content_self_attn'), h)
    mem = _ln(o + mem, sd, p + 'norm2')
    # MLP1 (:142-143)
    mem = _ln(mem + mlp_block(mem, sd, p + 'mlp1'), sd, p + 'norm3')
    # query self-attention (:145-149)
    qk = out + query_pos
    o, _ = mha(qk, qk, out, *_mha_p(sd, p + 'token_self_attn'), h)
    out = _ln(o + out, sd, p + 'norm4')
    # query -> video cross-attention (:151-156), key_padding_mask=True on pads
    o, _ = mha(out + query_pos, mem + vid_pos, mem, *_mha_p(sd, p + 'content_token_cross_attn'), h,
               key_padding_mask=vid_pad_mask)
    out = _ln(out + o, sd, p + 'norm5')
    # MLP2 (:157-158)
    out = _ln(out + mlp_block(out, sd, p + 'mlp2'), sd, p + 'norm6')
    return mem, out


def svanet_forward(sd: Dict[str, torch.Tensor], args, src_sketch, src_sketch_mask, src_video, src_video_mask,
                   return_hs: bool = False, dropout_masks=None):
    """SVANet.forward, svanet.py:65-141.  Eval mode by default (dropout identity); TRAINING mode (svanet.py:168-171: the only
    dropout on this path is ``input_dropout`` inside LinearLayer) when ``dropout_masks`` = {'video': [m_0, ...],
    'sketch': [m_0, ...]} is given — one keep mask per projection layer, already scaled by 1/(1-p), applied between LayerNorm
    and Linear exactly where nn.Dropout sits (svanet.py:166,178).
    ``sd`` uses the head's own state-dict keys (no ``head.`` prefix)."""
    d, h, nl = args.hidden_dim, args.nheads, args.num_layers
    dtype = src_video.dtype
    dm = dropout_masks or {}
    vid = input_proj(src_video, sd, 'input_video_proj', args.n_input_proj, dm.get('video'))
    mask_video = src_video_mask.bool()
    pos_video = position_embedding_sine(mask_video, d, dtype)
    skch = input_proj(src_sketch, sd, 'input_sketch_proj', args.n_input_proj, dm.get('sketch'))
    # transformer (cross_modal_transformer.py:27-81)
    B = vid.shape[0]
    query_pos = sd['query_embed.weight'].unsqueeze(0).expand(B, -1, -1)
    out = torch.zeros_like(query_pos)
    mem = vid
    hs = []
    for i in range(nl):
        mem, out = cross_modal_layer(sd, f'transformer.layers.{i}.', h, mem, skch, out, ~mask_video, pos_video,
                                     query_pos)
        hs.append(out)
    hs = torch.stack(hs)  # [nl,B,N,d]
    logits = linear(hs, sd['class_embed.weight'], sd['class_embed.bias'])
    x = hs
    for j in range(3):  # MLP svanet.py:144-156
        x = linear(x, sd[f'bbox_embed.layers.{j}.weight'], sd[f'bbox_embed.layers.{j}.bias'])
        if j < 2:
            x = torch.relu(x)
    boxes = torch.sigmoid(x)
    res = {'pred_logits': logits[-1], 'pred_boxes': boxes[-1]}
    if args.aux_loss:
        res['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(logits[:-1], boxes[:-1])]
    if return_hs:
        return res, hs
    return res


def expand_masks(src_sketch_mask, src_video_mask, l_sketch: int, tokens_per_frame: int):
    """SketchLocalizationModel.forward mask expansion, model.py:21-22."""
    return (src_sketch_mask.repeat_interleave(l_sketch, dim=1),
            src_video_mask.repeat_interleave(tokens_per_frame, dim=1))


# ----------------------------------------------------------------------------
# boxes (lib/utils/box_utils.py:9-61)
# ----------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def generalized_box_iou(b1, b2):
    """Pairwise [N,M] GIoU, box_utils.py:24-61 (same operation order)."""
    assert (b1[:, 2:] >= b1[:, :2]).all()
    assert (b2[:, 2:] >= b2[:, :2]).all()
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = torch.max(b1[:, None, :2], b2[:, :2])
    rb = torch.min(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    iou = inter / union
    lt = torch.min(b1[:, None, :2], b2[:, :2])
    rb = torch.max(b1[:, None, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / area


# ----------------------------------------------------------------------------
# targets
# ----------------------------------------------------------------------------
def flatten_targets(targets) -> Tuple[torch.Tensor, List[int], List[int]]:
    """Walk the nested target dicts exactly as matcher.py:62-70 / :140-148 do.
    Returns (boxes [sumM,4], boxes per video, boxes per (video,frame) from
    'num_boxes_per_frame')."""
    boxes, per_video, per_frame = [], [], []
    for tv in targets:
        per_frame.extend(tv['num_boxes_per_frame'])
        cnt = 0
        for frame in tv['bboxes'].values():
            cnt += len(frame)
            for inst in frame:
                boxes.append(inst['bbox'])
        per_video.append(cnt)
    return torch.stack(boxes), per_video, per_frame


def cost_matrix_block(logits, boxes, tgt, w_bbox, w_giou, w_class):
    """C = w_bbox*L1 + w_giou*(-GIoU) + w_class*(-p_fg) for one block
    (matcher.py:76-85 restricted to the block the reference actually uses)."""
    prob = logits.softmax(-1)
    cost_class = -prob[:, [0] * tgt.shape[0]]
    cost_bbox = torch.cdist(boxes, tgt, p=1)
    cost_giou = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tgt))
    return w_bbox * cost_bbox + w_giou * cost_giou + w_class * cost_class


@torch.no_grad()
def hungarian_matcher(args, pred_logits, pred_boxes, targets, solver=None):
    """HungarianMatcher.forward, matcher.py:131-159 (per-video LSAP)."""
    solver = solver or _lsap.linear_sum_assignment
    tgt, per_video, _ = flatten_targets(targets)
    tgt = tgt.to(pred_boxes.dtype)
    out, off = [], 0
    for b, m in enumerate(per_video):
        C = cost_matrix_block(pred_logits[b], pred_boxes[b], tgt[off:off + m], args.set_cost_bbox,
                              args.set_cost_giou, args.set_cost_class)
        r, c = solver(C.cpu().numpy())
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(c, dtype=torch.int64)))
        off += m
    return out


@torch.no_grad()
def per_frame_matcher(args, pred_logits, pred_boxes, targets, solver=None):
    """PerFrameMatcher.forward, matcher.py:38-119: one LSAP per (video, frame)
    over that frame's q queries x m boxes; prediction ids are video-local,
    target ids are batch-global box ids re-based by the per-video minimum
    MATCHED id (matcher.py:114-115 — quirk preserved)."""
    solver = solver or _lsap.linear_sum_assignment
    B, N = pred_boxes.shape[:2]
    T, q = args.num_frames, args.num_queries_per_frame
    assert N == T * q
    tgt, _, per_frame = flatten_targets(targets)
    tgt = tgt.to(pred_boxes.dtype)
    out, off = [], 0
    for b in range(B):
        pi, ti = [], []
        for t in range(T):
            m = per_frame[b * T + t]
            lo = t * q
            C = cost_matrix_block(pred_logits[b, lo:lo + q], pred_boxes[b, lo:lo + q], tgt[off:off + m],
                                  args.set_cost_bbox, args.set_cost_giou, args.set_cost_class) if m > 0 else \
                np.zeros((q, 0), np.float32)
            r, c = solver(C.cpu().numpy() if m > 0 else C)
            pi.extend((r + lo).tolist())
            ti.extend((c + off).tolist())
            off += m
        ti = np.asarray(ti, dtype=np.int64)
        ti = ti - ti.min()
        out.append((torch.as_tensor(pi, dtype=torch.int64), torch.as_tensor(ti, dtype=torch.int64)))
    return out


def match(args, pred_logits, pred_boxes, targets):
    if args.matcher == 'per_frame_matcher':
        return per_frame_matcher(args, pred_logits, pred_boxes, targets)
    if args.matcher == 'video_matcher':
        return hungarian_matcher(args, pred_logits, pred_boxes, targets)
    raise NotImplementedError  # matcher.py:178


# ----------------------------------------------------------------------------
# criterion (lib/modeling/loss.py)
# ----------------------------------------------------------------------------
def loss_labels(pred_logits, indices, eos_coef: float):
    """loss.py:39-60: weighted CE (fg=0 weight 1, bg=1 weight eos_coef),
    reduction 'none' then a PLAIN mean over B*N; class_error = 100 - top1 acc
    on matched queries (model_utils.py:5-21)."""
    B, N, _ = pred_logits.shape
    tgt = torch.ones(B, N, dtype=torch.int64)
    bi = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    si = torch.cat([s for s, _ in indices])
    tgt[bi, si] = 0
    logp = torch.log_softmax(pred_logits, -1)
    w = torch.tensor([1.0, eos_coef], dtype=pred_logits.dtype)
    nll = -logp.gather(-1, tgt[..., None]).squeeze(-1) * w[tgt]
    matched = pred_logits[bi, si]
    # topk(1) picks index 0 unless logit1 is strictly larger
    correct = (matched[:, 0] >= matched[:, 1]).to(pred_logits.dtype).sum()
    class_error = 100 - correct * (100.0 / matched.shape[0])
    return nll.mean(), class_error.detach()


def loss_boxes(pred_boxes, indices, targets):
    """loss.py:76-103: L1 mean over (#matched*4); GIoU = mean(1 - diag)."""
    per_video_boxes = []
    for tv in targets:
        bl = [inst['bbox'] for frame in tv['bboxes'].values() for inst in frame]
        per_video_boxes.append(torch.stack(bl).to(pred_boxes.dtype))
    bi = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    si = torch.cat([s for s, _ in indices])
    src = pred_boxes[bi, si]
    tgt = torch.cat([t[i] for t, (_, i) in zip(per_video_boxes, indices)], dim=0)
    l1 = (src - tgt).abs().mean()
    giou = 1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt)))
    return l1, giou.mean()


def weight_dict(args) -> Dict[str, float]:
    """build_loss, loss.py:195-202."""
    wd = {'loss_bbox': args.set_cost_bbox, 'loss_giou': args.set_cost_giou, 'loss_label': args.set_cost_class}
    if args.aux_loss:
        for i in range(args.num_layers - 1):
            wd.update({f'{k}_{i}': v for k, v in list(wd.items())[:3]})
    return wd


def set_criterion(args, outputs, targets, return_indices: bool = False, indices=None):
    """SetCriterion.forward, loss.py:126-157: match the last layer, then
    RE-MATCH every aux layer (loss.py:148-155).

    ``indices`` (test use): a list [last, aux_0, ...] of per-video (pred_idx, tgt_idx) assignments to score INSTEAD of matching —
    the matcher is @no_grad (matcher.py:38), so the losses and their gradients are functions of the outputs and a FIXED assignment;
    feeding the assignment a 16-bit device run found (bit-exact w.r.t. its own outputs, but possibly a near-tie flip away from
    the fp32 one) gives the fp32 gradient reference for exactly the loss that run differentiated."""
    losses = {}
    all_idx = []

    def one(lo, suffix):
        if indices is not None:
            idx = [(torch.as_tensor(p, dtype=torch.int64), torch.as_tensor(t, dtype=torch.int64)) for p, t in indices[len(all_idx)]]
        else:
            idx = match(args, lo['pred_logits'].detach(), lo['pred_boxes'].detach(), targets)
        all_idx.append(idx)
        ll, ce = loss_labels(lo['pred_logits'], idx, args.eos_coef)
        lb, lg = loss_boxes(lo['pred_boxes'], idx, targets)
        losses['loss_label' + suffix] = ll
        losses['class_error' + suffix] = ce
        losses['loss_bbox' + suffix] = lb
        losses['loss_giou' + suffix] = lg

    one(outputs, '')
    for i, aux in enumerate(outputs.get('aux_outputs', [])):
        one(aux, f'_{i}')
    if return_indices:
        return losses, all_idx
    return losses


def total_loss(args, loss_dict):
    """train.py:227-228."""
    wd = weight_dict(args)
    return sum(loss_dict[k] * wd[k] for k in loss_dict.keys() if k in wd)


def train_step(sd, args, inputs, targets):
    """One forward + criterion + backward on CPU (the cpu_baseline 'port').
    ``sd`` tensors must require grad.  Returns (loss_total, loss_dict)."""
    out = svanet_forward(sd, args, inputs['src_sketch'], inputs['src_sketch_mask'], inputs['src_video'],
                         inputs['src_video_mask'])
    ld = set_criterion(args, out, targets)
    tot = total_loss(args, ld)
    tot.backward()
    return tot.detach(), ld
