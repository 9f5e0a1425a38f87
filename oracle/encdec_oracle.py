"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch tensor arithmetic on state-dict tensors, no nn.Module) of the reference's enc/dec
``Transformer`` and the heads built on it — SURVEY.md §8 f2:

  transformer_forward     lib/modeling/transformer.py:43-81 (encoder :94-116, decoder :128-160, layers :175-208, :229-283)
  svanet_variant_forward  lib/modeling/svanet_variants.py:74-247 (concat_to_seq / append_to_seq / concat_to_qry)
  sketch_detr_forward     lib/modeling/sketch_detr.py:47-75

Eval-mode semantics by default (every dropout is the identity); with ``drop=MaskFeed(...)`` the training-mode forward of the
Transformer with GIVEN keep masks, consumed in the order the reference calls dropout (attention probabilities inside
nn.MultiheadAttention, dropout1, [cross-attention probabilities, dropout2,] FFN dropout, dropout2 / dropout3 —
transformer.py:165-208,225-283).  Only ``tests/`` may import this module; the product
(``svol_amd``) never does.  Parity is PINNED: tests/test_oracle_encdec.py checks these functions against fixtures the
reference itself produced (tests/golden/make_golden_encdec.py imports /root/reference on CPU).
"""
from __future__ import annotations

import torch

from .svol_oracle import _ln, _mha_p, input_proj, layer_norm, linear, mha, position_embedding_sine  # noqa: F401


class MaskFeed:
    """keep masks (already scaled: 0 or 1/(1-p)) handed out in call order; the tensor shapes must match the calls.
    seq_first: the masks of the [B,L,d] activations are stored in the reference's [L,B,d] element order (masks recorded from the
    reference itself); the attention-probability masks are [B*h, Lq, Lk] = batch-first in both worlds."""

    def __init__(self, masks, seq_first=False):
        self.masks, self.i, self.seq_first = list(masks), 0, seq_first

    def __call__(self, x):
        m = self.masks[self.i]
        self.i += 1
        assert m.numel() == x.numel(), (self.i - 1, tuple(m.shape), tuple(x.shape))
        if self.seq_first and x.dim() == 3:
            B, L, D = x.shape
            m = m.reshape(L, B, D).transpose(0, 1)
        return x * m.reshape(x.shape).to(x.dtype)


def _id(x):
    return x


def _act(x, activation: str):
    if activation == 'relu':
        return torch.relu(x)
    if activation == 'gelu':
        return torch.nn.functional.gelu(x)
    raise RuntimeError(activation)


def _ffn(x, sd, p, activation, dr=_id):
    return linear(dr(_act(linear(x, sd[p + 'linear1.weight'], sd[p + 'linear1.bias']), activation)),
                  sd[p + 'linear2.weight'], sd[p + 'linear2.bias'])


def encoder_layer(sd, p, h, src, pad_mask, pos, pre_norm, activation='relu', drop=None):
    """TransformerEncoderLayer.forward_post / forward_pre, transformer.py:175-208 (batch-first).  drop: MaskFeed or None."""
    dr = drop if drop is not None else _id
    if not pre_norm:
        qk = src + pos
        o, _ = mha(qk, qk, src, *_mha_p(sd, p + 'self_attn'), h, key_padding_mask=pad_mask, p_drop=drop)
        src = _ln(src + dr(o), sd, p + 'norm1')
        return _ln(src + dr(_ffn(src, sd, p, activation, dr)), sd, p + 'norm2')
    s2 = _ln(src, sd, p + 'norm1')
    qk = s2 + pos
    o, _ = mha(qk, qk, s2, *_mha_p(sd, p + 'self_attn'), h, key_padding_mask=pad_mask, p_drop=drop)
    src = src + dr(o)
    return src + dr(_ffn(_ln(src, sd, p + 'norm2'), sd, p, activation, dr))


def decoder_layer(sd, p, h, tgt, memory, pad_mask, pos, query_pos, pre_norm, activation='relu', drop=None):
    """TransformerDecoderLayer.forward_post / forward_pre, transformer.py:229-283.  Returns (tgt, att [B,N,L])."""
    dr = drop if drop is not None else _id
    if not pre_norm:
        qk = tgt + query_pos
        o, _ = mha(qk, qk, tgt, *_mha_p(sd, p + 'self_attn'), h, p_drop=drop)
        tgt = _ln(tgt + dr(o), sd, p + 'norm1')
        o, att = mha(tgt + query_pos, memory + pos, memory, *_mha_p(sd, p + 'multihead_attn'), h, key_padding_mask=pad_mask,
                     p_drop=drop)
        tgt = _ln(tgt + dr(o), sd, p + 'norm2')
        return _ln(tgt + dr(_ffn(tgt, sd, p, activation, dr)), sd, p + 'norm3'), att
    t2 = _ln(tgt, sd, p + 'norm1')
    qk = t2 + query_pos
    o, _ = mha(qk, qk, t2, *_mha_p(sd, p + 'self_attn'), h, p_drop=drop)
    tgt = tgt + dr(o)
    t2 = _ln(tgt, sd, p + 'norm2')
    o, att = mha(t2 + query_pos, memory + pos, memory, *_mha_p(sd, p + 'multihead_attn'), h, key_padding_mask=pad_mask, p_drop=drop)
    tgt = tgt + dr(o)
    return tgt + dr(_ffn(_ln(tgt, sd, p + 'norm3'), sd, p, activation, dr)), att


def transformer_forward(sd, prefix, args, src, pad_mask, query_embed, pos_embed, activation='relu', drop=None):
    """Transformer.forward, transformer.py:43-81, with return_intermediate_dec=True (build_transformer :321).
    src [B,L,d]; pad_mask [B,L] bool True on pads; query_embed [N,d] or [N,B,d]; pos_embed [B,L,d].
    Returns (hs [n_dec,B,N,d], memory [B,L,d], att [n_dec,B,N,L])."""
    h, pre = args.nheads, args.pre_norm
    B = src.shape[0]
    if query_embed.dim() == 3:
        query_pos = query_embed.transpose(0, 1)
    else:
        query_pos = query_embed.unsqueeze(0).expand(B, -1, -1)
    memory = src
    for i in range(args.enc_layers):
        memory = encoder_layer(sd, f'{prefix}encoder.layers.{i}.', h, memory, pad_mask, pos_embed, pre, activation, drop)
    if pre:  # encoder_norm only exists with normalize_before (:26)
        memory = _ln(memory, sd, prefix + 'encoder.norm')
    tgt = torch.zeros_like(query_pos)
    hs, atts = [], []
    for i in range(args.dec_layers):
        tgt, att = decoder_layer(sd, f'{prefix}decoder.layers.{i}.', h, tgt, memory, pad_mask, pos_embed, query_pos, pre,
                                 activation, drop)
        hs.append(_ln(tgt, sd, prefix + 'decoder.norm'))  # the shared decoder norm on every layer's output (:139-147)
        atts.append(att)
    return torch.stack(hs), memory, torch.stack(atts)


def _heads(sd, args, hs):
    logits = linear(hs, sd['class_embed.weight'], sd['class_embed.bias'])
    x = hs
    for j in range(3):
        x = linear(x, sd[f'bbox_embed.layers.{j}.weight'], sd[f'bbox_embed.layers.{j}.bias'])
        if j < 2:
            x = torch.relu(x)
    boxes = torch.sigmoid(x)
    res = {'pred_logits': logits[-1], 'pred_boxes': boxes[-1]}
    if args.aux_loss:
        res['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(logits[:-1], boxes[:-1])]
    return res


def _sketch_queries(sd, args, src_sketch, bs):
    """cat[query_embed, sketch] -> input_query_proj  (svanet_variants.py:215-219, sketch_detr.py:57-60) -> [N,B,d]."""
    qe = sd['query_embed.weight']
    nq = qe.shape[0]
    sk = src_sketch.repeat(1, nq, 1).permute(1, 0, 2)
    query = torch.cat([qe.unsqueeze(1).repeat(1, bs, 1), sk], dim=2)
    return input_proj(query, sd, 'input_query_proj', args.n_input_proj)


def svanet_variant_forward(sd, args, src_sketch, src_sketch_mask, src_video, src_video_mask, return_hs=False, drop=None):
    """SVANet.forward of svanet_variants.py (:74-84 dispatch; :86-134, :136-188, :190-247)."""
    d = args.hidden_dim
    dtype = src_video.dtype
    query = sd['query_embed.weight']
    if args.mode == 'concat_to_seq':
        sk = src_sketch.repeat(1, src_video.shape[1], 1)
        src = input_proj(torch.cat([sk, src_video], dim=2), sd, 'input_proj', args.n_input_proj)
        mask = src_video_mask.bool()
        pos = position_embedding_sine(mask, d, dtype)
    elif args.mode == 'append_to_seq':
        skch = input_proj(src_sketch, sd, 'input_sketch_proj', args.n_input_proj)
        vid = input_proj(src_video, sd, 'input_video_proj', args.n_input_proj)
        pos_s = (position_embedding_sine(src_sketch_mask, d, dtype) if args.use_sketch_pos else torch.zeros_like(skch))
        pos = torch.cat([pos_s, position_embedding_sine(src_video_mask, d, dtype)], dim=1)
        src = torch.cat([skch, vid], dim=1)
        mask = torch.cat([src_sketch_mask, src_video_mask], dim=1).bool()
    elif args.mode == 'concat_to_qry':
        src = input_proj(src_video, sd, 'input_video_proj', args.n_input_proj)
        mask = src_video_mask.bool()
        pos = position_embedding_sine(mask, d, dtype)
        query = _sketch_queries(sd, args, src_sketch, src_video.shape[0])
    else:
        raise NotImplementedError
    hs, memory, att = transformer_forward(sd, 'transformer.', args, src, ~mask, query, pos, drop=drop)
    res = _heads(sd, args, hs)
    if return_hs:
        return res, hs, memory, att
    return res


def sketch_detr_forward(sd, args, src_sketch, src_sketch_mask, src_video, src_video_mask):
    """SketchDETR.forward, sketch_detr.py:47-75: one enc/dec pass per frame feature -> list of per-frame dicts."""
    d = args.hidden_dim
    bs, T, _ = src_video.shape
    outputs = []
    for i in range(T):
        src = input_proj(src_video[:, i, :].unsqueeze(1), sd, 'input_video_proj', args.n_input_proj)
        mask = src_video_mask[:, i].unsqueeze(1).bool()
        pos = position_embedding_sine(mask, d, src.dtype)
        query = _sketch_queries(sd, args, src_sketch, bs)
        hs, _, _ = transformer_forward(sd, 'transformer.', args, src, ~mask, query, pos)
        outputs.append(_heads(sd, args, hs))
    return outputs
