"""Encoder/decoder ``Transformer`` (DETR style) on the MI355X kernels — SURVEY.md §8 f2.

Mirrors the reference's module tree (lib/modeling/transformer.py:18-333): same constructor signatures, same
child names and state-dict keys (``encoder.layers.{i}.self_attn.in_proj_weight`` … ``decoder.norm.bias``), the
same construction order (so a seed gives the same initial weights) and the same ``forward(src, mask,
query_embed, pos_embed) -> (hs, memory, att_weights)`` contract.  The ``nn.MultiheadAttention`` / ``nn.Linear`` /
``nn.LayerNorm`` children are parameter containers only; the arithmetic goes through ``svol_amd.ops``, i.e. the
same fused blocks as the cross-modal transformer:

  encoder layer, post-norm (:183-194)   AttnLNFn (q = k = x + pos, v = x, key_padding_mask, out_proj + residual, norm1)
                                        MLPLNFn  (linear1 + ReLU, linear2 + residual, norm2)
  decoder layer, post-norm (:237-250)   AttnLNFn (query self-attention, norm1), AttnLNFn (query -> memory, norm2,
                                        optionally the head-averaged weights), MLPLNFn (norm3)
  pre-norm variants (:196-208, :265-283) LNStreamFn opens each block, AttnLNFn / MLPLNFn run without their norm and
                                        hand the fp32 sum on; encoder.norm closes the encoder
  decoder.norm on every layer's output  LNStreamFn (fp32), shared parameters (:139-147)

The residual stream is fp32 throughout (as in the cross-modal transformer); GEMM / attention operands are in the
compute dtype.  Differences from the reference, all documented in DESIGN.md:

* dropout (training mode, reference default 0.1): attention-probability dropout inside the attention kernels
  (``svol_attn_fwd_dropout``), the residual dropouts (``svol_dropout_add``) and the FFN hidden dropout (``svol_dropout``), all
  stateless masks of (seed, element index) that backward regenerates; the masks of a step are a function of
  ``(drop_base_seed, step, layer, site)``.  The returned attention WEIGHTS are those of the undropped softmax (torch
  returns the dropped ones; no consumer in the reference reads them).
* activation: ``relu`` (the only one ``build_transformer`` can select, :312-322) and ``gelu``; ``glu`` raises.
* ``att_weights`` carry no gradient (every consumer in the reference discards them: sketch_detr.py:62,
  svanet_variants.py:107-116); ``need_weights=False`` skips computing them.
* attn_mask / tgt_mask / memory_mask / tgt_key_padding_mask are never passed by the reference's callers and are
  not supported (key padding masks are).
"""
from __future__ import annotations

import copy
from typing import Optional

import torch
from torch import nn

from .. import ops

_DTYPES = {'bf16': torch.bfloat16, 'fp32': torch.float32, 'fp16': torch.float16, torch.float16: torch.float16, torch.bfloat16: torch.bfloat16,
           torch.float32: torch.float32}


def _act_code(activation: str) -> int:
    """transformer.py:325-333"""
    if activation == 'relu':
        return ops.ACT_RELU
    if activation == 'gelu':
        return ops.ACT_GELU
    if activation == 'glu':
        raise NotImplementedError('glu halves the FFN width; not available on the HIP path')
    raise RuntimeError(f'activation should be relu/gelu, not {activation}.')


_n = lambda m: (m.weight, m.bias)
_mha = lambda m: (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias)


def _drop(mod, seed, site_a, site_b):
    """(p, seed_a, seed_b) of one block of layer `mod` in training mode, else None.  seed: the step's base seed (Transformer.forward),
    None outside a Transformer (then a per-layer counter stands in).  Sites 0..5 of a layer: self-attention probabilities / its
    residual dropout1, cross-attention probabilities / dropout2, FFN hidden dropout / the FFN's residual dropout."""
    if not (mod.training and mod.dropout_p > 0.0):
        return None
    if torch.cuda.is_current_stream_capturing():
        # the seeds are host integers passed BY VALUE into the launches: a captured graph would replay one step's masks forever
        # (ADVICE r3).  The SVANet input projections take a device-side seed offset (svol_layernorm_fwd's seed_offset_dev) and can be
        # captured; the enc/dec Transformer's attention / residual / FFN dropouts cannot.
        raise RuntimeError('training-mode dropout of the enc/dec Transformer cannot be captured in a hipGraph (its mask seeds are host '
                           'values): run the step eagerly, or set --dropout 0')
    if seed is None:
        mod._own_step = getattr(mod, '_own_step', 0) + 1
        seed = (0x5EED << 40) + (mod._own_step << 20)
    base = seed + (getattr(mod, 'layer_id', 0) << 4)
    return (mod.dropout_p, base + site_a, base + site_b)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.act = _act_code(activation)
        self.normalize_before = normalize_before
        self.nhead, self.dropout_p = nhead, float(dropout)

    def _ffn(self, x32, x, norm, pos_out, seed=None):
        g, b = _n(norm) if norm is not None else (None, None)
        return ops.mlp_ln(x32, x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, g, b,
                          pos_out, self.act, _drop(self, seed, 4, 5))

    def forward_post(self, state, pos, kbias, seed=None):
        """state = (x32, x, x + pos) -> same triple (transformer.py:175-194)."""
        x32, x, xpos = state
        y32, y = ops.self_attn_ln(x32, x, xpos, *_mha(self.self_attn), *_n(self.norm1), None, self.nhead, kbias,
                                  _drop(self, seed, 0, 1))
        return self._ffn(y32, y, self.norm2, pos, seed)

    def forward_pre(self, x32, pos, kbias, seed=None):
        """x32 -> x32 (transformer.py:196-208)."""
        dt = pos.dtype
        y, ypos = ops.ln_stream(x32, *_n(self.norm1), pos, dt)
        s32 = ops.self_attn_ln(x32, y, ypos, *_mha(self.self_attn), None, None, None, self.nhead, kbias, _drop(self, seed, 0, 1))
        y2 = ops.ln_stream(s32, *_n(self.norm2), None, dt)
        return self._ffn(s32, y2, None, None, seed)

    def forward(self, state, pos, kbias, seed=None):
        if self.normalize_before:
            return self.forward_pre(state, pos, kbias, seed)
        return self.forward_post(state, pos, kbias, seed)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.act = _act_code(activation)
        self.normalize_before = normalize_before
        self.nhead, self.dropout_p = nhead, float(dropout)

    def _ffn(self, x32, x, norm, pos_out, seed=None):
        g, b = _n(norm) if norm is not None else (None, None)
        return ops.mlp_ln(x32, x, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, g, b,
                          pos_out, self.act, _drop(self, seed, 4, 5))

    def forward_post(self, state, mem, mempos, qpos, kbias, need_weights, seed=None):
        """state = (t32, t, t + query_pos) -> (same triple, att | None)   (transformer.py:229-250)."""
        t32, t, tpos = state
        a32, a, apos = ops.self_attn_ln(t32, t, tpos, *_mha(self.self_attn), *_n(self.norm1), qpos, self.nhead, None,
                                        _drop(self, seed, 0, 1))
        r = ops.cross_attn_ln(a32, a, apos, mempos, mem, *_mha(self.multihead_attn), *_n(self.norm2), None, self.nhead,
                              kbias, need_weights, _drop(self, seed, 2, 3))
        c32, c = r[0], r[1]
        att = r[2] if need_weights else None
        return self._ffn(c32, c, self.norm3, qpos, seed), att

    def forward_pre(self, t32, mem, mempos, qpos, kbias, need_weights, seed=None):
        """t32 -> (t32, att | None)   (transformer.py:252-283)."""
        dt = mem.dtype
        y, ypos = ops.ln_stream(t32, *_n(self.norm1), qpos, dt)
        s32 = ops.self_attn_ln(t32, y, ypos, *_mha(self.self_attn), None, None, None, self.nhead, None, _drop(self, seed, 0, 1))
        y2, y2pos = ops.ln_stream(s32, *_n(self.norm2), qpos, dt)
        r = ops.cross_attn_ln(s32, y2, y2pos, mempos, mem, *_mha(self.multihead_attn), None, None, None, self.nhead,
                              kbias, need_weights, _drop(self, seed, 2, 3))
        c32, att = (r[0], r[1]) if need_weights else (r, None)
        y3 = ops.ln_stream(c32, *_n(self.norm3), None, dt)
        return self._ffn(c32, y3, None, None, seed), att

    def forward(self, state, mem, mempos, qpos, kbias, need_weights=False, seed=None):
        if self.normalize_before:
            return self.forward_pre(state, mem, mempos, qpos, kbias, need_weights, seed)
        return self.forward_post(state, mem, mempos, qpos, kbias, need_weights, seed)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        for i, l_ in enumerate(self.layers):
            l_.layer_id = i            # (dropout seeds: every layer draws its own masks)
        self.num_layers = num_layers
        self.norm = norm
        if return_intermediate:  # never requested by the reference's builder (transformer.py:27-29)
            raise NotImplementedError('encoder return_intermediate')

    def forward(self, src32, pos, kbias, seed=None):
        """src32 [B,L,d] fp32, pos [B,L,d] compute dtype, kbias [B,L] fp32 additive key mask or None ->
        (memory32 [B,L,d] fp32, memory, memory + pos) — what the decoder's cross-attention reads."""
        dt = pos.dtype
        pre = len(self.layers) > 0 and self.layers[0].normalize_before
        if pre:
            x32 = src32
            for layer in self.layers:
                x32 = layer(x32, pos, kbias, seed)
            if self.norm is not None:  # the builder gives the pre-norm encoder a closing norm (:26)
                return ops.ln_stream(x32, *_n(self.norm), pos, dt, True)
            x = ops.cast_ag(x32, dt)
            return x32, x, x + pos
        x = ops.cast_ag(src32, dt)
        state = (src32, x, x + pos)  # layer 0's q = k = src + pos (:183); later layers get it from the norm2 epilogue
        for layer in self.layers:
            state = layer(state, pos, kbias, seed)
        if self.norm is not None:
            return ops.ln_stream(state[0], *_n(self.norm), pos, dt, True)
        return state


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        for i, l_ in enumerate(self.layers):
            l_.layer_id = 64 + i
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, mem, mempos, qpos, kbias, need_weights=True, seed=None):
        """mem / mempos [B,L,d] compute dtype, qpos [N,d] or [B,N,d] compute dtype ->
        (hs [n,B,N,d] fp32, att [n,B,N,L] fp32 | None); n = num_layers with return_intermediate, else 1 (:116-152)."""
        B = mem.shape[0]
        dt = mem.dtype
        N, d = qpos.shape[-2:]
        qfull = qpos if qpos.dim() == 3 else qpos.unsqueeze(0).expand(B, -1, -1).contiguous()
        t32 = torch.zeros((B, N, d), dtype=torch.float32, device=mem.device)  # tgt = zeros_like(query_embed), :64
        pre = len(self.layers) > 0 and self.layers[0].normalize_before
        state = t32 if pre else (t32, torch.zeros((B, N, d), dtype=dt, device=mem.device), qfull)
        inter, atts = [], []
        for layer in self.layers:
            state, att = layer(state, mem, mempos, qpos, kbias, need_weights and self.return_intermediate, seed)
            if self.return_intermediate:
                o32 = state if pre else state[0]
                inter.append(ops.ln_stream(o32, *_n(self.norm), None, torch.float32) if self.norm is not None else o32)
                atts.append(att)
        if self.return_intermediate:
            return torch.stack(inter), (torch.stack(atts) if need_weights else None)
        o32 = state if pre else state[0]
        if self.norm is not None:
            o32 = ops.ln_stream(o32, *_n(self.norm), None, torch.float32)
        return o32.unsqueeze(0), None


class Transformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation='relu', normalize_before=False, return_intermediate_dec=False,
                 compute_dtype='bf16'):
        super().__init__()
        encoder_layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        encoder_norm = nn.LayerNorm(d_model) if normalize_before else None
        self.encoder = TransformerEncoder(encoder_layer, num_encoder_layers, encoder_norm)
        decoder_layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        decoder_norm = nn.LayerNorm(d_model)
        self.decoder = TransformerDecoder(decoder_layer, num_decoder_layers, decoder_norm,
                                          return_intermediate=return_intermediate_dec)
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead
        self.compute_dtype = _DTYPES[compute_dtype]
        # dropout masks are a function of (drop_base_seed + rank, training step, layer, site, element): different on every rank of a
        # data-parallel job (each rank sees its own videos AND its own masks, as with torch's per-process generators), continued
        # after a resume through dropout_state() / load_dropout_state() (svol_amd.utils.checkpoint keeps them in the file)
        self.drop_base_seed = 1
        self._drop_step = 0

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def dropout_state(self):
        return {'drop_base_seed': int(self.drop_base_seed), 'drop_step': int(self._drop_step)}

    def load_dropout_state(self, st):
        self.drop_base_seed, self._drop_step = int(st['drop_base_seed']), int(st['drop_step'])

    def forward(self, src, mask, query_embed, pos_embed, need_weights: bool = True):
        """src [B,L,d] (fp32: it starts the fp32 residual stream), mask [B,L] bool with True on PADDED positions (or
        None), query_embed [N,d] or [N,B,d], pos_embed [B,L,d].  Returns (hs [n_dec,B,N,d] fp32, memory [B,L,d] fp32,
        att_weights [n_dec,B,N,L] fp32 or None) — transformer.py:43-81."""
        if not src.is_cuda:
            raise RuntimeError('svol_amd Transformer runs on the MI355X HIP kernels only (no CPU path; the CPU oracle '
                               'lives under oracle/)')
        dt = self.compute_dtype
        src32 = src if src.dtype == torch.float32 else src.float()
        pos = pos_embed if pos_embed.dtype == dt else ops.cast_ag(pos_embed.float().contiguous(), dt)
        kbias = None
        if mask is not None:
            kbias = torch.zeros(mask.shape, dtype=torch.float32, device=src.device).masked_fill_(mask.bool(), float('-inf'))
        if query_embed.dim() == 3:  # [N,B,d] per-sample queries (sketch_detr.py:57-60, svanet_variants.py:215-219)
            qpos = query_embed.transpose(0, 1)
        else:
            qpos = query_embed
        qpos = ops.cast_ag(qpos.float().contiguous(), dt)
        seed = None
        if self.training:
            self._drop_step += 1
            seed = ((int(self.drop_base_seed) + _rank()) << 44) + (self._drop_step << 12)
        mem32, mem, mempos = self.encoder(src32.contiguous(), pos.contiguous(), kbias, seed)
        hs, att = self.decoder(mem, mempos, qpos, kbias, need_weights, seed)
        return hs, mem32, att


def _rank():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank()
    import os
    return int(os.environ.get('RANK', 0))


def build_transformer(args):
    """transformer.py:312-322 (+ the build's compute dtype)."""
    return Transformer(
        d_model=args.hidden_dim,
        dropout=args.dropout,
        nhead=args.nheads,
        dim_feedforward=args.dim_feedforward,
        num_encoder_layers=args.enc_layers,
        num_decoder_layers=args.dec_layers,
        normalize_before=args.pre_norm,
        return_intermediate_dec=True,
        compute_dtype=getattr(args, 'compute_dtype', 'bf16'),
    )
