"""SVANet head (reference lib/modeling/svanet.py:17-63,65-141,184-200) on the MI355X kernels.

Same constructor signature, same child-module names / state-dict keys, same
output dict; children are parameter containers, arithmetic is in ``svol_amd.ops``.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from .. import ops
from . import cross_modal_transformer as cmt
from .cross_modal_transformer import build_cross_modal_transformer
from .position_encoding import build_position_encoding

# the mask / sketch chain of the forward's head on the query stream beside the video projection (SVOL_NO_INPUT_OVERLAP=1: one chain)
INPUT_OVERLAP = os.environ.get('SVOL_NO_INPUT_OVERLAP') is None

_DTYPES = {'bf16': torch.bfloat16, 'fp32': torch.float32, 'fp16': torch.float16, torch.float16: torch.float16, torch.bfloat16: torch.bfloat16,
           torch.float32: torch.float32}


class MLP(nn.Module):
    """bbox FFN: Linear-ReLU-...-Linear (svanet.py:144-156)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x, last_act=ops.ACT_NONE):
        for i, layer in enumerate(self.layers):
            act = ops.ACT_RELU if i < self.num_layers - 1 else last_act
            x = ops.linear(x, layer.weight, layer.bias, act)
        return x


class LinearLayer(nn.Module):
    """LN -> Dropout -> Linear (-> ReLU)  (svanet.py:159-181).  ``net.0`` is the dropout slot."""

    def __init__(self, in_hsz, out_hsz, layer_norm=True, dropout=0.1, relu=True):
        super().__init__()
        self.relu = relu
        self.layer_norm = layer_norm
        self.p = dropout
        if layer_norm:
            self.LayerNorm = nn.LayerNorm(in_hsz)
        self.net = nn.Sequential(nn.Dropout(dropout), nn.Linear(in_hsz, out_hsz))

    def forward(self, x, seed=0, out_f32=False, seed_dev=None):
        p = self.p if self.training else 0.0
        if self.layer_norm:
            x = ops.layer_norm(x, self.LayerNorm.weight, self.LayerNorm.bias, p, seed, seed_dev)
        elif p > 0.0:
            raise NotImplementedError('dropout without LayerNorm is not on the reference path')
        lin = self.net[1]
        return ops.linear(x, lin.weight, lin.bias, ops.ACT_RELU if self.relu else ops.ACT_NONE, out_f32)


class SVANet(nn.Module):
    def __init__(self, transformer, sketch_position_embed, video_position_embed, input_vid_dim, input_skch_dim,
                 num_queries, input_dropout=0.1, aux_loss=True, use_sketch_pos=True, n_input_proj=2, num_classes=2,
                 vis_mode=None, compute_dtype='bf16'):
        super().__init__()
        self.num_queries = num_queries
        self.num_classes = num_classes
        self.transformer = transformer
        self.sketch_position_embed = sketch_position_embed
        self.video_position_embed = video_position_embed
        hidden_dim = transformer.d_model
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.use_sketch_pos = use_sketch_pos
        self.class_embed = nn.Linear(hidden_dim, 2)
        self.n_input_proj = n_input_proj
        self.class_head = nn.Linear(hidden_dim, num_classes)  # unused by the reference forward too
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        relu_args = [True] * 3
        relu_args[n_input_proj - 1] = False
        # the reference builds three layers per projection and keeps the first n (RNG-stream parity)
        self.input_video_proj = nn.Sequential(*[
            LinearLayer(input_vid_dim, hidden_dim, dropout=input_dropout, relu=relu_args[0]),
            LinearLayer(hidden_dim, hidden_dim, dropout=input_dropout, relu=relu_args[1]),
            LinearLayer(hidden_dim, hidden_dim, dropout=input_dropout, relu=relu_args[2])][:n_input_proj])
        self.input_sketch_proj = nn.Sequential(*[
            LinearLayer(input_skch_dim, hidden_dim, dropout=input_dropout, relu=relu_args[0]),
            LinearLayer(hidden_dim, hidden_dim, dropout=input_dropout, relu=relu_args[1]),
            LinearLayer(hidden_dim, hidden_dim, dropout=input_dropout, relu=relu_args[2])][:n_input_proj])
        self.vis_mode = vis_mode
        self.aux_loss = aux_loss
        self.compute_dtype = _DTYPES[compute_dtype]
        self._step = 0
        self.base_seed = 1
        # device-side step counter added to the dropout seed (svol_amd.graph advances it inside a captured
        # hipGraph so that every replay draws a fresh mask); None = host-side counter only
        self.step_dev = None

    def _proj(self, seq, x, salt):
        """input projection; the last layer emits fp32 (start of the fp32 residual stream)."""
        n = len(seq)
        for j, layer in enumerate(seq):
            x = layer(x, seed=(self.base_seed << 32) + (self._step << 8) + salt * 16 + j, out_f32=(j == n - 1),
                      seed_dev=self.step_dev)
        return x

    def forward(self, src_sketch, src_sketch_mask, src_video, src_video_mask):
        """Inputs/outputs as the reference (svanet.py:65-141); masks contain 1 on valid tokens."""
        if not src_video.is_cuda:
            raise RuntimeError('svol_amd.SVANet runs on the MI355X HIP kernels only; move the module and its '
                               'inputs to cuda (there is no CPU path — the CPU oracle lives under oracle/).')
        dt = self.compute_dtype
        d = self.transformer.d_model
        overlap = INPUT_OVERLAP and cmt.OVERLAP_QUERY_STREAM and isinstance(self.transformer, cmt.CrossModalTransformer)
        if not overlap:
            ops.weights.new_epoch()  # re-cast every fp32 master weight once per forward (see ops._WeightCache)
        if self.training:
            self._step += 1
        def mask_side(with_us):
            """what depends on the masks and the sketch only: position encoding, key bias, sketch projection, every layer's gate
            vectors (B*d-sized algebra; their backward is 2 launches per step that used to close the backward on the main stream)"""
            mask_f = src_video_mask.to(torch.float32)
            pos_video = self.video_position_embed(mask_f, d, dt)
            skch = self._proj(self.input_sketch_proj, ops.cast_ag(src_sketch.float(), dt), 1)
            skch = skch.reshape(skch.shape[0], -1)
            # key_padding_mask (True on pads) as an additive bias for the cross-attention kernel
            kbias = torch.zeros_like(mask_f).masked_fill_(mask_f == 0, float('-inf'))
            us = cmt.all_gate_vectors(list(self.transformer.layers), skch) if with_us else None
            return pos_video, skch, kbias, us

        if overlap:
            # Round 6: the step's head (and, replayed by autograd on the same streams, its tail) was one chain of ~25 dependent small
            # launches on the main stream.  The mask / sketch chain is independent of the video projection: it runs on the query
            # stream beside it (forward ~70 us, backward ~150 us of the 1.0 ms step boundary) — and so do the weight casts, beside
            # the cast of the video features (the first kernel that needs a cast weight is the first projection GEMM).
            main = torch.cuda.current_stream()
            side = cmt._side_stream(main.device)
            side.wait_stream(main)   # the optimizer's update, the last readers of the old copies, the caller's inputs
            with torch.cuda.stream(side):
                ops.weights.new_epoch()
                casts_done = side.record_event()
            xv = ops.cast_ag(src_video.float(), dt)
            with torch.cuda.stream(side):
                pos_video, skch, kbias, us = mask_side(True)
            main.wait_event(casts_done)
            vid = self._proj(self.input_video_proj, xv, 0)
            main.wait_stream(side)
            for t in [pos_video, skch, kbias] + list(us or []):
                t.record_stream(main)   # allocated on the query stream, read by the video half on this one
        else:
            vid = self._proj(self.input_video_proj, ops.cast_ag(src_video.float(), dt), 0)
            pos_video, skch, kbias, us = mask_side(False)
        hs = (self.transformer(vid, skch, kbias, pos_video, self.query_embed.weight, us) if us is not None else
              self.transformer(vid, skch, kbias, pos_video, self.query_embed.weight))   # [NL,B,N,d] fp32
        # heads run in fp32 (tiny; keeps logits / box coordinates at full precision)
        if ops.heads_fusable(hs, self.class_embed, self.bbox_embed):   # both heads, all layers, one launch (csrc/heads.hip)
            outputs_class, outputs_coord = ops.heads(hs, self.class_embed, self.bbox_embed)
        else:
            outputs_class = ops.linear(hs, self.class_embed.weight, self.class_embed.bias)
            outputs_coord = self.bbox_embed(hs, last_act=ops.ACT_SIGMOID)
        out = {'pred_logits': outputs_class[-1], 'pred_boxes': outputs_coord[-1]}
        if self.aux_loss:
            out['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b}
                                  for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
        # all layers stacked, so the criterion can match + score every layer in one launch
        out['_svol_stacked'] = (outputs_class, outputs_coord)
        if self.vis_mode is not None:
            return out, hs
        return out


def build_svanet(args):
    transformer = build_cross_modal_transformer(args)
    sketch_position_embed, video_position_embed = build_position_encoding(args)
    return SVANet(transformer, sketch_position_embed, video_position_embed, input_vid_dim=args.input_vid_dim,
                  input_skch_dim=args.input_skch_dim, num_queries=args.num_queries,
                  input_dropout=args.input_dropout, aux_loss=args.aux_loss, use_sketch_pos=args.use_sketch_pos,
                  n_input_proj=args.n_input_proj, vis_mode=args.vis_mode,
                  compute_dtype=getattr(args, 'compute_dtype', 'bf16'))
