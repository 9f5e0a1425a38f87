"""SketchDETR baseline head (reference lib/modeling/sketch_detr.py:13-132) — SURVEY.md §8 f2.

Per-frame sketch-conditioned DETR: every frame feature is a ONE-token encoder memory, the object queries are
``input_query_proj(cat[query_embed, sketch])``, and the reference runs the enc/dec Transformer once per frame in a
Python loop (:50-73).  Frames never interact, so this build folds the frame axis into the batch — one Transformer
call over B*T one-token sequences — and splits the result back into the reference's list of T per-frame output
dicts.  (In eval mode this is the same arithmetic; in training the reference would also redraw the input dropout
of the loop-invariant query projection once per frame, here it is drawn once.)
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .position_encoding import build_position_encoding
from .svanet import MLP, _DTYPES
from .svanet_variants import _DetrHead, _proj_stack
from .transformer import build_transformer


class SketchDETR(_DetrHead):
    def __init__(self, transformer, sketch_position_embed, video_position_embed, mode, input_dim, num_queries,
                 input_dropout=0.1, aux_loss=True, use_sketch_pos=True, n_input_proj=2, num_classes=2,
                 compute_dtype='bf16'):
        super().__init__()
        self.mode = mode
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
        self.class_head = nn.Linear(hidden_dim, num_classes)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.input_video_proj = _proj_stack(input_dim, hidden_dim, input_dropout, n_input_proj)
        self.input_query_proj = _proj_stack(input_dim + hidden_dim, hidden_dim, input_dropout, n_input_proj)
        self.aux_loss = aux_loss
        self.compute_dtype = _DTYPES[compute_dtype]
        self._step = 0
        self.base_seed = 1

    def forward(self, src_sketch, src_sketch_mask, src_video, src_video_mask):
        """src_sketch [B,1,D], src_video [B,T,D] (one feature per frame), masks 1 on valid -> list of T output dicts."""
        self._begin(src_video)
        dt = self.compute_dtype
        d = self.transformer.d_model
        bs, T, _ = src_video.shape
        src = self._proj(self.input_video_proj, ops.cast_ag(src_video.float().reshape(bs * T, 1, -1), dt), 0)  # [B*T,1,d]
        mask_f = src_video_mask.to(torch.float32).reshape(bs * T, 1)
        pos = self.video_position_embed(mask_f, d, dt)                                                       # [B*T,1,d]
        query = self._sketch_queries(src_sketch, bs, dt)                                                     # [N,B,d]
        nq = query.shape[0]
        query = query.unsqueeze(2).expand(nq, bs, T, d).reshape(nq, bs * T, d)                               # frame-major inside b
        hs, _, _ = self.transformer(src, mask_f == 0, query, pos, need_weights=False)                        # [n,B*T,N,d]
        hs = hs.view(hs.shape[0], bs, T, nq, d)
        outputs = []
        stacked = self._heads(hs)  # heads once over all frames, then per-frame views
        cls_all, box_all = stacked['_svol_stacked']
        for i in range(T):
            c, b = cls_all[:, :, i], box_all[:, :, i]
            out = {'pred_logits': c[-1], 'pred_boxes': b[-1]}
            if self.aux_loss:
                out['aux_outputs'] = [{'pred_logits': a_, 'pred_boxes': b_} for a_, b_ in zip(c[:-1], b[:-1])]
            outputs.append(out)
        return outputs


def build_sketchdetr(args):
    transformer = build_transformer(args)
    sketch_position_embed, video_position_embed = build_position_encoding(args)
    return SketchDETR(
        transformer, sketch_position_embed, video_position_embed, mode=args.mode, input_dim=args.feat_dim,
        num_queries=100,  # hard-coded in the reference (sketch_detr.py:127)
        input_dropout=args.input_dropout, aux_loss=args.aux_loss, use_sketch_pos=args.use_sketch_pos,
        n_input_proj=args.n_input_proj, compute_dtype=getattr(args, 'compute_dtype', 'bf16'))
