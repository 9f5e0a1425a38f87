"""SVANet fusion variants on the enc/dec ``Transformer`` (reference lib/modeling/svanet_variants.py:13-306) —
SURVEY.md §8 f2.  Three ways of presenting the sketch to a DETR-style encoder/decoder:

  ``concat_to_seq``  the sketch feature is concatenated to EVERY video token's feature ([B,L,2D] -> input_proj)   (:86-134)
  ``append_to_seq``  projected sketch tokens are prepended to the projected video tokens ([B,Ls+L,d])               (:136-188)
  ``concat_to_qry``  the sketch feature is concatenated to every object query ([N,B,d+D] -> input_query_proj)       (:190-247)

Same constructor signature, child names / state-dict keys, construction order and output dict as the reference;
children are parameter containers, the arithmetic is ``svol_amd.ops``.  The encoder memory / attention-weight
slices the reference computes and drops (:111-116) are not produced (``need_weights=False``).
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .position_encoding import build_position_encoding
from .svanet import MLP, LinearLayer, _DTYPES
from .transformer import build_transformer


def _proj_stack(input_dim, hidden_dim, input_dropout, n_input_proj):
    relu_args = [True] * 3
    relu_args[n_input_proj - 1] = False
    return nn.Sequential(*[
        LinearLayer(input_dim, hidden_dim, dropout=input_dropout, relu=relu_args[0]),
        LinearLayer(hidden_dim, hidden_dim, dropout=input_dropout, relu=relu_args[1]),
        LinearLayer(hidden_dim, hidden_dim, dropout=input_dropout, relu=relu_args[2])][:n_input_proj])


class _DetrHead(nn.Module):
    """what the variants and SketchDETR share: projections through LinearLayer stacks, box / class heads on ``hs``."""

    def _proj(self, seq, x, salt):
        n = len(seq)
        for j, layer in enumerate(seq):
            x = layer(x, seed=(self.base_seed << 32) + (self._step << 8) + salt * 16 + j, out_f32=(j == n - 1))
        return x

    def _begin(self, src_video):
        if not src_video.is_cuda:
            raise RuntimeError(f'svol_amd.{type(self).__name__} runs on the MI355X HIP kernels only; move the module and '
                               'its inputs to cuda (there is no CPU path — the CPU oracle lives under oracle/).')
        ops.weights.new_epoch()
        if self.training:
            self._step += 1

    def _heads(self, hs):
        outputs_class = ops.linear(hs, self.class_embed.weight, self.class_embed.bias)
        outputs_coord = self.bbox_embed(hs, last_act=ops.ACT_SIGMOID)
        out = {'pred_logits': outputs_class[-1], 'pred_boxes': outputs_coord[-1]}
        if self.aux_loss:
            out['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b}
                                  for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
        out['_svol_stacked'] = (outputs_class, outputs_coord)
        return out

    def _sketch_queries(self, src_sketch, bs, dt):
        """input_query_proj(cat[query_embed, sketch]) -> [N,B,d] fp32 (svanet_variants.py:215-219, sketch_detr.py:57-60)."""
        nq = self.query_embed.weight.shape[0]
        sk = src_sketch.float().repeat(1, nq, 1).permute(1, 0, 2)                 # [N,B,D]
        qw = self.query_embed.weight.unsqueeze(1).repeat(1, bs, 1)                # [N,B,d]
        query = torch.cat([qw, sk], dim=2).contiguous()                           # [N,B,d+D]
        return self._proj(self.input_query_proj, ops.cast_ag(query, dt), 3)


class SVANet(_DetrHead):
    def __init__(self, transformer, sketch_position_embed, video_position_embed, mode, input_dim, num_queries,
                 input_dropout=0.1, aux_loss=True, use_sketch_pos=True, n_input_proj=2, num_classes=2, vis_mode=None,
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
        self.class_head = nn.Linear(hidden_dim, num_classes)  # unused by the reference forward too
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.input_sketch_proj = _proj_stack(input_dim, hidden_dim, input_dropout, n_input_proj)
        self.input_video_proj = _proj_stack(input_dim, hidden_dim, input_dropout, n_input_proj)
        self.input_proj = _proj_stack(input_dim * 2, hidden_dim, input_dropout, n_input_proj)
        self.input_query_proj = _proj_stack(input_dim + hidden_dim, hidden_dim, input_dropout, n_input_proj)
        self.vis_mode = vis_mode
        self.aux_loss = aux_loss
        self.compute_dtype = _DTYPES[compute_dtype]
        self._step = 0
        self.base_seed = 1

    def forward(self, src_sketch, src_sketch_mask, src_video, src_video_mask):
        """src_sketch [B,Ls,D], src_video [B,L,D], masks 1 on valid tokens (svanet_variants.py:74-84)."""
        if self.mode not in ('concat_to_seq', 'append_to_seq', 'concat_to_qry'):
            raise NotImplementedError
        self._begin(src_video)
        dt = self.compute_dtype
        d = self.transformer.d_model
        vmask = src_video_mask.to(torch.float32)
        query = self.query_embed.weight
        if self.mode == 'concat_to_seq':
            sk = src_sketch.float().repeat(1, src_video.shape[1], 1)                      # needs Ls == 1, as the reference
            src = self._proj(self.input_proj, ops.cast_ag(torch.cat([sk, src_video.float()], dim=2).contiguous(), dt), 2)
            mask_f = vmask
            pos = self.video_position_embed(vmask, d, dt)
        elif self.mode == 'append_to_seq':
            skch = self._proj(self.input_sketch_proj, ops.cast_ag(src_sketch.float(), dt), 1)
            vid = self._proj(self.input_video_proj, ops.cast_ag(src_video.float(), dt), 0)
            smask = src_sketch_mask.to(torch.float32)
            pos_s = (self.sketch_position_embed(smask, d, dt) if self.use_sketch_pos
                     else torch.zeros(skch.shape, dtype=dt, device=skch.device))
            pos = torch.cat([pos_s, self.video_position_embed(vmask, d, dt)], dim=1)
            src = torch.cat([skch, vid], dim=1)
            mask_f = torch.cat([smask, vmask], dim=1)
        else:  # concat_to_qry
            src = self._proj(self.input_video_proj, ops.cast_ag(src_video.float(), dt), 0)
            mask_f = vmask
            pos = self.video_position_embed(vmask, d, dt)
            query = self._sketch_queries(src_sketch, src_video.shape[0], dt)
        hs, _, _ = self.transformer(src, mask_f == 0, query, pos, need_weights=False)
        out = self._heads(hs)
        if self.mode == 'concat_to_qry' and self.vis_mode is not None:  # only this branch returns hs (:244-247)
            return out, hs
        return out


def build_svanet(args):
    transformer = build_transformer(args)
    sketch_position_embed, video_position_embed = build_position_encoding(args)
    return SVANet(
        transformer, sketch_position_embed, video_position_embed, mode=args.mode,
        input_dim=args.feat_dim if args.backbone != 'resnet' else 512, num_queries=args.num_queries,
        input_dropout=args.input_dropout, aux_loss=args.aux_loss, use_sketch_pos=args.use_sketch_pos,
        n_input_proj=args.n_input_proj, vis_mode=args.vis_mode, compute_dtype=getattr(args, 'compute_dtype', 'bf16'))
