"""Set criterion (reference lib/modeling/loss.py:10-37,126-157,192-213) on the device kernels.

``SetCriterion.forward(outputs, targets)`` returns the same dict of 0-d tensors
(``loss_label``, ``class_error``, ``loss_bbox``, ``loss_giou`` and their
``_{i}`` aux copies) and exposes the same ``weight_dict``.  Matching of ALL
decoder layers (the reference re-matches every aux layer, loss.py:148-155),
the losses and their gradients take three kernel launches and no host sync.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .matcher import build_matcher


class SetCriterion(nn.Module):
    def __init__(self, matcher, weight_dict, eos_coef, losses, bbox_type, sketch_head):
        super().__init__()
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.bbox_type = bbox_type
        self.sketch_head = sketch_head
        self.foreground_label = 0
        self.background_label = 1
        self.eos_coef = eos_coef
        empty_weight = torch.ones(2)
        empty_weight[-1] = self.eos_coef
        self.register_buffer('empty_weight', empty_weight)
        for l_ in losses:
            if l_ not in ('labels', 'boxes'):
                raise AssertionError(f'do you really want to compute {l_} loss?')
        self.last_match = None
        self.last_packed = None
        self.last_losses = None
        self._loss_table = None
        self._prepacked = None
        # optional StaticPackedTargets (svol_amd.graph): when set, `targets` passed to forward() are ignored and
        # the pre-loaded static buffers are used (hipGraph capture / replay)
        self.static_packed = None

    def forward(self, outputs, targets):
        if self.sketch_head == 'sketch_detr':  # loss.py:133-134,159-190: the same criterion on every per-frame output dict
            packed = None
            res = []
            for o in outputs:
                ld, packed = self._forward_one(o, targets, packed)
                res.append(ld)
            return res
        return self._forward_one(outputs, targets, None)[0]

    def _forward_one(self, outputs, targets, packed):
        if '_svol_stacked' in outputs:
            logits_all, boxes_all = outputs['_svol_stacked']
        else:  # a hand-made outputs dict: aux layers first, last layer last (svanet.py:128-137 order)
            aux = outputs.get('aux_outputs', [])
            logits_all = torch.stack([a['pred_logits'] for a in aux] + [outputs['pred_logits']])
            boxes_all = torch.stack([a['pred_boxes'] for a in aux] + [outputs['pred_boxes']])
        if not logits_all.is_cuda:
            raise RuntimeError('svol_amd.SetCriterion runs on the MI355X only (no CPU fallback)')
        NL, B, N = logits_all.shape[:3]
        m = self.matcher
        if packed is None:
            pre = self._prepacked
            if self.static_packed is not None:
                packed = self.static_packed
            elif pre is not None and pre[0] is targets and pre[1] == (NL, B, N):
                packed, self._prepacked = pre[2], None
            else:
                packed = m.pack(targets, NL, B, N, logits_all.device)
        losses, match = ops.SetCriterionFn.apply(logits_all, boxes_all, packed, m.cost_bbox, m.cost_giou,
                                                 m.cost_class, self.eos_coef)
        # logging copy WITHOUT the autograd graph: keeping the graph-carrying table until the next call would pin the
        # step's saved activations and AccumulateGrad nodes (and their creation streams) across iterations.  The one
        # graph-carrying reference is consumed (and dropped) by weighted_total().
        self.last_match, self.last_packed, self.last_losses = match, packed, losses.detach()
        self._loss_table = losses
        out = {}
        names = []
        if 'labels' in self.losses:
            names += [(0, 'loss_label'), (3, 'class_error')]
        if 'boxes' in self.losses:
            names += [(1, 'loss_bbox'), (2, 'loss_giou')]
        # one unbind (its backward is ONE stack) instead of a select per dictionary entry (each select's backward is a
        # zero-fill + copy + add on the [NL,4] loss table)
        flat = losses.reshape(-1).unbind(0)
        for col, name in names:
            out[name] = flat[(NL - 1) * 4 + col]
        for i in range(NL - 1):
            for col, name in names:
                out[f'{name}_{i}'] = flat[i * 4 + col]
        return out, packed

    def prepack(self, targets, n_layers, B, N, device):
        """Flatten `targets` (the Python list of dicts) and stage them on the device NOW — call it before the model's forward so
        that the host-side packing and the host->device copies run under the forward's kernels instead of between the last
        forward kernel and the cost kernel.  The next ``forward(outputs, targets)`` with the same list object uses the result."""
        self._prepacked = (targets, (n_layers, B, N), self.matcher.pack(targets, n_layers, B, N, device))

    def weighted_total(self):
        """sum(loss_dict[k] * weight_dict[k] for k in loss_dict if k in weight_dict) (train.py:227-228) of the LAST forward,
        as one multiply + one reduction over the [n_layers, 4] loss table instead of a Python sum over 3 * n_layers
        0-d tensors (18 multiplies + 18 adds, and as many backward kernels, at 6 layers)."""
        losses, self._loss_table = self._loss_table, None
        if losses is None:
            raise RuntimeError('weighted_total(): call forward() first (the loss table of a forward is consumed once)')
        NL = losses.shape[0]
        w = getattr(self, '_wtab', None)
        if w is None or w.shape[0] != NL or w.device != losses.device:
            cols = {'loss_label': 0, 'loss_bbox': 1, 'loss_giou': 2}
            t = torch.zeros((NL, 4), dtype=torch.float32)
            for name, col in cols.items():
                t[NL - 1, col] = float(self.weight_dict.get(name, 0.0))
                for i in range(NL - 1):
                    t[i, col] = float(self.weight_dict.get(f'{name}_{i}', 0.0))
            w = self._wtab = t.to(losses.device)
        if ops.FUSED_HEADS:   # one tiny launch each way instead of a multiply + a reduction (and their three backward kernels)
            return ops.WeightedTotalFn.apply(losses, w)
        return (losses * w).sum()

    def last_indices(self):
        """Reference-format matcher indices of the last forward, per layer (synchronises)."""
        p = self.last_packed
        p.check_status()
        return [p.indices_from_match(self.last_match, l_) for l_ in range(p.n_layers)]


def build_loss(args):
    matcher = build_matcher(args)
    weight_dict = {'loss_bbox': args.set_cost_bbox, 'loss_giou': args.set_cost_giou,
                   'loss_label': args.set_cost_class}
    if args.aux_loss:
        aux = {}
        for i in range(args.num_layers - 1):
            aux.update({k + f'_{i}': v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    return SetCriterion(matcher=matcher, weight_dict=weight_dict, eos_coef=args.eos_coef, losses=['labels', 'boxes'],
                        bbox_type=args.bbox_type, sketch_head=args.sketch_head)
