"""Composition of the evaluation records from the head outputs (reference test.py:133-169), scores / box
conversion / per-frame sort on the device (``svol_postprocess``); only the final formatting of the JSONL wire format
(``float(f'{e:.4f}')``, test.py:154) is host work.

    results = compose_results(outputs, targets, args.num_frames)      # list of dicts, the reference's schema
    save_jsonl(results, path)                                         # one json.dumps per line (lib/utils/misc.py:46-49)
"""
from __future__ import annotations

import json

import torch

from . import _lib
from .ops import _ptr, _stream


def postprocess(pred_logits: torch.Tensor, pred_boxes: torch.Tensor, num_frames: int) -> torch.Tensor:
    """[B,N,2], [B,N,4] (cxcywh) -> [B,N,5] fp32 (x0,y0,x1,y1 in [0,1], foreground score), rows sorted by score
    (descending, stable) inside each ``torch.chunk(num_frames)`` block of queries."""
    if not pred_logits.is_cuda:
        raise RuntimeError('svol_amd.postprocess runs on the MI355X only (no CPU fallback)')
    lg = pred_logits.detach().float().contiguous()
    bx = pred_boxes.detach().float().contiguous()
    B, N = lg.shape[:2]
    chunk = -(-N // int(num_frames))  # torch.chunk: ceil(N / chunks) rows per chunk
    out = torch.empty((B, N, 5), dtype=torch.float32, device=lg.device)
    _lib.check(_lib.lib().svol_postprocess(_ptr(lg), _ptr(bx), _ptr(out), B, N, chunk, _stream()), 'svol_postprocess')
    return out


def _xyxy(b):
    """box_cxcywh_to_xyxy (lib/utils/box_utils.py:9-13) of one ground-truth box, in the tensor's own dtype (fp32)."""
    t = torch.as_tensor(b)
    cx, cy, w, h = t.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1).tolist()


def compose_results(outputs: dict, targets: list, num_frames: int, no_sort_results: bool = False) -> list:
    """One record per (video, annotated frame), keys and number formatting of test.py:158-166."""
    if no_sort_results:
        # the reference leaves `sorted_preds` unbound on this path (test.py:151-154) and fails; so does this
        raise UnboundLocalError("local variable 'sorted_preds' referenced before assignment")
    pp = postprocess(outputs['pred_logits'], outputs['pred_boxes'], num_frames).cpu()
    N = pp.shape[1]
    chunk = -(-N // int(num_frames))
    rows = pp.tolist()
    results = []
    for target, vid_rows in zip(targets, rows):
        frame_idxs = list(target['bboxes'].keys())
        n_chunks = -(-N // chunk)
        for c, fidx in zip(range(n_chunks), frame_idxs):
            preds = [[float(f'{e:.4f}') for e in row] for row in vid_rows[c * chunk:(c + 1) * chunk]]
            gt_boxes = [{'track_id': ib['track_id'], 'bbox': _xyxy(ib['bbox'])} for ib in target['bboxes'][fidx]]
            results.append(dict(video=target['video'], sketch=target['sketch'], shape=target['size'], frame=fidx,
                                gt_boxes=gt_boxes, pred_boxes=preds))
    return results


def save_jsonl(data, filename):
    """lib/utils/misc.py:46-49."""
    with open(filename, 'w') as f:
        f.write('\n'.join([json.dumps(e) for e in data]))
