"""``lib.evaluate.eval`` of the reference (eval.py:12-172) with the per-prediction work on the device.

Same functions, arguments and return values: ``eval_results(results, verbose, logger, match_number)``,
``eval_svol``, ``compute_ap``, ``compute_recall_at_k``.  ``results`` is the list of records of the JSONL wire format
(test.py:158-166).  The records are flattened once into fp64 arrays; IoUs, the per-(video, sketch) score sort, the
greedy matching at the ten IoU thresholds and the interpolated AP run in two kernels (``svol_eval_ap``,
``svol_eval_max_iou``) that reproduce the reference's fp64 arithmetic bit for bit; the remaining means / threshold
counts over a few thousand numbers and the ``float(f'{x:.2f}')`` formatting are numpy, exactly as in the reference.
"""
from __future__ import annotations

import time
from collections import OrderedDict

import numpy as np
import torch

from .. import _lib
from ..ops import _ptr, _stream


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('svol_amd.evaluate runs on the MI355X only (no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


class _Packed:
    """results -> flat arrays (record order, then row order: the order every reference loop walks them in)."""

    def __init__(self, results):
        pb, po, gb, go, grec = [], [0], [], [0], []
        for r, res in enumerate(results):
            for p in res['pred_boxes']:
                pb.append(p)
            po.append(len(pb))
            for g in res['gt_boxes']:
                gb.append(g['bbox'])
                grec.append(r)
            go.append(len(gb))
        self.n_rec = len(results)
        self.pred = np.asarray(pb, dtype=np.float64).reshape(-1, 5)
        self.pred_off = np.asarray(po, dtype=np.int32)
        self.gt = np.asarray(gb, dtype=np.float64).reshape(-1, 4)
        self.gt_off = np.asarray(go, dtype=np.int32)
        self.gt_rec = np.asarray(grec, dtype=np.int32)
        # groups of compute_ap (eval.py:22-50): key video+sketch, created in order of first prediction
        order, members = OrderedDict(), {}
        for r, res in enumerate(results):
            key = res['video'] + res['sketch']
            if len(res['pred_boxes']) and key not in order:
                order[key] = len(order)
            members.setdefault(key, []).append(r)
        self.groups = list(order.keys())
        gp_idx, gg_idx, gp_off, gg_off, pf, gf = [], [], [0], [0], [], []
        for key in self.groups:
            frames = {}
            for r in members[key]:
                fid = frames.setdefault(results[r]['frame'], len(frames))
                a, b = self.pred_off[r], self.pred_off[r + 1]
                gp_idx.extend(range(a, b))
                pf.extend([fid] * (b - a))
                a, b = self.gt_off[r], self.gt_off[r + 1]
                gg_idx.extend(range(a, b))
                gf.extend([fid] * (b - a))
            gp_off.append(len(gp_idx))
            gg_off.append(len(gg_idx))
        gp_idx = np.asarray(gp_idx, dtype=np.int64)
        gg_idx = np.asarray(gg_idx, dtype=np.int64)
        self.ap_pred = self.pred[gp_idx] if len(gp_idx) else np.zeros((0, 5))
        self.ap_gt = self.gt[gg_idx] if len(gg_idx) else np.zeros((0, 4))
        self.ap_pred_frame = np.asarray(pf, dtype=np.int32)
        self.ap_gt_frame = np.asarray(gf, dtype=np.int32)
        self.ap_pred_off = np.asarray(gp_off, dtype=np.int32)
        self.ap_gt_off = np.asarray(gg_off, dtype=np.int32)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ap_arrays(pk: _Packed, iou_thds) -> np.ndarray:
    dev = _dev()
    NG, P, G, K = len(pk.groups), len(pk.ap_pred), len(pk.ap_gt), len(iou_thds)
    if NG == 0:
        return np.zeros((0, K))
    box = _t(pk.ap_pred[:, :4], dev)
    score = _t(pk.ap_pred[:, 4], dev)
    pf, po = _t(pk.ap_pred_frame, dev), _t(pk.ap_pred_off, dev)
    gt = _t(pk.ap_gt, dev) if G else torch.zeros((1, 4), dtype=torch.float64, device=dev)
    gf = _t(pk.ap_gt_frame, dev) if G else torch.zeros((1,), dtype=torch.int32, device=dev)
    go = _t(pk.ap_gt_off, dev)
    thr = _t(np.asarray(iou_thds, dtype=np.float64), dev)
    ws_order = torch.empty((max(P, 1),), dtype=torch.int32, device=dev)
    ws_u8 = torch.empty((max(K * (P + G), 1),), dtype=torch.uint8, device=dev)
    ws_f64 = torch.empty((2 * K * (P + 2 * NG),), dtype=torch.float64, device=dev)
    ap = torch.empty((NG, K), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().svol_eval_ap(_ptr(box), _ptr(score), _ptr(pf), _ptr(po), _ptr(gt), _ptr(gf), _ptr(go), _ptr(thr), K,
                                       _ptr(ws_order), _ptr(ws_u8), _ptr(ws_f64), _ptr(ap), P, G, NG, _stream()),
               'svol_eval_ap')
    return ap.cpu().numpy()


def _max_ious(pk: _Packed, k: int) -> np.ndarray:
    dev = _dev()
    G = len(pk.gt)
    if G == 0:
        return np.zeros((0,))
    cnt = np.diff(pk.pred_off)
    if np.any(cnt[pk.gt_rec] == 0):
        raise ValueError('zero-size array to reduction operation maximum which has no identity')  # numpy's error
    out = torch.empty((G,), dtype=torch.float64, device=dev)
    # named tensors: they must stay alive until the launch has been enqueued (a temporary's block would be recycled)
    pb, po = _t(pk.pred[:, :4], dev), _t(pk.pred_off, dev)
    gb, go, gr = _t(pk.gt, dev), _t(pk.gt_off, dev), _t(pk.gt_rec, dev)
    _lib.check(_lib.lib().svol_eval_max_iou(_ptr(pb), _ptr(po), _ptr(gb), _ptr(go), _ptr(gr), _ptr(out), G, int(k), _stream()),
               'svol_eval_max_iou')
    return out.cpu().numpy()


def compute_ap(results, iou_thds=np.linspace(0.5, 0.95, 10), num_workers=0, chunksize=50, _packed=None):
    """eval.py:19-69 (num_workers / chunksize only parallelised the reference's Python loop; ignored)."""
    iou_thds = [float(f'{e:.2f}') for e in iou_thds]
    pk = _packed or _Packed(results)
    ap_array = _ap_arrays(pk, iou_thds)  # (#groups, #thd)
    ap_thds = ap_array.mean(0)
    iou_thd2ap = dict(zip([str(e) for e in iou_thds], ap_thds))
    iou_thd2ap['average'] = np.mean(ap_thds)
    return {k: float(f'{100 * v:.2f}') for k, v in iou_thd2ap.items()}


def compute_recall_at_k(results, iou_thds=np.linspace(0.1, 0.9, 9), k=1, _packed=None):
    """eval.py:72-99."""
    pk = _packed or _Packed(results)
    max_ious = _max_ious(pk, k)
    iou_thd2recall_at_k = {}
    iou_thds = [float(f'{e:.2f}') for e in iou_thds]
    for thd in iou_thds:
        iou_thd2recall_at_k[str(thd)] = float(f'{np.mean(max_ious >= thd) * 100:.2f}')
    miou = float(f'{np.mean(max_ious) * 100:.2f}')
    return iou_thd2recall_at_k, miou


def eval_svol(results, verbose=True, logger=None):
    """eval.py:102-118."""
    if verbose:
        start_time = time.time()
    pk = _Packed(results)
    iou_thd2average_precision = compute_ap(results, num_workers=8, chunksize=50, _packed=pk)
    iou_thd2recall_at_one, miou_at_one = compute_recall_at_k(results, k=1, _packed=pk)
    iou_thd2recall_at_five, miou_at_five = compute_recall_at_k(results, k=5, _packed=pk)
    ret_metrics = {
        'SVOL-mAP': iou_thd2average_precision,
        'SVOL-R1': iou_thd2recall_at_one,
        'SVOL-R5': iou_thd2recall_at_five,
        'mIoU@R1': miou_at_one,
        'mIoU@R5': miou_at_five,
    }
    if verbose and logger is not None:
        logger.info(f'[eval_svol] {time.time() - start_time:.2f} seconds')
    return ret_metrics


def eval_results(results, verbose=True, logger=None, match_number=False):
    """eval.py:121-172: full metrics + the sorted 'brief' dictionary."""
    eval_metrics = {}
    eval_metrics_brief = OrderedDict()
    svol_scores = eval_svol(results, verbose=verbose, logger=logger)
    eval_metrics.update(svol_scores)
    svol_scores_brief = {'SVOL-full-mAP': svol_scores['SVOL-mAP']['average']}
    for kk in ('R1', 'R5'):
        for t in ('0.1', '0.3', '0.5', '0.7'):
            svol_scores_brief[f'SVOL-full-{kk}@{t}'] = svol_scores[f'SVOL-{kk}'][t]
    svol_scores_brief['SVOL-full-mIoU@R1'] = svol_scores['mIoU@R1']
    svol_scores_brief['SVOL-full-mIoU@R5'] = svol_scores['mIoU@R5']
    eval_metrics_brief.update(sorted([(k, v) for k, v in svol_scores_brief.items()], key=lambda x: x[0]))
    final_eval_metrics = OrderedDict()
    final_eval_metrics['brief'] = eval_metrics_brief
    final_eval_metrics.update(sorted([(k, v) for k, v in eval_metrics.items()], key=lambda x: x[0]))
    return final_eval_metrics
