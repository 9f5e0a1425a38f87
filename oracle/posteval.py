"""ORACLE — test infrastructure only (see oracle/__init__.py).

CPU restatement of the step right AFTER the hot path in the reference's ``test.py``:

* ``compose_results``  — test.py:133-169: softmax score of the foreground class, cxcywh -> xyxy clamped to
  [0, 1], ``chunk(num_frames)``, per-chunk stable sort by score (descending), every number rounded through
  ``f'{e:.4f}'``, ground-truth boxes as xyxy lists; one record per (video, frame) with the keys of the
  reference's JSONL wire format (test.py:158-166).
* ``eval_results`` and helpers — lib/evaluate/eval.py:12-172 and lib/evaluate/utils.py:15-201: pairwise IoU in
  fp64, Pascal-VOC style AP per (video, sketch) at IoU thresholds 0.5:0.05:0.95 with greedy one-to-one locking,
  recall@k / mIoU@k over frames, the "brief" dictionary and the ``float(f'{x:.2f}')`` formatting.

The evaluation half is pinned against the reference's own ``lib.evaluate.eval`` (importable in the build
container) by ``tests/golden/make_golden_posteval.py`` -> ``tests/golden/posteval_*.json``; the composition half
has no importable counterpart (it is inline in ``test.py``, which needs apex) and is pinned through
``lib.utils.box_utils.box_cxcywh_to_xyxy`` plus the wire-format example in eval.py:122-147.
"""
from collections import OrderedDict, defaultdict

import numpy as np
import torch
import torch.nn.functional as F


# ---- test.py:133-169 -----------------------------------------------------------------------------
def box_cxcywh_to_xyxy(x):  # lib/utils/box_utils.py:9-13
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def compose_results(outputs, targets, num_frames, no_sort_results=False):
    prob = F.softmax(outputs['pred_logits'], -1)
    scores = prob[..., 0]
    pred_boxes = outputs['pred_boxes']
    results = []
    for target, boxes, score in zip(targets, pred_boxes.cpu(), scores.cpu()):
        frame_idxs = list(target['bboxes'].keys())
        boxes = torch.clamp(box_cxcywh_to_xyxy(boxes), min=0, max=1)
        preds = torch.cat([boxes, score[:, None]], dim=1)
        preds = preds.chunk(num_frames, dim=0)
        for preds_per_frame, fidx in zip(preds, frame_idxs):
            if no_sort_results:
                # the reference leaves `sorted_preds` unbound here (test.py:151-154) and dies with
                # UnboundLocalError on the next line; kept as an error, not silently "fixed"
                raise UnboundLocalError("local variable 'sorted_preds' referenced before assignment")
            sorted_preds = sorted(preds_per_frame, key=lambda x: x[4], reverse=True)
            sorted_preds = [[float(f'{e:.4f}') for e in row] for row in sorted_preds]
            gt_boxes = [{'track_id': ib['track_id'], 'bbox': box_cxcywh_to_xyxy(ib['bbox']).tolist()}
                        for ib in target['bboxes'][fidx]]
            results.append(dict(video=target['video'], sketch=target['sketch'], shape=target['size'], frame=fidx,
                                gt_boxes=gt_boxes, pred_boxes=sorted_preds))
    return results


# ---- lib/evaluate/utils.py -----------------------------------------------------------------------
def box_area(c):  # utils.py:15-32
    return (c[..., 2] - c[..., 0]) * (c[..., 3] - c[..., 1])


def compute_iou_batch_paired(box1, box2):  # utils.py:35-71
    xmin = np.maximum(box1[..., 0], box2[..., 0])
    ymin = np.maximum(box1[..., 1], box2[..., 1])
    xmax = np.minimum(box1[..., 2], box2[..., 2])
    ymax = np.minimum(box1[..., 3], box2[..., 3])
    inter = box_area(np.stack([xmin, ymin, xmax, ymax], axis=-1))
    union = (box_area(box1) + box_area(box2)) - inter
    valid = np.logical_and(xmin <= xmax, ymin <= ymax)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(valid, inter / union, 0)


def compute_iou_batch_cross(box1, box2):  # utils.py:74-98
    N, M = box1.shape[0], box2.shape[0]
    b1 = np.tile(box1, (M, 1))
    b2 = np.repeat(box2, N, axis=0)
    # NB (reference behaviour, kept): tile/repeat lay the pairs out as [m][n] but the result is reshaped to (N, M);
    # every caller has N == 1 (AP) or reduces over axis 0 of a (k, M) matrix (recall@k)
    return compute_iou_batch_paired(b1, b2).reshape(N, M)


def interpolated_precision_recall(precision, recall):  # utils.py:101-118
    mprecision = np.hstack([[0], precision, [0]])
    mrecall = np.hstack([[0], recall, [1]])
    for i in range(len(mprecision) - 1)[::-1]:
        mprecision[i] = max(mprecision[i], mprecision[i + 1])
    idx = np.where(mrecall[1::] != mrecall[0:-1])[0] + 1
    return np.sum((mrecall[idx] - mrecall[idx - 1]) * mprecision[idx])


def compute_average_precision_detection(ground_truth, prediction, iou_thresholds=np.linspace(0.5, 0.95, 10)):
    """utils.py:121-201."""
    K, N, M = len(iou_thresholds), len(ground_truth), len(prediction)
    ap = np.zeros(K)
    if M == 0:
        return ap
    num_positive = float(N)
    lock_gt = np.ones((K, N)) * -1
    prediction.sort(key=lambda x: -x['score'])
    tp = np.zeros((K, M))
    fp = np.zeros((K, M))
    by_frame = {}
    for i, item in enumerate(ground_truth):
        item['index'] = i
        by_frame.setdefault(item['frame'], []).append(item)
    for idx, pred in enumerate(prediction):
        if pred['frame'] in by_frame:
            gts = by_frame[pred['frame']]
        else:
            fp[:, idx] = 1
            continue
        _pred = np.array([[pred['top-left-x'], pred['top-left-y'], pred['bot-right-x'], pred['bot-right-y']]])
        _gt = np.array([[g['top-left-x'], g['top-left-y'], g['bot-right-x'], g['bot-right-y']] for g in gts])
        iou_arr = compute_iou_batch_cross(_pred, _gt).reshape(-1)
        order = iou_arr.argsort()[::-1]
        for t_idx, thr in enumerate(iou_thresholds):
            for j in order:
                if iou_arr[j] < thr:
                    fp[t_idx, idx] = 1
                    break
                if lock_gt[t_idx, gts[j]['index']] >= 0:
                    continue
                tp[t_idx, idx] = 1
                lock_gt[t_idx, gts[j]['index']] = idx
                break
            if fp[t_idx, idx] == 0 and tp[t_idx, idx] == 0:
                fp[t_idx, idx] = 1
    tp_c = np.cumsum(tp, axis=1).astype(float)
    fp_c = np.cumsum(fp, axis=1).astype(float)
    recall = tp_c / num_positive
    precision = tp_c / (tp_c + fp_c)
    for t in range(K):
        ap[t] = interpolated_precision_recall(precision[t, :], recall[t, :])
    return ap


# ---- lib/evaluate/eval.py ------------------------------------------------------------------------
def compute_ap(results, iou_thds=np.linspace(0.5, 0.95, 10)):  # eval.py:19-69 (the pool only parallelises)
    iou_thds = [float(f'{e:.2f}') for e in iou_thds]
    preds, gts = defaultdict(list), defaultdict(list)
    for res in results:
        key = res['video'] + res['sketch']
        for p in res['pred_boxes']:
            preds[key].append({'frame': res['frame'], 'top-left-x': p[0], 'top-left-y': p[1], 'bot-right-x': p[2],
                               'bot-right-y': p[3], 'score': p[4]})
        for g in res['gt_boxes']:
            gts[key].append({'frame': res['frame'], 'top-left-x': g['bbox'][0], 'top-left-y': g['bbox'][1],
                             'bot-right-x': g['bbox'][2], 'bot-right-y': g['bbox'][3]})
    video2ap = {v: compute_average_precision_detection(gts[v], preds[v], iou_thresholds=iou_thds) for v in preds}
    ap_array = np.array(list(video2ap.values()))
    ap_thds = ap_array.mean(0)
    out = dict(zip([str(e) for e in iou_thds], ap_thds))
    out['average'] = np.mean(ap_thds)
    return {k: float(f'{100 * v:.2f}') for k, v in out.items()}, ap_array


def compute_recall_at_k(results, iou_thds=np.linspace(0.1, 0.9, 9), k=1):  # eval.py:72-99
    max_ious = []
    for res in results:
        gts = [e['bbox'] for e in res['gt_boxes']]
        if len(gts) == 0:
            continue
        iou = compute_iou_batch_cross(np.array(res['pred_boxes'][:k]), np.array(gts))
        max_ious.extend(iou.max(axis=0))
    max_ious = np.asarray(max_ious)
    thds = [float(f'{e:.2f}') for e in iou_thds]
    recall = {str(t): float(f'{np.mean(max_ious >= t) * 100:.2f}') for t in thds}
    return recall, float(f'{np.mean(max_ious) * 100:.2f}'), max_ious


def eval_svol(results):  # eval.py:102-118
    ap, _ = compute_ap(results)
    r1, m1, _ = compute_recall_at_k(results, k=1)
    r5, m5, _ = compute_recall_at_k(results, k=5)
    return {'SVOL-mAP': ap, 'SVOL-R1': r1, 'SVOL-R5': r5, 'mIoU@R1': m1, 'mIoU@R5': m5}


def eval_results(results):  # eval.py:121-172
    scores = eval_svol(results)
    brief = {'SVOL-full-mAP': scores['SVOL-mAP']['average']}
    for k in ('R1', 'R5'):
        for t in ('0.1', '0.3', '0.5', '0.7'):
            brief[f'SVOL-full-{k}@{t}'] = scores[f'SVOL-{k}'][t]
    brief['SVOL-full-mIoU@R1'] = scores['mIoU@R1']
    brief['SVOL-full-mIoU@R5'] = scores['mIoU@R5']
    final = OrderedDict()
    final['brief'] = OrderedDict(sorted(brief.items(), key=lambda x: x[0]))
    final.update(sorted(scores.items(), key=lambda x: x[0]))
    return final
