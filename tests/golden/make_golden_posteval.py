#!/usr/bin/env python3
"""Golden fixtures for the post-processing + evaluation row, generated FROM THE REFERENCE.

Runs only in the build container.  Imports the reference's ``lib.evaluate.eval`` / ``lib.evaluate.utils`` /
``lib.utils.box_utils`` (plain numpy / torch; scikit-learn is installed) and stores DATA ONLY:

* the metrics dictionary ``eval_results`` returns for records composed from deterministic synthetic outputs
  (``svol_amd.synthetic.synth_eval_outputs``), the per-(video, sketch) AP arrays of
  ``compute_average_precision_detection`` and the per-box max-IoU vectors behind recall@1 / recall@5 (obtained by
  running the reference's own ``compute_iou_batch_cross`` exactly as ``compute_recall_at_k`` does), all as JSON
  (Python's float repr round-trips fp64 exactly);
* ``box_cxcywh_to_xyxy`` of the synthetic target boxes (pins the box half of the record composition).

The records themselves are NOT stored: tests regenerate them with ``oracle.posteval.compose_results``.

    python tests/golden/make_golden_posteval.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = os.environ.get('SVOL_REFERENCE', '/root/reference')
sys.path.insert(0, REF)
_tv = types.ModuleType('torchvision'); _ops = types.ModuleType('torchvision.ops'); _boxes = types.ModuleType('torchvision.ops.boxes')
_boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
_tv.ops = _ops; _ops.boxes = _boxes
for k, v in (('torchvision', _tv), ('torchvision.ops', _ops), ('torchvision.ops.boxes', _boxes)):
    sys.modules.setdefault(k, v)

from lib.evaluate import eval as ref_eval          # noqa: E402
from lib.evaluate import utils as ref_utils        # noqa: E402
from lib.utils import box_utils as ref_box         # noqa: E402

from oracle import posteval as O                   # noqa: E402
from svol_amd import synthetic as syn              # noqa: E402

CASES = {  # name: (B, T, N, seed, tie_every)
    'video_B8_T32_N100': (8, 32, 100, 1, 0),
    'frame_B4_T32_N320': (4, 32, 320, 2, 0),
    'tiny_B2_T4_N10_ties': (2, 4, 10, 3, 3),
}


class _Log:
    def info(self, *a, **k):
        pass


def main():
    for name, (B, T, N, seed, ties) in CASES.items():
        tg = syn.synth_targets(B, T, seed=seed)
        logits, boxes = syn.synth_eval_outputs(tg, N, T, seed=seed, tie_every=ties)
        results = O.compose_results({'pred_logits': logits, 'pred_boxes': boxes}, tg, T)
        import copy
        metrics = ref_eval.eval_results(copy.deepcopy(results), verbose=False, logger=_Log())
        # per-group AP arrays, grouped exactly as compute_ap does (eval.py:22-50)
        preds, gts = {}, {}
        for res in results:
            key = res['video'] + res['sketch']
            preds.setdefault(key, []); gts.setdefault(key, [])
            for p in res['pred_boxes']:
                preds[key].append({'frame': res['frame'], 'top-left-x': p[0], 'top-left-y': p[1], 'bot-right-x': p[2],
                                   'bot-right-y': p[3], 'score': p[4]})
            for g in res['gt_boxes']:
                gts[key].append({'frame': res['frame'], 'top-left-x': g['bbox'][0], 'top-left-y': g['bbox'][1],
                                 'bot-right-x': g['bbox'][2], 'bot-right-y': g['bbox'][3]})
        thds = [float(f'{e:.2f}') for e in np.linspace(0.5, 0.95, 10)]
        ap = {k: ref_utils.compute_average_precision_detection(gts[k], preds[k], iou_thresholds=thds).tolist() for k in preds}
        max_ious = {}
        for k in (1, 5):
            out = []
            for res in results:
                g = [e['bbox'] for e in res['gt_boxes']]
                if not g:
                    continue
                iou = ref_utils.compute_iou_batch_cross(np.array(res['pred_boxes'][:k]), np.array(g))
                out.extend(iou.max(axis=0).tolist())
            max_ious[str(k)] = out
        gt_xyxy = [[ref_box.box_cxcywh_to_xyxy(ib['bbox']).tolist() for f in t['bboxes'] for ib in t['bboxes'][f]] for t in tg]
        fx = {'case': dict(B=B, T=T, N=N, seed=seed, tie_every=ties), 'metrics': metrics, 'ap': ap, 'max_ious': max_ious,
              'gt_xyxy': gt_xyxy, 'n_records': len(results)}
        with open(os.path.join(HERE, f'posteval_{name}.json'), 'w') as f:
            json.dump(fx, f)
        print(name, 'records', len(results), 'mAP', metrics['brief']['SVOL-full-mAP'], 'R1@0.5', metrics['brief']['SVOL-full-R1@0.5'])


if __name__ == '__main__':
    main()
