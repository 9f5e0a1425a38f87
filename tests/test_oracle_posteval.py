"""The post-processing / evaluation oracle (oracle/posteval.py) against fixtures produced by the reference's own
lib.evaluate.eval (tests/golden/make_golden_posteval.py): metrics dictionaries identical, per-group AP arrays and
per-box max-IoU vectors bit-exact (fp64), xyxy conversion of the target boxes identical."""
import copy
import glob
import json
import os

import numpy as np
import pytest

from oracle import posteval as O
from svol_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, 'golden', 'posteval_*.json')))


def load_case(path):
    fx = json.load(open(path))
    c = fx['case']
    tg = syn.synth_targets(c['B'], c['T'], seed=c['seed'])
    logits, boxes = syn.synth_eval_outputs(tg, c['N'], c['T'], seed=c['seed'], tie_every=c['tie_every'])
    return fx, c, tg, logits, boxes


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(p)[9:-5] for p in FIXTURES])
def test_oracle_matches_reference_eval(path):
    fx, c, tg, logits, boxes = load_case(path)
    results = O.compose_results({'pred_logits': logits, 'pred_boxes': boxes}, tg, c['T'])
    assert len(results) == fx['n_records']
    # wire format (test.py:158-166): keys, 4-decimal rounding, per-frame descending scores
    r0 = results[0]
    assert list(r0.keys()) == ['video', 'sketch', 'shape', 'frame', 'gt_boxes', 'pred_boxes']
    for r in results:
        sc = [p[4] for p in r['pred_boxes']]
        assert sc == sorted(sc, reverse=True)
        assert all(float(f'{e:.4f}') == e for p in r['pred_boxes'] for e in p)
    got_xyxy = [[O.box_cxcywh_to_xyxy(ib['bbox']).tolist() for f in t['bboxes'] for ib in t['bboxes'][f]] for t in tg]
    assert got_xyxy == fx['gt_xyxy']
    metrics = O.eval_results(copy.deepcopy(results))
    assert json.loads(json.dumps(metrics)) == fx['metrics']
    _, ap_array = O.compute_ap(copy.deepcopy(results))
    assert ap_array.tolist() == list(fx['ap'].values())
    for k in (1, 5):
        _, _, mi = O.compute_recall_at_k(results, k=k)
        assert np.asarray(mi).tolist() == fx['max_ious'][str(k)]


def test_no_sort_results_fails_like_the_reference():
    tg = syn.synth_targets(1, 4, seed=1)
    logits, boxes = syn.synth_eval_outputs(tg, 8, 4, seed=1)
    with pytest.raises(UnboundLocalError):
        O.compose_results({'pred_logits': logits, 'pred_boxes': boxes}, tg, 4, no_sort_results=True)
