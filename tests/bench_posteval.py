#!/usr/bin/env python3
"""Post-processing + evaluation: device path vs the CPU oracle (= the reference's numpy code path,
proven equal by the goldens) on a synthetic validation set.  Lives under tests/ because it runs the oracle.
    python tests/bench_posteval.py [videos]"""
import copy
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import posteval as O  # noqa: E402
from svol_amd import postprocess as PP  # noqa: E402
from svol_amd import synthetic as syn  # noqa: E402
from svol_amd.evaluate import eval as E  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T, N = 32, 320
tg = syn.synth_targets(V, T, seed=11)
for i, t in enumerate(tg):
    t['video'] = f'vid_{i:05d}'
logits, boxes = syn.synth_eval_outputs(tg, N, T, seed=11)
out = {'pred_logits': logits.cuda(), 'pred_boxes': boxes.cuda()}
PP.compose_results({k: v[:2] for k, v in out.items()}, tg[:2], T)  # warm-up
torch.cuda.synchronize()
t0 = time.time(); res = PP.compose_results(out, tg, T); torch.cuda.synchronize(); t_pp = time.time() - t0
t0 = time.time(); ref = O.compose_results({'pred_logits': logits, 'pred_boxes': boxes}, tg, T); t_pp_cpu = time.time() - t0
E.eval_results(copy.deepcopy(res[:64]), verbose=False)  # warm-up
t0 = time.time(); pk = E._Packed(res); t_pack = time.time() - t0
t0 = time.time(); m = E.eval_results(res, verbose=False); torch.cuda.synchronize(); t_ev = time.time() - t0
t0 = time.time(); m_ref = O.eval_results(copy.deepcopy(res)); t_ev_cpu = time.time() - t0
import json  # noqa: E402
print(f'{V} videos x {T} frames x {N // T} predictions/frame: {len(res)} records, {V * N} predictions')
print(f'compose_results: device {t_pp * 1e3:.0f} ms   CPU oracle {t_pp_cpu * 1e3:.0f} ms')
print(f'eval_results   : device {t_ev * 1e3:.0f} ms (of which flattening the records {t_pack * 1e3:.0f} ms)   CPU oracle '
      f'{t_ev_cpu * 1e3:.0f} ms   -> {t_ev_cpu / t_ev:.1f}x')
print('metrics identical:', json.dumps(m) == json.dumps(m_ref), m['brief']['SVOL-full-mAP'])
