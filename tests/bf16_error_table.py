#!/usr/bin/env python3
"""Measured bf16 output error of the whole head against the reference's golden vectors, per case (markdown table on stdout):
max |pred_logits - golden|, max |pred_boxes - golden| of the final layer and of the auxiliary layers, and the largest golden
logit for scale.  Run on the MI355X: `python tests/bf16_error_table.py` (SVOL_QUERY_BF16=1 for the all-bf16 query stream)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_checks as G  # noqa: E402

CASES = ['tiny_video', 'tiny_frame', 'cfg1_video', 'cfg1_frame', 'mid_video', 'mid32_video', 'cfg2_b1_video', 'cfg2_b1_video_pad']


def main():
    mode = 'query stream bf16 (SVOL_QUERY_BF16=1)' if os.environ.get('SVOL_QUERY_BF16') else 'query stream fp32 (default)'
    dt = torch.float16 if os.environ.get('SVOL_TABLE_DTYPE') == 'fp16' else torch.bfloat16
    split = 'single-bf16 V weights (SVOL_NO_SPLIT_V=1)' if os.environ.get('SVOL_NO_SPLIT_V') else 'split hi + lo V weights (default)'
    print(f'{"fp16" if dt == torch.float16 else "bf16"} compute mode, {mode}' + (f', {split}' if dt == torch.bfloat16 else '') + '\n')
    print('| case | max abs logit err (final) | max abs box err (final) | aux logits | aux boxes | rms logit err | max golden logit |')
    print('|---|---|---|---|---|---|---|')
    for name in CASES:
        z, meta, args, out, ld, tot, model, crit = G.run_head_case(name, dt)
        dl = out['pred_logits'].cpu() - torch.from_numpy(z['pred_logits'])
        db = out['pred_boxes'].cpu() - torch.from_numpy(z['pred_boxes'])
        al = ab = float('nan')
        if 'aux_logits' in z.files:
            al = float((torch.stack([a['pred_logits'] for a in out['aux_outputs']]).cpu() - torch.from_numpy(z['aux_logits'])).abs().max())
            ab = float((torch.stack([a['pred_boxes'] for a in out['aux_outputs']]).cpu() - torch.from_numpy(z['aux_boxes'])).abs().max())
        print(f'| {name} | {float(dl.abs().max()):.2e} | {float(db.abs().max()):.2e} | {al:.2e} | {ab:.2e} | '
              f'{float(dl.pow(2).mean().sqrt()):.2e} | {float(np.abs(z["pred_logits"]).max()):.2f} |', flush=True)


if __name__ == '__main__':
    main()
