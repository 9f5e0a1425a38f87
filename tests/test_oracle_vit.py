"""oracle/vit_oracle.py against outputs of the Hugging Face ViTModel (tests/golden/make_golden_vit.py)."""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import vit_oracle as V
from svol_amd import synthetic as syn

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, 'golden', 'vit_*.npz')))


def load(path):
    z = np.load(path)
    cfg = syn.vit_config(**ast.literal_eval(str(z['over'])))
    return z, cfg, syn.synth_vit_state_dict(cfg, int(z['seed'])), syn.synth_images(int(z['n']), cfg, int(z['seed'])), int(z['stride'])


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(p)[4:-4] for p in FIXTURES])
def test_vit_oracle_matches_transformers(path):
    z, cfg, sd, x, st = load(path)
    with torch.no_grad():
        last, pre = V.vit_forward(sd, cfg, x)
    assert np.abs(last[:, ::st].numpy() - z['last_hidden_state']).max() < 2e-5
    assert np.abs(pre[:, ::st].numpy() - z['pre_norm']).max() < 5e-5
