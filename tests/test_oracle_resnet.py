"""Pin the ResNet oracle (oracle/resnet_oracle.py) against fixtures from transformers.ResNetModel (basic layers), and the build's
module tree against the torchvision key table (what a reference checkpoint's backbone.* entries look like)."""
import json

import numpy as np
import pytest
import torch

from oracle import resnet_oracle as R
from svol_amd import synthetic as syn
from tests.helpers import load_golden

CASES = ['resnet_tiny', 'resnet_tiny3', 'resnet18_1img', 'resnet34_1img']


def resnet_case(name):
    z, meta = load_golden(name)
    shapes = syn.resnet_param_shapes(tuple(meta['depths']), tuple(meta['widths']), meta['stem'])
    sd = syn.synth_resnet_state_dict(shapes, seed=1)
    x = syn.synth_images(meta['n'], syn.vit_config(image_size=meta['size']), seed=1)
    return z, meta, sd, x


@pytest.mark.parametrize('name', CASES)
def test_oracle_vs_hf_resnet(name):
    z, meta, sd, x = resnet_case(name)
    fmap = R.resnet_features(sd, meta['depths'], x)
    tok = fmap.flatten(2).transpose(1, 2)
    scale = float(np.abs(z['tokens']).max())
    assert float((tok - torch.from_numpy(z['tokens'])).abs().max()) <= 2e-5 * scale
    pooled = R.resnet_features(sd, meta['depths'], x, avgpool=True).flatten(1)
    assert float((pooled - torch.from_numpy(z['pooled'])).abs().max()) <= 2e-5 * scale


def test_module_keys_match_torchvision_table():
    from svol_amd.modeling.resnet import ResNetExtractor, resnet18, resnet34
    for m, depths in ((resnet18(), (2, 2, 2, 2)), (resnet34(), (3, 4, 6, 3))):
        want = syn.resnet_param_shapes(depths)
        got = m.state_dict()
        assert list(got.keys()) == list(want.keys())
        assert all(tuple(got[k].shape) == tuple(want[k]) for k in want)
    assert len(resnet34().state_dict()) == 216  # torchvision resnet34 without fc.weight / fc.bias
    m = ResNetExtractor((1, 2), (16, 32), 16)
    assert list(m.state_dict().keys()) == list(syn.resnet_param_shapes((1, 2), (16, 32), 16).keys())


def test_backbone_token_order():
    """ResNetBackbone.forward's reshape chain (backbone.py:82-87) = frames, then rows, then columns."""
    sd = syn.synth_resnet_state_dict(syn.resnet_param_shapes((1,), (8,), 8), seed=1)
    vid = syn.synth_images(4, syn.vit_config(image_size=32), seed=2).view(2, 2, 3, 32, 32)
    sk = syn.synth_images(2, syn.vit_config(image_size=32), seed=3).view(2, 1, 3, 32, 32)
    s, v = R.resnet_backbone_forward(sd, (1,), sd, (1,), sk, vid)
    f = R.resnet_features(sd, (1,), vid.flatten(0, 1))  # [4, 8, 8, 8]
    assert s.shape == (2, 1, 8) and v.shape == (2, 2 * 64, 8)
    assert torch.equal(v[1, 64 + 8 * 3 + 5], f[3, :, 3, 5])


TRAIN_CASES = ['resnet_tiny_train', 'resnet_tiny3_train']


@pytest.mark.parametrize('name', TRAIN_CASES)
def test_oracle_training_mode_vs_hf_resnet(name):
    """The TRAINING-mode restatement (batch-statistics BatchNorm, autograd) against transformers.ResNetModel.train() in fp64: features,
    every parameter gradient of loss = sum(features * probe), and the updated running statistics (tests/golden/make_golden_resnet.py)."""
    z, meta, sd, x = resnet_case(name)
    probe = torch.from_numpy(z['probe'])
    f, grads, stats = R.resnet_train_grads(sd, meta['depths'], x, probe)
    assert float((f - torch.from_numpy(z['features'])).abs().max()) <= 1e-9 * float(np.abs(z['features']).max())
    keys = [k[5:] for k in z.files if k.startswith('grad/')]
    assert sorted(keys) == sorted(grads.keys())
    for k in keys:
        ref = torch.from_numpy(z['grad/' + k])
        assert float((grads[k] - ref).abs().max()) <= 1e-8 * max(1.0, float(ref.abs().max())), k
    skeys = [k[5:] for k in z.files if k.startswith('stat/')]
    assert len(skeys) > 0 and sorted(skeys) == sorted(stats.keys())
    for k in skeys:
        ref = torch.from_numpy(z['stat/' + k])
        assert float((stats[k].double() - ref).abs().max()) <= 1e-9 * max(1.0, float(ref.abs().max())), k
