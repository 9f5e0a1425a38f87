"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol that
include/svol_hip.h declares (no compute calls without a GPU), and the product refuses to run on CPU."""
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(REPO, 'include', 'svol_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(svol_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_exported():
    from svol_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f'{n} declared in include/svol_hip.h but not exported'
    assert set(_lib.SIGNATURES) | {'svol_abi_version', 'svol_strerror', 'svol_block_slot_names'} == set(names)
    # the composite block programs reject null tables before any launch
    for fn in ('svol_video_half_fwd', 'svol_query_self_fwd', 'svol_query_cross_fwd', 'svol_query_cross_wgrad'):
        assert getattr(L, fn)(0, 0, 0) == -1
    for blk, first in ((0, 'X32'), (1, 'O32'), (2, 'O32')):
        assert L.svol_block_slot_names(blk).decode().split(',')[0] == first
    assert L.svol_abi_version() == 7
    assert b'invalid' in L.svol_strerror(-1)


def _host_lsap(cost):
    import ctypes
    import numpy as np
    from svol_amd import _lib
    c = np.ascontiguousarray(cost, dtype=np.float64)
    nr, nc = c.shape
    n = min(nr, nc)
    rows, cols = np.full(max(n, 1), -7, np.int64), np.full(max(n, 1), -7, np.int64)
    rc = _lib.lib().svol_lsap_solve(c.ctypes.data_as(ctypes.c_void_p), nr, nc, rows.ctypes.data_as(ctypes.c_void_p),
                                    cols.ctypes.data_as(ctypes.c_void_p))
    return rc, rows[:n], cols[:n]


def test_host_lsap_known_answers():
    """svol_lsap_solve (host entry of SURVEY §8b, svol_amd/csrc/lsap_host.hip — not the oracle's file) against the scipy 1.15.3
    known answers of tests/golden/lsap_known_answers.npz (ties, zeros, tall / wide / empty, +inf), SURVEY §8c's vectors, live
    random comparisons with scipy incl. heavy ties, and scipy's error conventions (NaN / -inf invalid, all-inf row infeasible)."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    z = np.load(os.path.join(REPO, 'tests', 'golden', 'lsap_known_answers.npz'))
    for i in range(int(z['n'])):
        c = z[f'c{i}/cost']
        rc, r, k = _host_lsap(c)
        assert rc == len(z[f'c{i}/rows']), (i, rc)
        assert r.tolist() == z[f'c{i}/rows'].tolist() and k.tolist() == z[f'c{i}/cols'].tolist(), (i, str(z[f'c{i}/kind']))
    for c, rows, cols in [(np.zeros((4, 2)), [0, 1], [0, 1]), (np.zeros((2, 4)), [0, 1], [0, 1]), (np.zeros((3, 3)), [0, 1, 2], [0, 1, 2]),
                          (np.array([[1, 2], [1, 2], [0, 2], [1, 0], [1, 0]], float), [2, 3], [0, 1]),
                          (np.array([[.3, .1], [.1, .3], [.2, .2]], np.float32), [0, 1], [1, 0]),
                          (np.random.default_rng(0).random((10, 3)).astype(np.float32), [1, 3, 4], [0, 2, 1])]:
        rc, r, k = _host_lsap(c)
        assert rc == len(rows) and r.tolist() == rows and k.tolist() == cols
    assert _host_lsap(np.zeros((10, 0)))[0] == 0
    rng = np.random.RandomState(11)
    for it in range(300):
        nr, nc = rng.randint(1, 90), rng.randint(1, 90)
        c = (rng.randint(0, 3, size=(nr, nc)).astype(np.float32) if it % 3 == 0 else rng.random_sample((nr, nc)).astype(np.float32)
             if it % 3 == 1 else rng.standard_normal((nr, nc)))
        rr, cc = linear_sum_assignment(c)
        rc, r, k = _host_lsap(c)
        assert rc == len(rr) and r.tolist() == rr.tolist() and k.tolist() == cc.tolist(), (it, nr, nc)
    bad = np.ones((3, 3))
    bad[1, 1] = np.nan
    assert _host_lsap(bad)[0] == -1
    bad[1, 1] = -np.inf
    assert _host_lsap(bad)[0] == -1
    bad[1, :] = np.inf
    assert _host_lsap(bad)[0] == -2
    ok = np.array([[np.inf, 1.0], [2.0, np.inf], [0.5, 0.25]])
    rr, cc = linear_sum_assignment(ok)
    rc, r, k = _host_lsap(ok)
    assert rc == 2 and r.tolist() == rr.tolist() and k.tolist() == cc.tolist()


def test_argument_validation_without_gpu():
    """Null pointers / bad shapes are rejected before any launch (safe to call without a device)."""
    from svol_amd import _lib
    L = _lib.lib()
    assert L.svol_gemm_nt(0, 8, 0, 0, 0, 8, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 8, 1, 0) == -1
    assert L.svol_attn_fwd(0, 8, 0, 8, 0, 8, 0, 8, 0, 0, 1, 1, 1, 1, 8, 1.0, 0.0, 0, 0, 1, 0) == -1
    # few queries: key-split partials; many queries: one int per (batch, 128-key tile) for the masked fast kernels or one per
    # workgroup (batch, head, 128-query tile) for the fast forward's redo flags, whichever is larger
    assert L.svol_attn_ws_bytes(8, 8, 100, 6272, 32) > 8 * 49 * 4 and L.svol_attn_ws_bytes(8, 8, 6272, 6272, 32) == (8 * 6272 * 256 + 64 * 4 * 2 * 128 * 32) * 4   # single-pass backward: fp32 dQ image + tail partials
    assert L.svol_cast(0, 0, 0, 1, 10, 0) == -1


def test_reference_style_cli_trains_the_resnet_backbone_by_default():
    """the reference's shipped runs (train_*.sh: --backbone resnet; train.py:72 optimises every parameter under
    model.train(); its --freeze_backbone is parsed and never read) train the backbone.  A reference command line through
    svol_amd.configs + build_model must therefore build TRAINABLE extractors by default; --freeze_backbone (honoured here) and
    --train_backbone 0 select the frozen ones; --sync_bn reaches the extractors (train.py:65-68)."""
    from svol_amd import configs
    from svol_amd.modeling.model import build_model
    base = ['--backbone', 'resnet', '--num_layers', '1', '--num_queries', '10', '--matcher', 'video_matcher']
    for argv, want_train, want_sync in [(base, True, False), (base + ['--freeze_backbone'], False, False),
                                        (base + ['--train_backbone', '0'], False, False),
                                        (base + ['--freeze_backbone', '--train_backbone', '1', '--sync_bn'], True, True)]:
        args = configs.parse_args(argv)
        model = build_model(args)
        bb = list(model.backbone.parameters())
        assert len(bb) == 108 + 60
        assert all(p.requires_grad == want_train for p in bb), argv
        assert model.backbone.video_backbone.trainable == want_train and model.backbone.sketch_backbone.trainable == want_train
        assert model.backbone.video_backbone.sync_bn == want_sync and model.backbone.sketch_backbone.sync_bn == want_sync
        n_opt = sum(1 for p in model.parameters() if p.requires_grad)   # train.py:72's list
        assert n_opt == sum(1 for p in model.head.parameters() if p.requires_grad) + (len(bb) if want_train else 0)


def test_product_has_no_cpu_path():
    from svol_amd import synthetic as syn
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.svanet import build_svanet
    a = syn.head_args(hidden_dim=32, nheads=4, num_layers=1, num_queries=8, num_frames=4, input_vid_dim=32,
                      input_skch_dim=32)
    m = build_svanet(a)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 32), torch.ones(1, 1), torch.zeros(1, 8, 32), torch.ones(1, 8))
    c = build_loss(a)
    with pytest.raises(RuntimeError):
        c({'pred_logits': torch.zeros(1, 8, 2), 'pred_boxes': torch.rand(1, 8, 4)}, syn.synth_targets(1, 4))


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = ('import sys; import svol_amd, svol_amd.ops, svol_amd.modeling.model, svol_amd.modeling.loss, '
            'svol_amd.configs; assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules), '
            '"product imported oracle"')
    subprocess.check_call([sys.executable, '-c', code], cwd=REPO)


def test_state_dict_keys_match_reference():
    from svol_amd import synthetic as syn
    from svol_amd.modeling.svanet import build_svanet
    from tests.helpers import load_golden
    z, meta = load_golden('head_mid_video')
    a = syn.head_args(**meta['args'])
    m = build_svanet(a)
    assert list(m.state_dict().keys()) == str(z['keys']).split('\n')


def test_option_surface_defaults():
    import json
    from svol_amd import configs
    ref = json.load(open(os.path.join(REPO, 'tests', 'golden', 'configs_defaults.json')))
    assert configs.reference_defaults() == ref
    a = configs.parse_args(['--num_layers', '6', '--matcher', 'video_matcher', '--num_queries', '100'])
    assert a.num_layers == 6 and a.matcher == 'video_matcher' and a.compute_dtype == 'bf16'


def test_driver_build_entry_agrees_with_the_library_version():
    """__graft_entry__.build() asserts the ABI version it was written against: keep it in step with svol_abi_version()."""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '__graft_entry__.py')).read()
    m = re.search(r'svol_abi_version\(\) == (\d+)', src)
    assert m, 'no ABI assertion in __graft_entry__.build()'
    from svol_amd import _lib
    assert int(m.group(1)) == _lib.lib().svol_abi_version()
