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
    assert L.svol_abi_version() == 5
    assert b'invalid' in L.svol_strerror(-1)


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
