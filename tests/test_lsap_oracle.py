"""oracle/lsap.c (+ the pure-Python twin) against scipy's known answers
(tests/golden/lsap_known_answers.npz, scipy 1.15.3) and against live scipy."""
import numpy as np
import pytest

from oracle import lsap
from tests.helpers import load_golden


def _cases():
    z, _ = load_golden('lsap_known_answers')
    return [(str(z[f'c{i}/kind']), z[f'c{i}/cost'], z[f'c{i}/rows'], z[f'c{i}/cols']) for i in range(int(z['n']))]


def test_known_answers_c():
    for kind, c, r, cc in _cases():
        rr, rc = lsap.linear_sum_assignment(c)
        assert rr.tolist() == r.tolist() and rc.tolist() == cc.tolist(), (kind, c.shape)


def test_known_answers_py_small():
    for kind, c, r, cc in _cases():
        if c.size > 700:
            continue
        rr, rc = lsap.linear_sum_assignment_py(c)
        assert rr.tolist() == r.tolist() and rc.tolist() == cc.tolist(), (kind, c.shape)


def test_survey_vectors():
    # SURVEY.md §8c known-answer vectors (probed from scipy 1.15.3)
    f = lsap.linear_sum_assignment
    assert [x.tolist() for x in f(np.zeros((4, 2)))] == [[0, 1], [0, 1]]
    assert [x.tolist() for x in f(np.zeros((2, 4)))] == [[0, 1], [0, 1]]
    assert [x.tolist() for x in f(np.zeros((3, 3)))] == [[0, 1, 2], [0, 1, 2]]
    r, c = f(np.zeros((10, 0)))
    assert r.dtype == np.int64 and len(r) == 0 and len(c) == 0
    assert [x.tolist() for x in f(np.array([[1, 2], [1, 2], [0, 2], [1, 0], [1, 0]]))] == [[2, 3], [0, 1]]
    assert [x.tolist() for x in f(np.array([[.3, .1], [.1, .3], [.2, .2]], np.float32))] == [[0, 1], [1, 0]]
    assert [x.tolist() for x in f(np.random.default_rng(0).random((10, 3)).astype(np.float32))] == \
        [[1, 3, 4], [0, 2, 1]]
    with pytest.raises(ValueError):
        f(np.array([[np.nan, 1.0]]))
    with pytest.raises(ValueError):
        f(np.array([[-np.inf, 1.0]]))
    f(np.array([[np.inf, 1.0], [1.0, np.inf]]))  # +inf accepted


def test_live_scipy_random():
    scipy_opt = pytest.importorskip('scipy.optimize')
    rng = np.random.RandomState(123)
    for _ in range(200):
        nr, nc = rng.randint(1, 40, size=2)
        c = rng.randint(0, 4, size=(nr, nc)).astype(np.float64) if rng.rand() < 0.5 else rng.standard_normal((nr, nc))
        r0, c0 = scipy_opt.linear_sum_assignment(c)
        r1, c1 = lsap.linear_sum_assignment(c)
        assert r0.tolist() == r1.tolist() and c0.tolist() == c1.tolist()


def test_lsap_c_restatement_under_asan_and_ubsan():
    """SURVEY §5 (sanitizers): the oracle's C restatement of scipy's LSAP built with AddressSanitizer + UBSan (`make -C oracle asan`,
    CPU only — GPU sanitizers are not available on this pool) and run over the same known-answer and random problems in a child
    process with the sanitizer runtime preloaded; any heap error or undefined behaviour aborts the child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    odir = os.path.join(root, 'oracle')
    r = subprocess.run(['make', '-C', odir, 'asan'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(odir, 'liboracle_lsap_asan.so')
    asan_rt = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan_rt) or not os.path.exists(asan_rt):
        pytest.skip('libasan runtime not found')
    code = r"""
import ctypes, numpy as np, sys
L = ctypes.CDLL(sys.argv[1])
rng = np.random.default_rng(0)
n_ok = 0
def solve(c):
    nr, nc = c.shape
    k = min(nr, nc)
    rows = np.full(max(k, 1), -1, np.int64); cols = np.full(max(k, 1), -1, np.int64)
    if c.dtype == np.float32:
        f, p = L.svol_oracle_lsap_f32, ctypes.POINTER(ctypes.c_float)
    else:
        f, p = L.svol_oracle_lsap_f64, ctypes.POINTER(ctypes.c_double)
    f.restype = ctypes.c_int64
    f.argtypes = [p, ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    rc = f(c.ctypes.data_as(p), nr, nc, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), cols.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return rc, rows[:k], cols[:k]
shapes = [(4, 2), (2, 4), (3, 3), (10, 0), (0, 5), (1, 1), (100, 37), (37, 100), (64, 64)] + [(int(a), int(b)) for a, b in rng.integers(1, 40, (60, 2))]
for i, (nr, nc) in enumerate(shapes):
    c = np.ascontiguousarray(rng.random((nr, nc)), dtype=np.float32 if i % 3 == 0 else np.float64)
    if nr * nc and rng.random() < 0.3:
        c = (np.round(c * 3) / 3).astype(c.dtype)          # heavy ties
    rc, rows, cols = solve(c)
    k = min(nr, nc)
    assert rc == (k if nr * nc else 0), (nr, nc, rc)
    assert len(set(cols.tolist())) == k and len(set(rows.tolist())) == k
    n_ok += 1
assert solve(np.array([[1.0, float('nan')], [0.0, 1.0]]))[0] == -1
assert solve(np.array([[1.0, float('-inf')], [0.0, 1.0]]))[0] == -1
assert solve(np.array([[float('inf'), float('inf')], [float('inf'), float('inf')]]))[0] in (-2, 2)
print('asan ok', n_ok)
"""
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1')
    p = subprocess.run([sys.executable, '-c', code, lib], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'asan ok' in p.stdout, p.stdout[-1000:] + p.stderr[-3000:]
