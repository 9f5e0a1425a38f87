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
