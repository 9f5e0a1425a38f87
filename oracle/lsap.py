"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

ctypes binding of ``oracle/lsap.c`` plus a pure-Python transcription of the same
algorithm for tiny cases.  Restates ``scipy.optimize.linear_sum_assignment`` as
called by the reference at lib/modeling/matcher.py:93,158.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile oracle/lsap.c with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, 'liboracle_lsap.so')
    src = os.path.join(_HERE, 'lsap.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, 'liboracle_lsap.so'])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle_lsap.so')
        if not os.path.exists(so):
            so = build()
        L = ctypes.CDLL(so)
        for name, ct in (('svol_oracle_lsap_f64', ctypes.c_double), ('svol_oracle_lsap_f32', ctypes.c_float)):
            f = getattr(L, name)
            f.restype = ctypes.c_int64
            f.argtypes = [ctypes.POINTER(ct), ctypes.c_int64, ctypes.c_int64,
                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        _LIB = L
    return _LIB


def linear_sum_assignment(cost):
    """Same contract as scipy's: returns (row_ind, col_ind) int64, rows ascending;
    raises ValueError on NaN / -inf entries or an infeasible matrix."""
    c = np.asarray(cost)
    if c.ndim != 2:
        raise ValueError('expected a matrix (2-D array), got a %r array' % (c.shape,))
    nr, nc = c.shape
    n = min(nr, nc)
    rows = np.zeros(n, np.int64)
    cols = np.zeros(n, np.int64)
    if n == 0:
        return rows, cols
    L = _lib()
    if c.dtype == np.float32:
        c = np.ascontiguousarray(c)
        r = L.svol_oracle_lsap_f32(c.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), nr, nc,
                                   rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                   cols.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    else:
        c = np.ascontiguousarray(c, dtype=np.float64)
        r = L.svol_oracle_lsap_f64(c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), nr, nc,
                                   rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                   cols.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    if r == -1:
        raise ValueError('matrix contains invalid numeric entries')
    if r == -2:
        raise ValueError('cost matrix is infeasible')
    return rows, cols


def linear_sum_assignment_py(cost):
    """Pure-Python loops (small cases only) — independent second statement of
    the same algorithm, used to cross-check lsap.c."""
    c = [[float(x) for x in row] for row in np.asarray(cost, dtype=np.float64)]
    nr = len(c)
    nc = len(c[0]) if nr else np.asarray(cost).shape[1]
    if nr == 0 or nc == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    transpose = nc < nr
    if transpose:
        c = [[c[i][j] for i in range(nr)] for j in range(nc)]
        nr, nc = nc, nr
    for row in c:
        for x in row:
            if x != x or x == -math.inf:
                raise ValueError('matrix contains invalid numeric entries')
    u = [0.0] * nr
    v = [0.0] * nc
    path = [-1] * nc
    col4row = [-1] * nr
    row4col = [-1] * nc
    for cur in range(nr):
        min_val = 0.0
        remaining = [nc - it - 1 for it in range(nc)]
        num_remaining = nc
        SR = [False] * nr
        SC = [False] * nc
        spc = [math.inf] * nc
        sink = -1
        i = cur
        while sink == -1:
            index = -1
            lowest = math.inf
            SR[i] = True
            for it in range(num_remaining):
                j = remaining[it]
                r = min_val + c[i][j] - u[i] - v[j]
                if r < spc[j]:
                    path[j] = i
                    spc[j] = r
                if spc[j] < lowest or (spc[j] == lowest and row4col[j] == -1):
                    lowest = spc[j]
                    index = it
            min_val = lowest
            if min_val == math.inf:
                raise ValueError('cost matrix is infeasible')
            j = remaining[index]
            if row4col[j] == -1:
                sink = j
            else:
                i = row4col[j]
            SC[j] = True
            num_remaining -= 1
            remaining[index] = remaining[num_remaining]
        u[cur] += min_val
        for i in range(nr):
            if SR[i] and i != cur:
                u[i] += min_val - spc[col4row[i]]
        for j in range(nc):
            if SC[j]:
                v[j] -= min_val - spc[j]
        j = sink
        while True:
            i = path[j]
            row4col[j] = i
            col4row[i], j = j, col4row[i]
            if i == cur:
                break
    if transpose:
        order = sorted(range(nr), key=lambda k: col4row[k])
        return (np.asarray([col4row[k] for k in order], np.int64), np.asarray(order, np.int64))
    return np.arange(nr, dtype=np.int64), np.asarray(col4row, np.int64)
