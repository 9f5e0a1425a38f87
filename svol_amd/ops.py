"""Host-side operators of the SVOL hot path: thin tensor wrappers over the C-ABI
(libsvol_hip.so) and the ``torch.autograd.Function`` s that compose them.

PyTorch here is plumbing only: device memory, the current HIP stream and the
autograd tape.  Every arithmetic op on the hot path is a hand-written gfx950
kernel behind ``include/svol_hip.h``; there is no CPU or eager fallback — a
non-CUDA tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_RELU_RES, ACT_GELU_D = 0, 1, 2, 3, 4, 5
_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}   # SVOL_F32 / SVOL_BF16 / SVOL_F16


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise _lib.SvolHipError(f'unsupported dtype {t.dtype}')


# the raw handle of torch's CURRENT stream.  torch.cuda.current_stream() costs 5-8 us of Python per call (device-index
# resolution through is_available / os.getenv) and every C-ABI call needs it: ~3 ms of host time per training step; the two
# private C entry points below are what it wraps (0.3 us).
_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_CUR_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream() -> int:
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _RAW_STREAM(_CUR_DEVICE())
    return torch.cuda.current_stream().cuda_stream


_STREAM_OBJS = {}


def _current_stream_obj():
    """torch.cuda.current_stream() through a cache keyed by the raw handle (same reason as _stream)."""
    if _RAW_STREAM is None or _CUR_DEVICE is None:
        return torch.cuda.current_stream()
    key = (_CUR_DEVICE(), _RAW_STREAM(_CUR_DEVICE()))
    obj = _STREAM_OBJS.get(key)
    if obj is None:
        obj = _STREAM_OBJS[key] = torch.cuda.current_stream()
    return obj


def _ptr(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise _lib.SvolHipError('svol_amd ops need device tensors (no CPU fallback)')
    return t.data_ptr()


class KernelTimer:
    """Optional per-kernel timing with HIP events recorded on the launch stream (the stream every
    C-ABI call is enqueued on = torch's current stream).  Disabled by default (zero overhead);
    bench.py enables it for the dominant kernels to report the live roofline numbers."""

    def __init__(self):
        self.enabled = False
        self.names = set()
        self.records = {}

    def enable(self, names):
        self.enabled, self.names, self.records = True, set(names), {}

    def disable(self):
        self.enabled = False

    def clear(self):
        self.records = {}

    def start(self, name, meta=None):
        if not self.enabled or name not in self.names:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, meta, e0, e1)

    def stop(self, tok):
        if tok is None:
            return
        name, meta, e0, e1 = tok
        e1.record()
        self.records.setdefault((name, meta), []).append((e0, e1))

    def summary(self):
        """{(name, meta): (launches, mean_ms)} — call after torch.cuda.synchronize()."""
        out = {}
        for k, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[k] = (len(ms), sum(ms) / len(ms))
        return out


timer = KernelTimer()


# ----------------------------------------------------------------------------
# raw wrappers (no autograd)
# ----------------------------------------------------------------------------
def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.check(_lib.lib().svol_cast(_ptr(x), _dt(x), _ptr(out), _DT[dtype], x.numel(), _stream()), 'svol_cast')
    return out


def cast_transpose(w: torch.Tensor, dtype: torch.dtype, want: bool = True, want_t: bool = True):
    """fp32 [R,C] -> (dtype [R,C] or None, dtype [C,R] or None)."""
    assert w.dtype == torch.float32 and w.dim() == 2
    w = w.contiguous()
    R, C = w.shape
    d = torch.empty((R, C), dtype=dtype, device=w.device) if want else None
    dT = torch.empty((C, R), dtype=dtype, device=w.device) if want_t else None
    _lib.check(_lib.lib().svol_cast_transpose(_ptr(w), _ptr(d), _ptr(dT), _DT[dtype], R, C, _stream()),
               'svol_cast_transpose')
    return d, dT


def gemm_nt(A, B, bias=None, act=ACT_NONE, residual=None, want_pre=False, out=None, out_f32=False, colscale=None, pre_out=None):
    """out[M,N] = act(A[M,K] @ B[N,K]^T + bias) + residual.  A/B/out/residual may be column slices
    of wider row-major buffers (stride(0) is the leading dimension).  out_f32: `out` and `residual`
    are fp32 (the fp32 residual stream) whatever the operand dtype."""
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    M, K = A.shape
    N = B.shape[0]
    cdt = torch.float32 if out_f32 else A.dtype
    if out is None:
        out = torch.empty((M, N), dtype=cdt, device=A.device)
    assert out.dtype == cdt and (residual is None or residual.dtype == cdt)
    pre = (pre_out if pre_out is not None else torch.empty((M, N), dtype=A.dtype, device=A.device)) if want_pre else None
    rc = _lib.lib().svol_gemm_nt(_ptr(A), A.stride(0), 0, 0, _ptr(B), B.stride(0), _ptr(out), out.stride(0),
                                 _ptr(bias), _ptr(colscale), act, _ptr(pre), pre.stride(0) if pre is not None else 0,
                                 _ptr(residual),
                                 residual.stride(0) if residual is not None else 0, 1 if out_f32 else 0, M, N, K,
                                 _dt(A), _stream())
    _lib.check(rc, 'svol_gemm_nt')
    return (out, pre) if want_pre else out


def gemm_nt_split(A, W_hilo, bias=None, out=None):
    """out[M,N] (bf16) = A[M,K] @ (W_hi + W_lo)[N,K]^T + bias with W_hilo [N, 2K] = [hi | lo] (weights.get_split): one
    K-concatenated product, [A | A] is never materialised."""
    assert A.dim() == 2 and W_hilo.dim() == 2 and A.stride(1) == 1 and W_hilo.stride(1) == 1 and A.dtype == torch.bfloat16
    M, K = A.shape
    N = W_hilo.shape[0]
    assert W_hilo.shape[1] == 2 * K
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    rc = _lib.lib().svol_gemm_nt_split(_ptr(A), A.stride(0), _ptr(W_hilo), W_hilo.stride(0), _ptr(out), out.stride(0), _ptr(bias),
                                       M, N, K, _stream())
    _lib.check(rc, 'svol_gemm_nt_split')
    return out


def gemm_tn(A, B, out=None, colsum=None, stream=None):
    """out[N,K] (fp32) += A[Mc,N]^T @ B[Mc,K]; a fresh zeroed `out` is allocated when not given.
    colsum (fp32 [N], zeroed by the caller) += column sums of A (bias gradient, fused).  stream: raw handle (default: current)."""
    assert A.dim() == 2 and B.dim() == 2 and A.shape[0] == B.shape[0]
    assert A.stride(1) == 1 and B.stride(1) == 1
    Mc, N = A.shape
    K = B.shape[1]
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=A.device)
    rc = _lib.lib().svol_gemm_tn(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), out.stride(0), _ptr(colsum),
                                 Mc, N, K, _dt(A), _stream() if stream is None else stream)
    _lib.check(rc, 'svol_gemm_tn')
    return out


def gemm_tn_grouped(problems, stream=None):
    """several gemm_tn problems of one element type in one launch: problems = [(A, B, out, colsum | None), ...], out / colsum fp32,
    accumulated into (svol_gemm_tn_grouped)."""
    n = len(problems)
    arr = (_lib.TnProblem * n)()
    dt = None
    for i, (A, B, out, cs) in enumerate(problems):
        assert A.dim() == 2 and B.dim() == 2 and A.shape[0] == B.shape[0] and A.stride(1) == 1 and B.stride(1) == 1 and A.dtype == B.dtype
        assert dt is None or dt == A.dtype
        dt = A.dtype
        arr[i] = _lib.TnProblem(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), out.stride(0), _ptr(cs), A.shape[0], A.shape[1],
                                B.shape[1])
    rc = _lib.lib().svol_gemm_tn_grouped(ctypes.cast(arr, ctypes.c_void_p), n, _DT[dt], _stream() if stream is None else stream)
    _lib.check(rc, 'svol_gemm_tn_grouped')


# ---- weight gradients off the critical path --------------------------------------------------------------------------------
# dW = dY^T X is needed by nobody until the optimizer step (or the bucket's all-reduce), while the dX chain behind it IS the
# critical path of backward.  When the gradient lands in a reducer-owned bucket (a "sink": persistent memory, no autograd
# consumer), the TN GEMM is issued on a dedicated stream that waits for the producing stream's work so far and is joined by
# BucketedGradAllReduce (before a bucket's all-reduce and in finish()).  The split-M TN kernels are HBM / atomic bound, the
# attention-backward kernels they now run beside are instruction-issue bound: the two overlap almost for free.
WGRAD_ASYNC = os.environ.get('SVOL_NO_WGRAD_STREAM') is None
WGRAD_IN_CAPTURE = False   # (svol_amd.graph: a captured step keeps everything on one stream unless told otherwise)
_WGRAD = {}


def wgrad_streams(dev):
    """streams weight-gradient GEMMs have been issued on for `dev` (svol_amd.parallel joins them)."""
    s = _WGRAD.get(torch.device(dev) if not isinstance(dev, torch.device) else dev)
    return [s] if s is not None else []


# Deferral: the weight-gradient GEMMs are HBM / atomic bound; the video attention backward — 1.3 ms per layer — is instruction-issue
# bound and is the part of the backward they disturb least.  So a gradient-sink TN GEMM is only QUEUED here (with an event that says
# "its operands exist") and the queue is flushed onto the weight-gradient stream right before the next large attention backward is
# launched, before a bucket's all-reduce, and in BucketedGradAllReduce.finish() (same-box A/B: -0.2 ms per step).  The autograd engine
# runs every backward node of a device on one thread, so the queue needs no lock.
WGRAD_DEFER = os.environ.get('SVOL_NO_WGRAD_DEFER') is None
_WGRAD_PENDING = []
_WGRAD_FLUSH_MIN_SCORES = 1 << 27   # B*H*Lq*Lk of an attention backward worth hiding weight gradients under (cfg2: 2.5e9)
_BIG_ATTN = {'left': 0}             # large attention forwards of this step whose backward has not run yet: nothing is deferred once
                                    # it reaches 0 (after the first layer's attention backward there is nothing left to hide under)


def _wgrad_stream(dev):
    ws = _WGRAD.get(dev)
    if ws is None:
        ws = _WGRAD[dev] = torch.cuda.Stream(device=dev)
    return ws


# Round 3 gated the queued weight-gradient GEMMs to start beside the (two-pass, 256-VGPR, two workgroups per CU) attention backward.
# Round 4's single-pass backward owns a CU's whole register file (one 512-register wave per SIMD): nothing can run beside it, a gated
# GEMM only takes CUs away from it between its workgroup rounds — same box, same build: 19.72 ms/step gated (attention backward 1.31 ms
# in the step against 1.04 alone), 19.45 ungated (1.04 in the step).  Off by default; SVOL_WGRAD_GATE=1 restores it.
WGRAD_GATE = os.environ.get('SVOL_WGRAD_GATE') is not None and os.environ.get('SVOL_NO_WGRAD_GATE') is None


def flush_wgrad(gate=False):
    """issue every queued weight-gradient launch (callables that put it on the weight-gradient stream behind the event recorded when
    it was queued).  ``gate=True`` (the caller is about to launch a large attention backward on the current stream): the
    weight-gradient stream additionally waits for THIS point of the current stream.  Deferring the launches on the host alone orders
    nothing on the device once the host runs ahead of it (block programs: a step is issued in ~6 ms): the queued GEMMs then start the
    moment their operands exist, i.e. beside the next layer's LayerNorm / dgelu kernels, which are HBM-bound like they are
    (tools/block_trace.py timeline: LN3' 182 us against 47 alone), instead of beside the issue-bound attention backward."""
    if not _WGRAD_PENDING:
        return
    items = list(_WGRAD_PENDING)
    _WGRAD_PENDING.clear()
    if gate and WGRAD_GATE and not torch.cuda.is_current_stream_capturing():
        cur = _current_stream_obj()
        ev = torch.cuda.Event()
        ev.record(cur)
        _wgrad_stream(cur.device).wait_event(ev)
    for it in items:
        it()


_FLUSH_ARMED = [False]


def _final_flush():
    _FLUSH_ARMED[0] = False
    flush_wgrad()


def arm_wgrad_flush():
    """called when something is queued during a backward: make the autograd engine drain the queue before ``backward()`` returns, so
    a caller that never reaches ``BucketedGradAllReduce.finish()`` (an exception, a skipped batch) cannot leak a failed step's
    weight-gradient GEMMs into the next step's freshly zeroed buckets."""
    if _FLUSH_ARMED[0]:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_final_flush)
        _FLUSH_ARMED[0] = True
    except RuntimeError:   # not inside a backward pass: finish() / the next flush point drains the queue
        pass


def drop_pending_wgrad():
    """a new step starts: whatever a FAILED backward left queued must not run into this step's buffers."""
    _WGRAD_PENDING.clear()
    _FLUSH_ARMED[0] = False


def gemm_tn_sink(A, B, out, colsum=None):
    """gemm_tn into a persistent gradient bucket view, off the critical path (see above).  `out` / `colsum` must be sink
    views: nothing on the current stream may read them before BucketedGradAllReduce.finish()."""
    if not WGRAD_ASYNC or (not WGRAD_IN_CAPTURE and torch.cuda.is_current_stream_capturing()):
        return gemm_tn(A, B, out=out, colsum=colsum)
    cur = _current_stream_obj()
    ev = torch.cuda.Event()
    ev.record(cur)

    def later():
        ws = _wgrad_stream(A.device)
        ws.wait_event(ev)
        gemm_tn(A, B, out=out, colsum=colsum, stream=ws.cuda_stream)   # (explicit handle: no current-stream switch on the host)
        A.record_stream(ws)
        B.record_stream(ws)

    if WGRAD_DEFER and _BIG_ATTN['left'] > 0 and not torch.cuda.is_current_stream_capturing():
        _WGRAD_PENDING.append(later)
        arm_wgrad_flush()
    else:
        later()
    return out


def gemm_nt_dact(A, B, aux, act, want_colsum=True, colsum_out=None):
    """(A @ B^T) * act'(aux), and its column sums (fp32, accumulated into colsum_out when given) — fused MLP
    backward step.  act = ACT_GELU (aux = saved pre-activation) or ACT_RELU (aux = saved post-activation)."""
    M, K = A.shape
    N = B.shape[0]
    out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    cs = colsum_out
    if cs is None and want_colsum:
        cs = torch.zeros((N,), dtype=torch.float32, device=A.device)
    rc = _lib.lib().svol_gemm_nt_dact(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), out.stride(0), _ptr(aux),
                                      aux.stride(0), act, _ptr(cs), M, N, K, _dt(A), _stream())
    _lib.check(rc, 'svol_gemm_nt_dact')
    return out, cs


def gemm_nt_dgelu(A, B, pre, want_colsum=True, colsum_out=None):
    return gemm_nt_dact(A, B, pre, ACT_GELU, want_colsum, colsum_out)


def colsum(X, out=None):
    assert X.dim() == 2 and X.stride(1) == 1
    M, N = X.shape
    if out is None:
        out = torch.zeros((N,), dtype=torch.float32, device=X.device)
    _lib.check(_lib.lib().svol_colsum(_ptr(X), X.stride(0), _ptr(out), M, N, _dt(X), _stream()), 'svol_colsum')
    return out


def act_bwd(dy, aux, act):
    dy = dy.contiguous()
    aux = aux.contiguous()
    out = torch.empty_like(dy)
    _lib.check(_lib.lib().svol_act_bwd(_ptr(dy), _ptr(aux), _ptr(out), act, dy.numel(), _dt(dy), _stream()),
               'svol_act_bwd')
    return out


def layernorm_fwd(x, gamma, beta, dtype, pos=None, p=0.0, seed=0, want32=False, want_t=True, seed_dev=None):
    """x [M,D] (fp32 residual stream or `dtype`) -> (y32 | None, y | None, ypos | None, mean, rstd)."""
    x = x.contiguous()
    M, D = x.shape
    x_f32 = 1 if (x.dtype == torch.float32 and dtype != torch.float32) else 0
    assert x.dtype in (dtype, torch.float32)
    y32 = torch.empty((M, D), dtype=torch.float32, device=x.device) if want32 else None
    y = torch.empty((M, D), dtype=dtype, device=x.device) if want_t else None
    ypos = torch.empty((M, D), dtype=dtype, device=x.device) if pos is not None else None
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    if pos is not None:
        pos = pos.contiguous()
        assert pos.dtype == dtype and pos.shape[-1] == D
    rc = _lib.lib().svol_layernorm_fwd(_ptr(x), x_f32, _ptr(gamma), _ptr(beta), _ptr(y32), _ptr(y), _ptr(ypos),
                                       _ptr(pos), pos.numel() // D if pos is not None else 0, _ptr(mean), _ptr(rstd),
                                       M, D, float(p), int(seed), _ptr(seed_dev), _DT[dtype], _stream())
    _lib.check(rc, 'svol_layernorm_fwd')
    return y32, y, ypos, mean, rstd


def layernorm_bwd(dy32, dy, dy2, x, gamma, mean, rstd, dtype, p=0.0, seed=0, want32=False, want_t=True,
                  want_colsum=False, seed_dev=None, dg_out=None, db_out=None, cs_out=None):
    """LN backward; any of dy32 (fp32) / dy / dy2 (`dtype`) may be None.  Returns (dx32|None, dx|None, dg, db)
    (+ the column sums of dx when want_colsum: the bias gradient of the Linear feeding this LN)."""
    M, D = x.shape
    x_f32 = 1 if (x.dtype == torch.float32 and dtype != torch.float32) else 0
    cont = lambda t: None if t is None else t.reshape(M, D).contiguous()
    dy32, dy, dy2 = cont(dy32), cont(dy), cont(dy2)
    assert dy32 is None or dy32.dtype == torch.float32
    assert (dy is None or dy.dtype == dtype) and (dy2 is None or dy2.dtype == dtype)
    dx32 = torch.empty((M, D), dtype=torch.float32, device=x.device) if want32 else None
    dx = torch.empty((M, D), dtype=dtype, device=x.device) if want_t else None
    dg, db, cs = dg_out, db_out, (cs_out if want_colsum else None)  # accumulated into (gradient sinks) when given
    if dg is None or db is None or (want_colsum and cs is None):
        red = torch.zeros((3, D), dtype=torch.float32, device=x.device)  # one memset for dgamma | dbeta | colsum(dx)
        dg = red[0] if dg is None else dg
        db = red[1] if db is None else db
        cs = (red[2] if cs is None else cs) if want_colsum else None
    rc = _lib.lib().svol_layernorm_bwd(_ptr(dy32), _ptr(dy), _ptr(dy2), _ptr(x), x_f32, _ptr(gamma), _ptr(mean),
                                       _ptr(rstd), _ptr(dx32), _ptr(dx), _ptr(dg), _ptr(db), _ptr(cs), M, D, float(p),
                                       int(seed), _ptr(seed_dev), _DT[dtype], _stream())
    _lib.check(rc, 'svol_layernorm_bwd')
    if want_colsum:
        return dx32, dx, dg, db, cs
    return dx32, dx, dg, db


def posenc_sine(mask_f32: torch.Tensor, D: int, dtype: torch.dtype) -> torch.Tensor:
    mask_f32 = mask_f32.contiguous()
    B, L = mask_f32.shape
    pos = torch.empty((B, L, D), dtype=dtype, device=mask_f32.device)
    _lib.check(_lib.lib().svol_posenc_sine(_ptr(mask_f32), _ptr(pos), B, L, D, _DT[dtype], _stream()),
               'svol_posenc_sine')
    return pos


def _attn_ws(q, B, H, Lq, Lk, dh, masked):
    """scratch of the attention launches: partial results of the key-split (few queries, many keys), the key-tile classes of
    the masked fast kernels, or the per-workgroup redo flags of the fast unmasked forward; None when the library needs none."""
    n = _lib.lib().svol_attn_ws_bytes(B, H, Lq, Lk, dh) if q.dtype != torch.float32 else 0
    if n <= 0:
        return None
    return torch.empty((n // 4,), dtype=torch.float32, device=q.device)


def dropout(x, p, seed, out=None):
    """x * keep(seed, row, column) (0 or 1/(1-p)) over x viewed as [-1, x.shape[-1]]; out may be x (in place).  x contiguous."""
    assert x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().svol_dropout(_ptr(x), _ptr(out), x.numel(), max(1, x.shape[-1] if x.dim() else 1), float(p), int(seed), _dt(x),
                                       _stream()), 'svol_dropout')
    return out


def dropout_add(t32, res32, p, seed):
    """res32 + dropout(t32), fp32, written over t32 (rows = t32.shape[-1] wide)."""
    assert t32.is_contiguous() and res32.is_contiguous() and t32.dtype == torch.float32 and res32.dtype == torch.float32
    _lib.check(_lib.lib().svol_dropout_add(_ptr(t32), _ptr(res32), _ptr(t32), t32.numel(), max(1, t32.shape[-1]), float(p), int(seed),
                                           _stream()), 'svol_dropout_add')
    return t32


def attn_fwd(q, k, v, B, H, Lq, Lk, dh, kbias=None, premul=0.0, drop=None):
    """q/k/v: 2-D [B*L, >=H*dh] views (column slices allowed). Returns o [B*Lq, H*dh], lse2 [B,H,Lq].
    drop = (p, seed): attention-probability dropout (svol_attn_fwd_dropout)."""
    o = torch.empty((B * Lq, H * dh), dtype=q.dtype, device=q.device)
    lse2 = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    ws = _attn_ws(q, B, H, Lq, Lk, dh, kbias is not None or Lk % 128 != 0)
    if B * H * Lq * Lk >= _WGRAD_FLUSH_MIN_SCORES:
        _BIG_ATTN['left'] += 1
    tok = timer.start('attn_fwd', (B, H, Lq, Lk, dh))
    if drop is not None and drop[0] > 0.0:
        rc = _lib.lib().svol_attn_fwd_dropout(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o),
                                              o.stride(0), _ptr(lse2), _ptr(kbias), B, H, Lq, Lk, dh, 1.0 / math.sqrt(dh),
                                              float(premul), _ptr(ws), ws.numel() * 4 if ws is not None else 0, float(drop[0]),
                                              int(drop[1]), _dt(q), _stream())
    else:
        rc = _lib.lib().svol_attn_fwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o),
                                      o.stride(0), _ptr(lse2), _ptr(kbias), B, H, Lq, Lk, dh, 1.0 / math.sqrt(dh),
                                      float(premul), _ptr(ws), ws.numel() * 4 if ws is not None else 0, _dt(q), _stream())
    timer.stop(tok)
    _lib.check(rc, 'svol_attn_fwd')
    return o, lse2


def attn_bwd(q, k, v, o, do, lse2, B, H, Lq, Lk, dh, dq, dk, dv, kbias=None, premul=0.0, drop=None):
    """Writes dq/dk/dv (2-D views, column slices allowed)."""
    do = do if do.stride(1) == 1 else do.contiguous()
    if B * H * Lq * Lk >= _WGRAD_FLUSH_MIN_SCORES:
        _BIG_ATTN['left'] -= 1
        flush_wgrad(gate=True)   # the queued weight-gradient GEMMs run beside this launch
    delta = torch.empty((3, B, H, Lq), dtype=torch.float32, device=q.device)  # delta | -lse2 pairs | -delta pairs (svol_hip.h)
    ws = _attn_ws(q, B, H, Lq, Lk, dh, kbias is not None or Lk % 128 != 0)
    tok = timer.start('attn_bwd', (B, H, Lq, Lk, dh))
    if drop is not None and drop[0] > 0.0:
        rc = _lib.lib().svol_attn_bwd_dropout(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o),
                                              o.stride(0), _ptr(do), do.stride(0), _ptr(lse2), _ptr(delta), _ptr(kbias),
                                              _ptr(dq), dq.stride(0), _ptr(dk), dk.stride(0), _ptr(dv), dv.stride(0), B, H, Lq,
                                              Lk, dh, 1.0 / math.sqrt(dh), float(premul), _ptr(ws),
                                              ws.numel() * 4 if ws is not None else 0, float(drop[0]), int(drop[1]), _dt(q),
                                              _stream())
    else:
        rc = _lib.lib().svol_attn_bwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o),
                                      o.stride(0), _ptr(do), do.stride(0), _ptr(lse2), _ptr(delta), _ptr(kbias),
                                      _ptr(dq), dq.stride(0), _ptr(dk), dk.stride(0), _ptr(dv), dv.stride(0), B, H, Lq,
                                      Lk, dh, 1.0 / math.sqrt(dh), float(premul), _ptr(ws),
                                      ws.numel() * 4 if ws is not None else 0, _dt(q), _stream())
    timer.stop(tok)
    _lib.check(rc, 'svol_attn_bwd')


def im2col(x, N, H, W, C, kh, kw, stride, pad, dtype, strides=None, ldcols=None):
    """x: image batch addressed by element `strides` (sn, sh, sw, sc) — default NHWC contiguous — -> (cols [N*Ho*Wo, ldcols]
    in `dtype`, Ho, Wo); column order (ky, kx, c), zero padding outside the image and in columns >= kh*kw*C."""
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    K = kh * kw * C
    ldcols = K if ldcols is None else ldcols
    sn, sh, sw, sc = strides if strides is not None else (H * W * C, W * C, C, 1)
    cols = torch.empty((N * Ho * Wo, ldcols), dtype=dtype, device=x.device)
    rc = _lib.lib().svol_im2col(_ptr(x), sn, sh, sw, sc, _dt(x), _ptr(cols), ldcols, N, H, W, C, kh, kw, stride, pad, _DT[dtype],
                                _stream())
    _lib.check(rc, 'svol_im2col')
    return cols, Ho, Wo


def conv_nhwc(x, w, bias, act, N, H, W, C, kh, kw, stride, pad, residual=None):
    """convolution of an NHWC activation [N*H*W, C] with folded weights w [Cout, kh*kw*C (+pad)] -> (y [N*Ho*Wo, Cout], Ho, Wo).
    The implicit-GEMM kernel when the shape fits it (bf16, C % 32 == 0), else im2col + gemm_nt."""
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    Cout = w.shape[0]
    if x.dtype == torch.bfloat16 and C % 32 == 0 and _CONV_IMPLICIT:
        y = torch.empty((N * Ho * Wo, Cout), dtype=x.dtype, device=x.device)
        rc = _lib.lib().svol_conv_nhwc(_ptr(x), _ptr(w), w.stride(0), _ptr(y), _ptr(bias), act, _ptr(residual), N, H, W, C, Cout, kh,
                                       kw, stride, pad, _dt(x), _stream())
        if rc == 0:
            return y, Ho, Wo
        if rc != -2:  # SVOL_E_UNSUPPORTED falls through to the explicit path
            _lib.check(rc, 'svol_conv_nhwc')
    cols, Ho, Wo = im2col(x, N, H, W, C, kh, kw, stride, pad, x.dtype, ldcols=w.shape[1])
    return gemm_nt(cols, w, bias, act, residual=residual), Ho, Wo


_CONV_IMPLICIT = os.environ.get('SVOL_CONV_IM2COL') is None


def maxpool_nhwc(x, N, H, W, C, k, stride, pad):
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((N * Ho * Wo, C), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().svol_maxpool_nhwc(_ptr(x), _ptr(y), N, H, W, C, k, stride, pad, _dt(x), _stream()), 'svol_maxpool_nhwc')
    return y, Ho, Wo


def avgpool_nhwc(x, N, HW, C):
    y = torch.empty((N, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().svol_avgpool_nhwc(_ptr(x), _ptr(y), N, HW, C, _dt(x), _stream()), 'svol_avgpool_nhwc')
    return y


# ---- the backbone in training mode (csrc/resnet_train.hip) --------------------------------------------------------------
BN_STATS_TWO_PASS = os.environ.get('SVOL_BN_TWO_PASS') is not None   # batch mean first, then the moment about it (tests: torch's own order)


def _bn_sync_world(sync):
    """ranks whose batch statistics are shared (apex convert_syncbn_model, train.py:65-68): the default process group, > 1 only."""
    if not sync:
        return 1
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def bn_train_stats(z, gamma, beta, running_mean, running_var, momentum, eps, sync=False):
    """train-mode nn.BatchNorm2d statistics of an NHWC activation z [M, C] (16-bit) in ONE pass over z: sums of (z - s) and (z - s)^2
    about a per-channel pivot s = the first row of z (a sample of the channel: the variance's subtraction loses a few bits, not the
    variance — a plain sum z^2 would at |mean| >> std; measured against fp64: mean 1e-7 of a standard deviation, rstd 1e-6), then
    everything [C]-sized at once (svol_bn_finalize): -> fp32 [C] rows (mean, rstd, scale = gamma * rstd, shift = beta - mean * scale);
    running_mean / running_var (may be None) are updated in place.  BN_STATS_TWO_PASS (SVOL_BN_TWO_PASS=1): the batch mean first, then
    the second moment about it — one more pass over z, statistics equal to torch's to the last bits (the end-to-end gradient test
    compares with a torch network whose ReLU masks flip with the 7th digit of a statistic).
    sync (``--sync_bn`` with more than one rank: apex SyncBatchNorm, train.py:65-68): the statistics are those of the GLOBAL batch —
    every rank sums about rank 0's pivot (one [C] broadcast), the [2, C] sums are all-reduced and divided by the global row count
    (every rank holds the same number of rows: the per-GPU batch of svol_dataloader.py:70)."""
    M, C = z.shape
    buf = torch.zeros((7, C), dtype=torch.float32, device=z.device)
    L_ = _lib.lib()
    dt, s = _dt(z), _stream()
    world = _bn_sync_world(sync)
    if world > 1:
        import torch.distributed as dist
    if BN_STATS_TWO_PASS:
        _lib.check(L_.svol_bn_colstats(_ptr(z), None, 0.0, _ptr(buf[0]), _ptr(buf[6]), M, C, dt, s), 'svol_bn_colstats')
        buf[6].zero_()
        if world > 1:
            dist.all_reduce(buf[0])
        _lib.check(L_.svol_bn_colstats(_ptr(z), _ptr(buf[0]), 1.0 / (M * world), _ptr(buf[6]), _ptr(buf[1]), M, C, dt, s), 'svol_bn_colstats')
        if world > 1:
            dist.all_reduce(buf[1])
        pivot = None
    else:
        buf[6].copy_(z[0])
        if world > 1:
            dist.broadcast(buf[6], src=0)
        _lib.check(L_.svol_bn_colstats(_ptr(z), _ptr(buf[6]), 1.0, _ptr(buf[0]), _ptr(buf[1]), M, C, dt, s), 'svol_bn_colstats')
        if world > 1:
            dist.all_reduce(buf[0:2])
        pivot = buf[6]
    _lib.check(L_.svol_bn_finalize(_ptr(buf[0]), _ptr(buf[1]), _ptr(pivot), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                   float(momentum), float(eps), M * world, C, _ptr(buf[2]), _ptr(buf[3]), _ptr(buf[4]), _ptr(buf[5]), s),
               'svol_bn_finalize')
    return buf[2], buf[3], buf[4], buf[5]


def conv_weight_pack(w, dtype, flip=False):
    """nn.Conv2d weight fp32 [Cout, Cin, kh, kw] -> the GEMMs' layout in `dtype`: [Cout, pad32(kh*kw*Cin)] in (ky, kx, c) order, or with
    flip the stride-1 data gradient's [Cin, pad32(kh*kw*Cout)] (taps reversed, channels swapped)."""
    Cout, Cin, kh, kw = w.shape
    rows, K = (Cin, kh * kw * Cout) if flip else (Cout, kh * kw * Cin)
    Kp = (K + 31) // 32 * 32
    out = torch.empty((rows, Kp), dtype=dtype, device=w.device)
    wc = w.detach()
    wc = wc if wc.is_contiguous() else wc.contiguous()
    _lib.check(_lib.lib().svol_conv_weight_pack(_ptr(wc), _ptr(out), Cout, Cin, kh, kw, Kp, 1 if flip else 0, _DT[dtype], _stream()),
               'svol_conv_weight_pack')
    return out


def conv_wgrad_nhwc(dz, x, N, H, W, C, kh, kw, stride, pad, Kp):
    """[Cout, Kp] fp32 = dz^T im2col(x) without the im2col matrix (svol_conv_wgrad_nhwc); None when the kernel does not take the shape."""
    Cout = dz.shape[1]
    dwp = torch.zeros((Cout, Kp), dtype=torch.float32, device=dz.device)
    rc = _lib.lib().svol_conv_wgrad_nhwc(_ptr(dz), _ptr(x), _ptr(dwp), N, H, W, C, Cout, kh, kw, stride, pad, Kp, _dt(dz), _stream())
    if rc == -2:
        return None
    _lib.check(rc, 'svol_conv_wgrad_nhwc')
    return dwp


def conv_weight_unpack_add(dwp, grad):
    """grad [Cout, Cin, kh, kw] (fp32, contiguous) += dwp [Cout, Kp] (fp32, (ky, kx, c) order)"""
    Cout, Cin, kh, kw = grad.shape
    assert grad.is_contiguous() and dwp.is_contiguous() and dwp.dtype == torch.float32 and grad.dtype == torch.float32
    _lib.check(_lib.lib().svol_conv_weight_unpack_add(_ptr(dwp), _ptr(grad), Cout, Cin, kh, kw, dwp.shape[1], _stream()),
               'svol_conv_weight_unpack_add')


def bn_apply(z, scale, shift, residual=None, relu=False):
    M, C = z.shape
    y = torch.empty_like(z)
    _lib.check(_lib.lib().svol_bn_apply(_ptr(z), _ptr(scale), _ptr(shift), _ptr(residual), 1 if relu else 0, _ptr(y), M, C, _dt(z),
                                        _stream()), 'svol_bn_apply')
    return y


def bn_bwd(dy, y, z, mean, rstd, gamma, want_dres, sync=False):
    """(dz, dres | None, dgamma, dbeta) of y = relu?(BN_train(z) + res): y = the saved output when a ReLU follows, else None.
    sync: the statistics were the global batch's (bn_train_stats(sync=True)) — dz needs the GLOBAL means of g and g * xhat (one
    [2, C] all-reduce); dgamma / dbeta stay this rank's sums, the gradient exchange averages them like every other parameter's."""
    M, C = z.shape
    sums = torch.zeros((2, C), dtype=torch.float32, device=z.device)
    L_ = _lib.lib()
    _lib.check(L_.svol_bn_bwd_reduce(_ptr(dy), _ptr(y), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(sums[0]), _ptr(sums[1]), M, C, _dt(z),
                                     _stream()), 'svol_bn_bwd_reduce')
    world = _bn_sync_world(sync)
    if world > 1:
        import torch.distributed as dist
        local = sums.clone()
        dist.all_reduce(sums)
        sums.mul_(1.0 / world)   # svol_bn_bwd_apply divides by ITS row count M: (sum over ranks / world) / M = global sum / (M * world)
        dz = torch.empty_like(z)
        dres = torch.empty_like(z) if want_dres else None
        _lib.check(L_.svol_bn_bwd_apply(_ptr(dy), _ptr(y), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(sums[0]), _ptr(sums[1]), _ptr(dz),
                                        _ptr(dres), M, C, _dt(z), _stream()), 'svol_bn_bwd_apply')
        return dz, dres, local[1], local[0]
    dz = torch.empty_like(z)
    dres = torch.empty_like(z) if want_dres else None
    _lib.check(L_.svol_bn_bwd_apply(_ptr(dy), _ptr(y), _ptr(z), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(sums[0]), _ptr(sums[1]), _ptr(dz),
                                    _ptr(dres), M, C, _dt(z), _stream()), 'svol_bn_bwd_apply')
    return dz, dres, sums[1], sums[0]


def col2im_nhwc(dcols, N, H, W, C, kh, kw, stride, pad):
    dx = torch.empty((N * H * W, C), dtype=dcols.dtype, device=dcols.device)
    _lib.check(_lib.lib().svol_col2im_nhwc(_ptr(dcols), dcols.stride(0), _ptr(dx), N, H, W, C, kh, kw, stride, pad, _dt(dcols), _stream()),
               'svol_col2im_nhwc')
    return dx


def maxpool_idx_nhwc(x, N, H, W, C, k, stride, pad):
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty((N * Ho * Wo, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((N * Ho * Wo, C), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().svol_maxpool_idx_nhwc(_ptr(x), _ptr(y), _ptr(idx), N, H, W, C, k, stride, pad, _dt(x), _stream()),
               'svol_maxpool_idx_nhwc')
    return y, idx, Ho, Wo


def maxpool_bwd_nhwc(dy, idx, N, H, W, C, k, stride, pad):
    dx = torch.empty((N * H * W, C), dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.lib().svol_maxpool_bwd_nhwc(_ptr(dy), _ptr(idx), _ptr(dx), N, H, W, C, k, stride, pad, _dt(dy), _stream()),
               'svol_maxpool_bwd_nhwc')
    return dx


def attn_weights_mean(q, k, lse2, B, H, Lq, Lk, dh, kbias=None, premul=0.0):
    """head-averaged softmax probabilities [B,Lq,Lk] fp32 (nn.MultiheadAttention's second return value), recomputed
    from q, k and the lse2 of attn_fwd."""
    att = torch.empty((B, Lq, Lk), dtype=torch.float32, device=q.device)
    rc = _lib.lib().svol_attn_weights_mean(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(lse2), _ptr(kbias), _ptr(att),
                                           B, H, Lq, Lk, dh, 1.0 / math.sqrt(dh), float(premul), _dt(q), _stream())
    _lib.check(rc, 'svol_attn_weights_mean')
    return att


# ----------------------------------------------------------------------------
# parameter cache: fp32 master weights -> compute-dtype copies (W and W^T), refreshed when the
# optimizer bumps the tensor version.  One cast per weight per step.
# ----------------------------------------------------------------------------
class _WeightCache:
    """fp32 master weight -> compute-dtype copies (W and W^T) in persistent buffers.

    Every weight is (re)cast once per EPOCH — the model advances the epoch at the start of every forward —
    because tensor version counters cannot be trusted to see optimizer updates (``torch.optim.AdamW(fused=True)``
    rewrites parameters without bumping ``_version`` on ROCm).  The first time a weight is seen it is cast on the
    spot and registered; from then on ``new_epoch()`` refreshes ALL registered weights of a (device, dtype) with
    ONE ``svol_cast_transpose_multi`` launch driven by a device-side descriptor table (64 launches -> 1 per step
    at the benchmark size).  With ``static=True`` (frozen weights, e.g. serving) nothing is refreshed."""

    def __init__(self):
        self.epoch = 0
        self.static = False
        self._sets = {}  # (device, dtype) -> {'items': [(weakref(param), entry)], 'table': tensor|None, 'tiles': int}

    class _Entry:
        __slots__ = ('wc', 'wt', 'ws', 'ptr', 'epoch', 'version', 'shape')

    def new_epoch(self):
        _BIG_ATTN['left'] = 0   # a new forward starts (see gemm_tn_sink)
        if not torch.is_grad_enabled() or not _FLUSH_ARMED[0]:
            drop_pending_wgrad()   # (an armed flush belongs to a backward that is still running: leave it alone)
        if self.static:
            return
        self.epoch += 1
        for key, st in self._sets.items():
            self._refresh(key, st)

    def _refresh(self, key, st):
        dev, dtype = key
        if dtype not in _DT:
            return
        def alive(r, e):
            w = r()
            if w is None:
                return False
            if e.ws is not None:   # split entry: keyed by (row0, rows), ptr = address of the first row
                return any(v is e for v in getattr(w, '_svol_split', {}).values())
            return getattr(w, '_svol_cache', {}).get(dtype) is e and w.data_ptr() == e.ptr
        live = [(r, e) for (r, e) in st['items'] if alive(r, e)]
        if len(live) != len(st['items']) or st['table'] is None:
            st['items'] = live
            st['table'] = None
            if not live:
                return
            recs = []
            t0 = 0
            for r, e in live:
                R, C = e.shape
                tc = (C + 31) // 32
                # src, dst, dstT, dstS (8 bytes each) ; R, C, tiles_c, tile_begin (int32)
                if e.ws is not None:   # split copy of a row range of the weight (get_split): its own record
                    recs.append((e.ptr, 0, 0, e.ws.data_ptr(), R, C, tc, t0))
                else:
                    recs.append((e.ptr, 0 if dtype == torch.float32 else e.wc.data_ptr(), e.wt.data_ptr(), 0, R, C, tc, t0))
                t0 += ((R + 31) // 32) * tc
            import struct
            blob = b''.join(struct.pack('<QQQQiiii', *rec) for rec in recs)
            st['table'] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
            st['tiles'] = t0
            st['gen'] = st.get('gen', 0) + 1
        _lib.check(_lib.lib().svol_cast_transpose_multi(_ptr(st['table']), len(st['items']), st['tiles'], _DT[dtype],
                                                        _stream()), 'svol_cast_transpose_multi')
        # The copies were allocated on the stream of their first use (the query half's weights: the side stream) and are
        # rewritten HERE, on the caller's stream.  Tell the allocator, or a copy whose owner dies (a model dropped between two
        # tests, a re-built module) goes back to its own stream's pool while this launch is still queued and the refresh lands
        # in whoever got the block next (seen as a 1-in-15 garbage forward in the test suite, never with one long-lived model)
        if dev.type == 'cuda':
            cur = _current_stream_obj()
            done = st.setdefault('recorded', set())
            key_ = (cur.cuda_stream, st.get('gen', 0))
            if key_ not in done:
                done.clear()
                done.add(key_)
                for r, e in st['items']:
                    if e.ws is not None:
                        e.ws.record_stream(cur)
                        continue
                    if e.wc is not None and e.wc.dtype != torch.float32:
                        e.wc.record_stream(cur)
                    e.wt.record_stream(cur)
        for r, e in st['items']:
            e.epoch = self.epoch
            e.version = r()._version

    def get(self, w: torch.Tensor, dtype: torch.dtype):
        cache = getattr(w, '_svol_cache', None)
        if cache is None:
            cache = {}
            try:
                w._svol_cache = cache
            except Exception:  # pragma: no cover  (non-leaf views etc.: just do not cache)
                pass
        e = cache.get(dtype)
        if e is not None and e.ptr == w.data_ptr() and e.version == w._version and (self.static or e.epoch == self.epoch):
            return e.wc, e.wt
        wd = w.detach()
        assert wd.dtype == torch.float32 and wd.dim() == 2 and wd.is_contiguous()
        fresh = e is None or e.ptr != w.data_ptr() or tuple(w.shape) != e.shape
        if fresh:
            e = self._Entry()
            e.ws = None
            e.shape, e.ptr = tuple(w.shape), w.data_ptr()
            R, C = e.shape
            e.wc = wd if dtype == torch.float32 else torch.empty((R, C), dtype=dtype, device=w.device)
            e.wt = torch.empty((C, R), dtype=dtype, device=w.device)
        R, C = e.shape
        _lib.check(_lib.lib().svol_cast_transpose(_ptr(wd), 0 if dtype == torch.float32 else _ptr(e.wc), _ptr(e.wt),
                                                  _DT[dtype], R, C, _stream()), 'svol_cast_transpose')
        e.epoch, e.version = self.epoch, w._version
        if fresh:
            cache[dtype] = e
            st = self._sets.setdefault((w.device, dtype), {'items': [], 'table': None, 'tiles': 0})
            try:
                import weakref
                st['items'].append((weakref.ref(w), e))
                st['table'] = None
            except TypeError:  # pragma: no cover
                pass
        return e.wc, e.wt

    def get_split(self, w: torch.Tensor, row0: int, rows: int):
        """bf16 split copy [rows, 2C] = [hi | lo] of rows [row0, row0 + rows) of the fp32 master weight `w` (svol_cast_split):
        the operand of gemm_nt_split.  Refreshed once per epoch by the same multi-weight launch as the plain copies."""
        cache = getattr(w, '_svol_split', None)
        if cache is None:
            cache = {}
            try:
                w._svol_split = cache
            except Exception:  # pragma: no cover
                pass
        C = w.shape[1]
        ptr = w.data_ptr() + row0 * C * 4
        e = cache.get((row0, rows))
        if e is not None and e.ptr == ptr and e.version == w._version and (self.static or e.epoch == self.epoch):
            return e.ws
        wd = w.detach()
        assert wd.dtype == torch.float32 and wd.dim() == 2 and wd.is_contiguous()
        fresh = e is None or e.ptr != ptr
        if fresh:
            e = self._Entry()
            e.wc = e.wt = None
            e.shape, e.ptr = (rows, C), ptr
            e.ws = torch.empty((rows, 2 * C), dtype=torch.bfloat16, device=w.device)
        _lib.check(_lib.lib().svol_cast_split(ptr, C, _ptr(e.ws), rows, C, _stream()), 'svol_cast_split')
        e.epoch, e.version = self.epoch, w._version
        if fresh:
            cache[(row0, rows)] = e
            st = self._sets.setdefault((w.device, torch.bfloat16), {'items': [], 'table': None, 'tiles': 0})
            try:
                import weakref
                st['items'].append((weakref.ref(w), e))
                st['table'] = None
            except TypeError:  # pragma: no cover
                pass
        return e.ws


weights = _WeightCache()


# ----------------------------------------------------------------------------
# gradient sinks: when svol_amd.parallel.BucketedGradAllReduce owns the gradients (param.grad is a view into
# a flat fp32 bucket that is zeroed once per step), the weight-/bias-/LayerNorm-gradient kernels accumulate
# STRAIGHT into that view and the Function returns None for the parameter — no per-parameter memset, no
# AccumulateGrad add.  The parameter's post-accumulate-grad hook (the reducer's bucket countdown) still fires:
# autograd runs the AccumulateGrad node with an undefined gradient after the producing Function has finished.
# ----------------------------------------------------------------------------
class GradSink:
    """view: the parameter's slice of a flat gradient bucket.  owner / bucket: who to tell that the gradient is complete when the
    producing Function keeps the parameter OUT of the autograd graph (svol_amd.blocks: no AccumulateGrad node, no hook) —
    ``owner.params_done(bucket, n)``."""
    __slots__ = ('view', 'owner', 'bucket')

    def __init__(self, view, owner=None, bucket=-1):
        self.view, self.owner, self.bucket = view, owner, bucket


def _claim(p, needed=True):
    """forward: the sink of parameter `p` if its gradient can be written in place, else None."""
    s = getattr(p, '_svol_sink', None) if (needed and p is not None) else None
    g = p.grad if s is not None else None
    if g is None or g.data_ptr() != s.view.data_ptr():
        return None
    return s


def _tgt(sink, shape, device):
    return sink.view if sink is not None else torch.zeros(shape, dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------
# autograd Functions
# ----------------------------------------------------------------------------
def _pos_grad(dypos, pos_shape, D):
    """gradient of a broadcast positional operand (the learnable query embedding): sum over the batch."""
    rows = 1
    for n in pos_shape[:-1]:
        rows *= n
    g = colsum(dypos.reshape(-1, rows * D).contiguous())
    return cast(g.view(pos_shape), dypos.dtype)


class LayerNormFn(torch.autograd.Function):
    """y = dropout(LN(x)), compute dtype in / out (input projections, svanet.py:168-178)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, p, seed, seed_dev):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        _, y, _, mean, rstd = layernorm_fwd(x2, gamma, beta, x.dtype, None, p, seed, seed_dev=seed_dev)
        ctx.save_for_backward(x2, gamma, mean, rstd, seed_dev)
        ctx.p, ctx.seed, ctx.shp = p, seed, shp
        ctx.sinks = (_claim(gamma, ctx.needs_input_grad[1]), _claim(beta, ctx.needs_input_grad[2]))
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, mean, rstd, seed_dev = ctx.saved_tensors
        sg, sb = ctx.sinks
        _, dx, dg, db = layernorm_bwd(None, dy, None, x2, gamma, mean, rstd, x2.dtype, ctx.p, ctx.seed, seed_dev=seed_dev,
                                      dg_out=sg.view if sg else None, db_out=sb.view if sb else None)
        return dx.view(ctx.shp), None if sg else dg, None if sb else db, None, None, None


def layer_norm(x, gamma, beta, p=0.0, seed=0, seed_dev=None):
    return LayerNormFn.apply(x, gamma, beta, p, seed, seed_dev)


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) (nn.Linear + F.relu / sigmoid).  out_f32: y is emitted in fp32 (it starts the
    fp32 residual stream); its gradient then arrives in fp32 and is cast once."""

    @staticmethod
    def forward(ctx, x, W, b, act, out_f32):
        if act == ACT_GELU:
            raise _lib.SvolHipError('LinearFn: GELU is only available fused in MLPLNFn')
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        Wc, WcT = weights.get(W, x.dtype)
        y = gemm_nt(x2, Wc, b, act, out_f32=out_f32)
        ctx.save_for_backward(x2, y if act != ACT_NONE else None)
        ctx.WcT, ctx.act, ctx.shp, ctx.has_b = WcT, act, shp, b is not None
        ctx.need_dx = ctx.needs_input_grad[0]
        epc = 4 if x.dtype == torch.float32 else 8
        ok = W.shape[0] % epc == 0
        ctx.sinks = (_claim(W, ok and ctx.needs_input_grad[1]), _claim(b, ok and ctx.needs_input_grad[2]))
        return y.view(*shp[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, y = ctx.saved_tensors
        N = dy.shape[-1]
        d = dy.reshape(-1, N)
        if not d.is_contiguous():
            d = d.contiguous()
        if d.dtype != x2.dtype:
            d = cast(d, x2.dtype)
        if ctx.act != ACT_NONE:
            d = act_bwd(d, y, ctx.act)
        # out-features that are not a multiple of the 16-byte chunk (2-class / 4-coordinate heads):
        # zero-pad the contraction / column dimension
        epc = 4 if d.dtype == torch.float32 else 8
        WcT = ctx.WcT
        Np = (N + epc - 1) // epc * epc
        if N % epc:
            dp = torch.zeros((d.shape[0], Np), dtype=d.dtype, device=d.device)
            dp[:, :N] = d
            wp = torch.zeros((WcT.shape[0], Np), dtype=WcT.dtype, device=WcT.device)
            wp[:, :N] = WcT
            d, WcT = dp, wp
        # db = column sums of d, fused into the dW GEMM; without gradient sinks dW and db share one zeroed buffer
        K_ = x2.shape[1]
        sW, sb = ctx.sinks
        if sW is not None and (sb is not None or not ctx.has_b):
            gemm_tn_sink(d, x2, out=sW.view, colsum=sb.view if sb else None)
            dW = db = None
        else:
            buf = torch.zeros((Np * K_ + Np,), dtype=torch.float32, device=d.device)
            dWp, dbp = buf[:Np * K_].view(Np, K_), buf[Np * K_:]
            gemm_tn(d, x2, out=dWp, colsum=dbp if ctx.has_b else None)
            dW = dWp[:N]
            db = dbp[:N] if ctx.has_b else None
            if sW is not None:
                sW.view.add_(dW)
                dW = None
            if sb is not None:
                sb.view.add_(db)
                db = None
        dx = gemm_nt(d, WcT).view(ctx.shp) if ctx.need_dx else None
        return dx, dW, db, None, None


def linear(x, W, b=None, act=ACT_NONE, out_f32=False):
    return LinearFn.apply(x, W, b, act, out_f32)


# ---- residual-stream blocks ------------------------------------------------------------------
# Every block maps the stream triple (x32 fp32 residual value, x compute-dtype copy, xpos = x + pos)
# to the next triple; the pre-norm sum s = x32 + f(x) stays fp32 and never leaves the block.
def _ln_out(s32, gamma, beta, pos_out, dt):
    y32, y, ypos, mean, rstd = layernorm_fwd(s32, gamma, beta, dt, pos_out, want32=True)
    return y32, y, ypos, mean, rstd


def _res_grad(ds32, dt, cs_out):
    """blocks without a post-norm (pre-norm layers): the incoming fp32 stream gradient IS the gradient of the
    pre-norm sum; cast it for the GEMMs and take its column sums (bias gradient of the Linear that fed the sum)."""
    D = ds32.shape[-1]
    ds32 = ds32.reshape(-1, D)
    if not ds32.is_contiguous():
        ds32 = ds32.contiguous()
    g = cast(ds32, dt)
    return ds32, g, colsum(g, out=cs_out)


class MLPLNFn(torch.autograd.Function):
    """LN(x32 + fc2(act(fc1(x))))  — cross_modal_transformer.py:142-143,157-158 + MLP :163-179 (GELU); the FFN of the
    enc/dec Transformer (ReLU, transformer.py:191-193,245-247).  gamma None: no norm, the fp32 sum is the output
    (pre-norm layers, transformer.py:205-207: x is then LN(x32), not a copy of x32)."""

    @staticmethod
    def forward(ctx, x32, x, W1, b1, W2, b2, gamma, beta, pos_out, act=ACT_GELU, drop=None):
        """drop = (p, seed_hidden, seed_residual): training-mode dropout of the enc/dec FFN — linear2(dropout(act(linear1 x))) and
        x + dropout2(.) (transformer.py:168,171,238-240)."""
        ctx.set_materialize_grads(False)
        if drop is not None and drop[0] <= 0.0:
            drop = None
        ctx.drop = drop
        shp = x.shape
        D = shp[-1]
        dt = x.dtype
        x2, x32_2 = x.reshape(-1, D), x32.reshape(-1, D)
        W1c, W1T = weights.get(W1, dt)
        W2c, W2T = weights.get(W2, dt)
        if act == ACT_GELU:
            hid, aux = gemm_nt(x2, W1c, b1, ACT_GELU, want_pre=True)   # aux = pre-activation
        elif act == ACT_RELU:
            hid = aux = gemm_nt(x2, W1c, b1, ACT_RELU)                 # relu' = [hid > 0]
        else:
            raise _lib.SvolHipError('MLPLNFn: activation must be GELU or ReLU')
        if drop is not None:
            dropout(hid, drop[0], drop[1], out=hid)   # (ReLU: aux IS hid — kept entries stay positive, dropped ones get relu' = 0, as the mask wants)
            s32 = dropout_add(gemm_nt(hid, W2c, b2, ACT_NONE, out_f32=True), x32_2 if x32_2.is_contiguous() else x32_2.contiguous(),
                              drop[0], drop[2])
        else:
            s32 = gemm_nt(hid, W2c, b2, ACT_NONE, residual=x32_2, out_f32=True)
        ctx.W1T, ctx.W2T, ctx.shp, ctx.act, ctx.has_ln = W1T, W2T, shp, act, gamma is not None
        ctx.pos_shape = pos_out.shape if pos_out is not None else None
        nig = ctx.needs_input_grad
        ctx.sinks = tuple(_claim(p_, nig[i]) for i, p_ in ((2, W1), (3, b1), (4, W2), (5, b2), (6, gamma), (7, beta)))
        if gamma is None:
            ctx.save_for_backward(x2, aux, hid, None, None, None, None)
            return s32.view(shp)
        y32, y, ypos, mean, rstd = _ln_out(s32, gamma, beta, pos_out, dt)
        ctx.save_for_backward(x2, aux, hid, s32, gamma, mean, rstd)
        if pos_out is not None:
            return y32.view(shp), y.view(shp), ypos.view(shp)
        return y32.view(shp), y.view(shp)

    @staticmethod
    def backward(ctx, dy32, dy=None, dypos=None):
        x2, aux, hid, s32, gamma, mean, rstd = ctx.saved_tensors
        dt = x2.dtype
        D = x2.shape[1]
        dpos = None
        if ctx.pos_shape is not None and ctx.needs_input_grad[8] and dypos is not None:
            dpos = _pos_grad(dypos, ctx.pos_shape, D)
        sW1, sb1, sW2, sb2, sg, sbt = ctx.sinks
        vw = lambda s_: s_.view if s_ is not None else None
        drop = ctx.drop
        if drop is not None:   # the residual branch's gradient passes the residual dropout's mask before it meets any GEMM
            if ctx.has_ln:
                ds32, ds, dg, dbt = layernorm_bwd(dy32, dy, dypos, s32, gamma, mean, rstd, dt, want32=True, dg_out=vw(sg),
                                                  db_out=vw(sbt))
            else:
                ds32 = dy32.reshape(-1, D)
                ds32 = ds32 if ds32.is_contiguous() else ds32.contiguous()
                ds = torch.empty((ds32.shape[0], D), dtype=dt, device=ds32.device)   # (never ds32 itself: that is the stream's gradient)
                dg = dbt = None
                ds = dropout(cast(ds32, dt), drop[0], drop[2], out=ds)
            if ctx.has_ln:
                dropout(ds, drop[0], drop[2], out=ds)
            db2 = colsum(ds, out=vw(sb2))
        elif ctx.has_ln:
            ds32, ds, dg, dbt, db2 = layernorm_bwd(dy32, dy, dypos, s32, gamma, mean, rstd, dt, want32=True,
                                                   want_colsum=True, dg_out=vw(sg), db_out=vw(sbt), cs_out=vw(sb2))
        else:
            ds32, ds, db2 = _res_grad(dy32, dt, vw(sb2))
            dg = dbt = None
        F_ = hid.shape[1]
        if sW1 is not None and sW2 is not None:
            dW1, dW2 = sW1.view, sW2.view
        else:
            wbuf = torch.zeros((2 * D * F_,), dtype=torch.float32, device=ds.device)  # one memset for dW1 | dW2
            dW1, dW2 = wbuf[:D * F_].view(F_, D), wbuf[D * F_:].view(D, F_)
        tn = gemm_tn_sink if (sW1 is not None and sW2 is not None) else (lambda A_, B_, out: gemm_tn(A_, B_, out=out))
        tn(ds, hid, out=dW2)
        # (ds W2) * act'(aux) and its column sums, one kernel
        if drop is not None:   # ... then the hidden dropout's mask, then the column sums
            dpre, _ = gemm_nt_dact(ds, ctx.W2T, aux, ctx.act, want_colsum=False)
            dropout(dpre, drop[0], drop[1], out=dpre)
            db1 = colsum(dpre, out=vw(sb1))
        else:
            dpre, db1 = gemm_nt_dact(ds, ctx.W2T, aux, ctx.act, colsum_out=vw(sb1))
        tn(dpre, x2, out=dW1)
        dx = gemm_nt(dpre, ctx.W1T)
        if (sW1 is None) != (sW2 is None):  # only one of the two has a sink: add the other by hand
            for s_, g_ in ((sW1, dW1), (sW2, dW2)):
                if s_ is not None:
                    s_.view.add_(g_)
        n_ = lambda s_, g_: None if s_ is not None else g_
        return (ds32.view(ctx.shp), dx.view(ctx.shp), n_(sW1, dW1), n_(sb1, db1), n_(sW2, dW2), n_(sb2, db2), n_(sg, dg),
                n_(sbt, dbt), dpos, None, None)


# bf16 mode: the VALUE projections of every attention block multiply by split weights W_hi + W_lo (two bf16 operands = 16
# mantissa bits of the fp32 master weight, one K-concatenated GEMM).  Rounding W_v to bf16 shifts every token's value the same way,
# which attention over thousands of keys and LayerNorm do not average out: at the benchmark depth (6 layers, L = 6272) the V weights
# of the query -> video attention (4.3e-3 rms) and of the video self-attention (2.7e-3) were 95 % of the 6.8e-3 rms logit error, the
# q / k / MLP weights together 1.1e-3 (profiles/round3_bf16_output_error.md).  Forward only: gradients take the plain bf16 copy.
# SVOL_NO_SPLIT_V=1 restores single-bf16 V weights (A/B; changes results).
SPLIT_V = os.environ.get('SVOL_NO_SPLIT_V') is None


class AttnLNFn(torch.autograd.Function):
    """LN(xq32 + out_proj(MHA(q = Wq xq_pos, k = Wk xk_pos, v = Wv xv))) with packed in_proj
    (nn.MultiheadAttention + post-norm, cross_modal_transformer.py:137-141,145-149,151-156; transformer.py:183-189,
    237-250).  ``self_attn``: xk_pos is xq_pos and xv is xq (one packed projection buffer).  gamma None: no norm, the
    fp32 sum is the output (pre-norm layers, transformer.py:198-203: xq is then LN(xq32)).  ``need_weights``: also
    return the head-averaged attention weights [B,Lq,Lk] fp32 (no gradient flows into them)."""

    @staticmethod
    def forward(ctx, xq32, xq, xq_pos, xk_pos, xv, W_in, b_in, W_o, b_o, gamma, beta, pos_out, H, kbias, self_attn,
                need_weights=False, drop=None):
        """drop = (p, seed_attention, seed_residual): training-mode dropout of the enc/dec attention blocks — on the attention
        probabilities (nn.MultiheadAttention(dropout=p)) and on the block output before the residual add (dropout1 / dropout2,
        transformer.py:158-177,216-247).  The returned attention weights stay those of the undropped softmax."""
        ctx.set_materialize_grads(False)
        if drop is not None and drop[0] <= 0.0:
            drop = None
        ctx.drop = drop
        B, Lq, d = xq.shape
        dt = xq.dtype
        Lk = Lq if self_attn else xv.shape[1]
        dh = d // H
        # MIXED cross-attention (the bf16 model's query -> video attention): the few object queries travel in fp32
        # (xq / xq_pos fp32: q-projection, out-projection, residual and norm in exact fp32 GEMMs — a few hundred rows),
        # the L video tokens in bf16 (K / V projections and the attention core on the MFMA bf16 path).  q and O cross
        # the boundary through one rounding each.
        mixed = (not self_attn) and dt == torch.float32 and xv.dtype != torch.float32
        dkv = xv.dtype if mixed else dt
        Wc, WcT = weights.get(W_in, dkv)
        Woc, WoT = weights.get(W_o, dt)
        # bf16: the projection GEMM emits q already multiplied by d_h^-1/2 * log2(e) (rounded once, in the fp32
        # epilogue), which lets the attention kernels exponentiate raw MFMA results
        premul = (LOG2E / math.sqrt(dh)) if dkv != torch.float32 else 0.0
        qscale = _qscale(d, premul, xq.device) if premul else None
        a_qp = xq_pos.reshape(B * Lq, d)
        a_q = xq.reshape(B * Lq, d)
        a_kp = a_qp if self_attn else xk_pos.reshape(B * Lk, d)
        a_v = a_q if self_attn else xv.reshape(B * Lk, d)
        Wq32T = None
        # bf16: the V projection uses SPLIT weights (hi + lo, 16 mantissa bits; svol_gemm_nt_split) — see SPLIT_V
        Wv_hilo = weights.get_split(W_in, 2 * d, d) if (SPLIT_V and dkv == torch.bfloat16 and d % 32 == 0) else None
        if self_attn:
            qkv = torch.empty((B * Lq, 3 * d), dtype=dt, device=xq.device)
            gemm_nt(a_qp, Wc[:2 * d], b_in[:2 * d], out=qkv[:, :2 * d], colscale=qscale)
            if Wv_hilo is not None:
                gemm_nt_split(a_q, Wv_hilo, b_in[2 * d:], out=qkv[:, 2 * d:])
            else:
                gemm_nt(a_q, Wc[2 * d:], b_in[2 * d:], out=qkv[:, 2 * d:])
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        else:
            if mixed:
                W32, W32T = weights.get(W_in, torch.float32)
                Wq32T = W32T[:, :d]
                q = cast(gemm_nt(a_qp, W32[:d], b_in[:d], colscale=qscale[:d]), dkv)
            else:
                q = gemm_nt(a_qp, Wc[:d], b_in[:d], colscale=qscale[:d] if qscale is not None else None)
            kv = torch.empty((B * Lk, 2 * d), dtype=dkv, device=xq.device)
            gemm_nt(a_kp, Wc[d:2 * d], b_in[d:2 * d], out=kv[:, :d])
            if Wv_hilo is not None:
                gemm_nt_split(a_v, Wv_hilo, b_in[2 * d:], out=kv[:, d:])
            else:
                gemm_nt(a_v, Wc[2 * d:], b_in[2 * d:], out=kv[:, d:])
            k, v = kv[:, :d], kv[:, d:]
        o, lse2 = attn_fwd(q, k, v, B, H, Lq, Lk, dh, kbias, premul, drop=(drop[0], drop[1]) if drop is not None else None)
        att = attn_weights_mean(q, k, lse2, B, H, Lq, Lk, dh, kbias, premul) if need_weights else None
        if drop is not None:
            r32 = xq32.reshape(B * Lq, d)
            s32 = dropout_add(gemm_nt(cast(o, dt) if mixed else o, Woc, b_o, out_f32=True), r32 if r32.is_contiguous() else r32.contiguous(),
                              drop[0], drop[2])
        else:
            s32 = gemm_nt(cast(o, dt) if mixed else o, Woc, b_o, residual=xq32.reshape(B * Lq, d), out_f32=True)
        ctx.WcT, ctx.WoT, ctx.dims, ctx.self_attn, ctx.premul = WcT, WoT, (B, H, Lq, Lk, dh, d), self_attn, premul
        ctx.mixed, ctx.Wq32T = mixed, Wq32T
        ctx.pos_shape = pos_out.shape if pos_out is not None else None
        ctx.has_ln = gamma is not None
        nig = ctx.needs_input_grad
        ctx.sinks = tuple(_claim(p_, nig[i]) for i, p_ in ((5, W_in), (6, b_in), (7, W_o), (8, b_o), (9, gamma),
                                                           (10, beta)))
        if any(s_ is None for s_ in ctx.sinks[:6 if ctx.has_ln else 4]):  # all or nothing
            ctx.sinks = None
        shp = (B, Lq, d)
        if att is not None:
            ctx.mark_non_differentiable(att)
        tail = (att,) if att is not None else ()
        if gamma is None:
            ctx.save_for_backward(a_qp, a_q, a_kp, a_v, q, k, v, o, lse2, kbias, None, None, None, None)
            return (s32.view(shp),) + tail if tail else s32.view(shp)
        y32, y, ypos, mean, rstd = _ln_out(s32, gamma, beta, pos_out, dt)
        ctx.save_for_backward(a_qp, a_q, a_kp, a_v, q, k, v, o, lse2, kbias, s32, gamma, mean, rstd)
        if pos_out is not None:
            return (y32.view(shp), y.view(shp), ypos.view(shp)) + tail
        return (y32.view(shp), y.view(shp)) + tail

    @staticmethod
    def backward(ctx, dy32, *rest):
        a_qp, a_q, a_kp, a_v, q, k, v, o, lse2, kbias, s32, gamma, mean, rstd = ctx.saved_tensors
        B, H, Lq, Lk, dh, d = ctx.dims
        dt = a_q.dtype
        WcT, WoT = ctx.WcT, ctx.WoT  # WcT: [d, 3d] ; WoT: [d, d]
        dy = rest[0] if ctx.has_ln and len(rest) > 0 else None
        dypos = rest[1] if ctx.has_ln and ctx.pos_shape is not None and len(rest) > 1 else None
        dpos = None
        if ctx.pos_shape is not None and ctx.needs_input_grad[11] and dypos is not None:
            dpos = _pos_grad(dypos, ctx.pos_shape, d)
        sk = ctx.sinks
        dg = dbt = None
        drop = ctx.drop
        if drop is not None:   # the block output's gradient passes the residual dropout's mask before out_proj's backward
            if ctx.has_ln:
                ds32, g, dg, dbt = layernorm_bwd(dy32, dy, dypos, s32, gamma, mean, rstd, dt, want32=True,
                                                 dg_out=sk[4].view if sk is not None else None,
                                                 db_out=sk[5].view if sk is not None else None)
            else:
                ds32 = dy32.reshape(-1, d)
                ds32 = ds32 if ds32.is_contiguous() else ds32.contiguous()
                g = dropout(cast(ds32, dt), drop[0], drop[2], out=torch.empty((ds32.shape[0], d), dtype=dt, device=ds32.device))
            if ctx.has_ln:
                dropout(g, drop[0], drop[2], out=g)
            dbo = colsum(g, out=sk[3].view if sk is not None else None)
            if sk is not None:
                dW_in, db_in, dWo = sk[0].view, sk[1].view, sk[2].view
        elif sk is not None:
            dW_in, db_in, dWo = sk[0].view, sk[1].view, sk[2].view
            if ctx.has_ln:
                ds32, g, dg, dbt, dbo = layernorm_bwd(dy32, dy, dypos, s32, gamma, mean, rstd, dt, want32=True,
                                                      want_colsum=True, dg_out=sk[4].view, db_out=sk[5].view,
                                                      cs_out=sk[3].view)
            else:
                ds32, g, dbo = _res_grad(dy32, dt, sk[3].view)
        else:
            if ctx.has_ln:
                ds32, g, dg, dbt, dbo = layernorm_bwd(dy32, dy, dypos, s32, gamma, mean, rstd, dt, want32=True,
                                                      want_colsum=True)
            else:
                ds32, g, dbo = _res_grad(dy32, dt, None)
        if sk is None:
            gbuf = torch.zeros((3 * d * d + 3 * d + d * d,), dtype=torch.float32, device=g.device)  # one memset
            dW_in, db_in = gbuf[:3 * d * d].view(3 * d, d), gbuf[3 * d * d:3 * d * d + 3 * d]
            dWo = gbuf[3 * d * d + 3 * d:].view(d, d)
        mixed = ctx.mixed
        adrop = (drop[0], drop[1]) if drop is not None else None
        tn = gemm_tn_sink if sk is not None else (lambda A_, B_, out, colsum=None: gemm_tn(A_, B_, out=out, colsum=colsum))
        tn(g, cast(o, dt) if mixed else o, out=dWo)
        do = gemm_nt(g, WoT)
        if mixed:
            do = cast(do, o.dtype)
        shq = (B, Lq, d)
        if ctx.self_attn:
            dqkv = torch.empty((B * Lq, 3 * d), dtype=dt, device=g.device)
            dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
            attn_bwd(q, k, v, o, do, lse2, B, H, Lq, Lk, dh, dq, dk, dv, kbias, ctx.premul, drop=adrop)
            tn(dqkv[:, :2 * d], a_qp, out=dW_in[:2 * d], colsum=db_in[:2 * d])
            tn(dv, a_q, out=dW_in[2 * d:], colsum=db_in[2 * d:])
            dxq_pos = gemm_nt(dqkv[:, :2 * d], WcT[:, :2 * d])  # d(x + pos) = [dq dk] W_qk
            dxq = gemm_nt(dv, WcT[:, 2 * d:])                    # d(x) through V
            if sk is not None:
                dW_in = db_in = dWo = dbo = dg = dbt = None
            return (ds32.view(shq), dxq.view(shq), dxq_pos.view(shq), None, None, dW_in, db_in, dWo, dbo, dg, dbt,
                    dpos, None, None, None, None, None)
        dq = torch.empty((B * Lq, d), dtype=q.dtype, device=g.device)
        dkv = torch.empty((B * Lk, 2 * d), dtype=k.dtype, device=g.device)
        dk, dv = dkv[:, :d], dkv[:, d:]
        attn_bwd(q, k, v, o, do, lse2, B, H, Lq, Lk, dh, dq, dk, dv, kbias, ctx.premul, drop=adrop)
        if mixed:
            dq = cast(dq, dt)
        tn(dq, a_qp, out=dW_in[:d], colsum=db_in[:d])
        tn(dk, a_kp, out=dW_in[d:2 * d], colsum=db_in[d:2 * d])
        tn(dv, a_v, out=dW_in[2 * d:], colsum=db_in[2 * d:])
        dxq_pos = gemm_nt(dq, ctx.Wq32T if mixed else WcT[:, :d])
        dxk_pos = gemm_nt(dk, WcT[:, d:2 * d])
        dxv = gemm_nt(dv, WcT[:, 2 * d:])
        shk = (B, Lk, d)
        if sk is not None:
            dW_in = db_in = dWo = dbo = dg = dbt = None
        return (ds32.view(shq), None, dxq_pos.view(shq), dxk_pos.view(shk), dxv.view(shk), dW_in, db_in, dWo, dbo, dg,
                dbt, dpos, None, None, None, None, None)


class LNStreamFn(torch.autograd.Function):
    """(y, y + pos) = LN(x32) in the compute dtype from the fp32 residual stream — the norm that OPENS a pre-norm
    block (transformer.py:198-199, 205, 267-268, 274, 280) and the decoder's shared output norm."""

    @staticmethod
    def forward(ctx, x32, gamma, beta, pos, dtype, want32):
        ctx.set_materialize_grads(False)
        shp = x32.shape
        D = shp[-1]
        x2 = x32.reshape(-1, D)
        y32, y, ypos, mean, rstd = layernorm_fwd(x2, gamma, beta, dtype, pos, want32=want32)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.shp, ctx.dtype, ctx.has_pos, ctx.want32 = shp, dtype, pos is not None, want32
        ctx.pos_shape = pos.shape if pos is not None else None
        ctx.sinks = (_claim(gamma, ctx.needs_input_grad[1]), _claim(beta, ctx.needs_input_grad[2]))
        out = ((y32.view(shp),) if want32 else ()) + (y.view(shp),) + ((ypos.view(shp),) if pos is not None else ())
        return out if len(out) > 1 else out[0]

    @staticmethod
    def backward(ctx, *grads):
        x2, gamma, mean, rstd = ctx.saved_tensors
        g = list(grads)
        dy32 = g.pop(0) if ctx.want32 else None
        dy = g.pop(0)
        dypos = g.pop(0) if ctx.has_pos else None
        if dy32 is None and dy is None and dypos is None:
            return None, None, None, None, None, None
        sg, sb = ctx.sinks
        dpos = None
        if dypos is not None and ctx.needs_input_grad[3]:
            dpos = _pos_grad(dypos, ctx.pos_shape, x2.shape[1])
        dx32, _, dg, db = layernorm_bwd(dy32, dy, dypos, x2, gamma, mean, rstd, ctx.dtype, want32=True, want_t=False,
                                        dg_out=sg.view if sg else None, db_out=sb.view if sb else None)
        return dx32.view(ctx.shp), None if sg else dg, None if sb else db, dpos, None, None


LOG2E = 1.4426950408889634
_QSCALE = {}


def _qscale(d, premul, device):
    """per-column epilogue factor of the packed q|k projection: premul on the q columns, 1 on the k columns."""
    key = (d, premul, str(device))
    t = _QSCALE.get(key)
    if t is None:
        t = torch.ones((2 * d,), dtype=torch.float32, device=device)
        t[:d] = premul
        _QSCALE[key] = t
    return t


def mlp_ln(x32, x, W1, b1, W2, b2, gamma, beta, pos_out=None, act=ACT_GELU, drop=None):
    return MLPLNFn.apply(x32, x, W1, b1, W2, b2, gamma, beta, pos_out, act, drop)


def self_attn_ln(x32, x, x_pos, W_in, b_in, W_o, b_o, gamma, beta, pos_out, H, kbias=None, drop=None):
    return AttnLNFn.apply(x32, x, x_pos, None, None, W_in, b_in, W_o, b_o, gamma, beta, pos_out, H, kbias, True, False, drop)


def cross_attn_ln(xq32, xq, xq_pos, xk_pos, xv, W_in, b_in, W_o, b_o, gamma, beta, pos_out, H, kbias,
                  need_weights=False, drop=None):
    return AttnLNFn.apply(xq32, xq, xq_pos, xk_pos, xv, W_in, b_in, W_o, b_o, gamma, beta, pos_out, H, kbias, False,
                          need_weights, drop)


def ln_stream(x32, gamma, beta, pos, dtype, want32=False):
    return LNStreamFn.apply(x32, gamma, beta, pos, dtype, want32)


class GateFn(torch.autograd.Function):
    """(mem32, mem, mem + pos) with mem = LN1(x * (1 + a)), a = head-mean softmax of the 1-query
    sketch->video attention (cross_modal_transformer.py:122-127); u = per-(batch, head) folded
    key projection [B,H,d] fp32.  x32 is the fp32 residual stream."""

    @staticmethod
    def forward(ctx, x32, pos, u, gamma, beta, H):
        ctx.set_materialize_grads(False)
        B, L, D = x32.shape
        dt = pos.dtype
        x2 = x32.reshape(B * L, D).contiguous()
        assert x2.dtype == torch.float32
        pos2 = pos.reshape(B * L, D).contiguous()
        u = u.contiguous().float()
        y32 = torch.empty((B * L, D), dtype=torch.float32, device=x2.device)
        y = torch.empty((B * L, D), dtype=dt, device=x2.device)
        ypos = torch.empty((B * L, D), dtype=dt, device=x2.device)
        a = torch.empty((B * L,), dtype=torch.float32, device=x2.device)
        mean = torch.empty_like(a)
        rstd = torch.empty_like(a)
        ws = torch.empty((B * H * (L + 2),), dtype=torch.float32, device=x2.device)
        rc = _lib.lib().svol_gate_fwd(_ptr(x2), _ptr(pos2), _ptr(u), _ptr(gamma), _ptr(beta), _ptr(y32), _ptr(y),
                                      _ptr(ypos), _ptr(a), _ptr(mean), _ptr(rstd), _ptr(ws), B, L, D, H, _DT[dt],
                                      _stream())
        _lib.check(rc, 'svol_gate_fwd')
        ctx.save_for_backward(x2, pos2, u, gamma, a, mean, rstd, ws)
        ctx.dims = (B, L, D, H)
        ctx.sinks = (_claim(gamma, ctx.needs_input_grad[3]), _claim(beta, ctx.needs_input_grad[4]))
        return y32.view(B, L, D), y.view(B, L, D), ypos.view(B, L, D)

    @staticmethod
    def backward(ctx, dy32, dy, dypos):
        x2, pos2, u, gamma, a, mean, rstd, ws = ctx.saved_tensors
        B, L, D, H = ctx.dims
        cont = lambda t: None if t is None else t.reshape(B * L, D).contiguous()
        dy32, dy, dypos = cont(dy32), cont(dy), cont(dypos)
        dx32 = torch.empty_like(x2)
        sg, sb = ctx.sinks
        zb = torch.zeros((B * H * D + 2 * D,), dtype=torch.float32, device=x2.device)  # one memset: du | dgamma | dbeta
        du = zb[:B * H * D].view(B, H, D)
        dg = sg.view if sg else zb[B * H * D:B * H * D + D]
        db = sb.view if sb else zb[B * H * D + D:]
        ws2 = torch.empty((B * L + B * H,), dtype=torch.float32, device=x2.device)
        rc = _lib.lib().svol_gate_bwd(_ptr(dy32), _ptr(dy), _ptr(dypos), _ptr(x2), _ptr(pos2), _ptr(u), _ptr(gamma),
                                      _ptr(a), _ptr(mean), _ptr(rstd), _ptr(ws), _ptr(ws2), _ptr(dx32), _ptr(du),
                                      _ptr(dg), _ptr(db), B, L, D, H, _DT[pos2.dtype], _stream())
        _lib.check(rc, 'svol_gate_bwd')
        return dx32.view(B, L, D), None, du, None if sg else dg, None if sb else db, None


def gate(x32, pos, u, gamma, beta, H):
    return GateFn.apply(x32, pos, u, gamma, beta, H)


class CastFn(torch.autograd.Function):
    """dtype change on the autograd tape (fp32 master parameter / feature -> compute dtype and back)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return cast(dy.contiguous(), ctx.src), None


def cast_ag(x, dtype):
    return x if x.dtype == dtype else CastFn.apply(x, dtype)


# ---- the forward -> backward turn: heads, criterion glue (csrc/heads.hip) ------------------------------------------------------------
FUSED_HEADS = os.environ.get('SVOL_NO_FUSED_HEADS') is None and os.environ.get('SVOL_DETERMINISTIC') is None


def heads_fusable(hs, class_embed, bbox_embed):
    """the one-launch heads take: fp32 decoder states of width D (multiple of 32, <= 512), Linear(D, 2) and a 3-layer MLP(D, D, 4)."""
    if not FUSED_HEADS or hs.dtype != torch.float32 or not hs.is_cuda:
        return False
    D = hs.shape[-1]
    ls = list(bbox_embed.layers)
    return (D % 32 == 0 and D <= 512 and len(ls) == 3 and tuple(class_embed.weight.shape) == (2, D) and class_embed.bias is not None
            and tuple(ls[0].weight.shape) == (D, D) and tuple(ls[1].weight.shape) == (D, D) and tuple(ls[2].weight.shape) == (4, D)
            and all(l.bias is not None for l in ls) and all(t.dtype == torch.float32 for t in (class_embed.weight, ls[0].weight)))


class HeadsFn(torch.autograd.Function):
    """(pred_logits, pred_boxes) of ALL decoder layers from the stacked states hs [..., D]: class_embed = Linear(D, 2) and
    bbox_embed = MLP(D, D, 4, 3) -> sigmoid (svanet.py:125-127) in ONE launch, exact fp32; the backward is one launch for dhs and the
    2- / 4-row weight gradients, the two D x D weight gradients go to the weight-gradient stream."""

    @staticmethod
    def forward(ctx, hs, Wc, bc, W0, b0, W1, b1, W2, b2):
        shp = hs.shape
        D = shp[-1]
        hs2 = hs.reshape(-1, D)
        if not hs2.is_contiguous():
            hs2 = hs2.contiguous()
        R = hs2.shape[0]
        dev = hs.device
        logits = torch.empty((R, 2), dtype=torch.float32, device=dev)
        boxes = torch.empty((R, 4), dtype=torch.float32, device=dev)
        hid = torch.empty((2, R, D), dtype=torch.float32, device=dev)
        rc = _lib.lib().svol_heads_fwd(_ptr(hs2), _ptr(Wc), _ptr(bc), _ptr(W0), _ptr(b0), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(logits),
                                       _ptr(hid[0]), _ptr(hid[1]), _ptr(boxes), R, D, _stream())
        _lib.check(rc, 'svol_heads_fwd')
        ctx.save_for_backward(hs2, hid, boxes, Wc, W0, W1, W2)
        ctx.shp = shp
        ni = ctx.needs_input_grad
        ctx.sinks = tuple(_claim(p_, ni[i + 1]) for i, p_ in enumerate((Wc, bc, W0, b0, W1, b1, W2, b2)))
        return logits.view(*shp[:-1], 2), boxes.view(*shp[:-1], 4)

    @staticmethod
    def backward(ctx, dlogits, dboxes):
        hs2, hid, boxes, Wc, W0, W1, W2 = ctx.saved_tensors
        R, D = hs2.shape
        dev = hs2.device
        dlogits = torch.zeros((R, 2), dtype=torch.float32, device=dev) if dlogits is None else dlogits.reshape(R, 2).float().contiguous()
        dboxes = torch.zeros((R, 4), dtype=torch.float32, device=dev) if dboxes is None else dboxes.reshape(R, 4).float().contiguous()
        sWc, sbc, sW0, sb0, sW1, sb1, sW2, sb2 = ctx.sinks
        g = torch.empty((3, R, D), dtype=torch.float32, device=dev)   # dhs, g_pre0, g_pre1
        small = None
        if not (sWc is not None and sbc is not None and sW2 is not None and sb2 is not None):
            small = torch.zeros((6 * D + 6,), dtype=torch.float32, device=dev)
        dWc = sWc.view if sWc is not None else small[:2 * D].view(2, D)
        dbc = sbc.view if sbc is not None else small[6 * D:6 * D + 2]
        dW2 = sW2.view if sW2 is not None else small[2 * D:6 * D].view(4, D)
        db2 = sb2.view if sb2 is not None else small[6 * D + 2:]
        rc = _lib.lib().svol_heads_bwd(_ptr(dlogits), _ptr(dboxes), _ptr(hs2), _ptr(hid[0]), _ptr(hid[1]), _ptr(boxes), _ptr(Wc), _ptr(W0),
                                       _ptr(W1), _ptr(W2), _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(dWc), _ptr(dbc), _ptr(dW2), _ptr(db2),
                                       R, D, _stream())
        _lib.check(rc, 'svol_heads_bwd')
        ni = ctx.needs_input_grad
        out = [g[0].view(ctx.shp) if ni[0] else None]
        out += [None if sWc is not None else dWc, None if sbc is not None else dbc]
        for gp, x, sW, sb, iw in ((g[1], hs2, sW0, sb0, 3), (g[2], hid[0], sW1, sb1, 5)):
            if sW is not None and sb is not None:
                gemm_tn_sink(gp, x, out=sW.view, colsum=sb.view)      # off the critical path (weight-gradient stream)
                out += [None, None]
            else:
                buf = torch.zeros((D * D + D,), dtype=torch.float32, device=dev)
                gemm_tn(gp, x, out=buf[:D * D].view(D, D), colsum=buf[D * D:])
                dW_, db_ = buf[:D * D].view(D, D), buf[D * D:]
                if sW is not None:
                    sW.view.add_(dW_)
                    dW_ = None
                if sb is not None:
                    sb.view.add_(db_)
                    db_ = None
                out += [dW_ if ni[iw] else None, db_ if ni[iw + 1] else None]
        out += [None if sW2 is not None else dW2, None if sb2 is not None else db2]
        return tuple(out)


def heads(hs, class_embed, bbox_embed):
    ls = bbox_embed.layers
    return HeadsFn.apply(hs, class_embed.weight, class_embed.bias, ls[0].weight, ls[0].bias, ls[1].weight, ls[1].bias, ls[2].weight, ls[2].bias)


class WeightedTotalFn(torch.autograd.Function):
    """sum(losses * w) over the [layers, 4] loss table (train.py:227-228) as one tiny launch each way."""

    @staticmethod
    def forward(ctx, losses, w):
        x = losses.contiguous()
        out = torch.empty((), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().svol_weighted_total(_ptr(x), _ptr(w), x.numel(), _ptr(out), _stream()), 'svol_weighted_total')
        ctx.save_for_backward(w)
        ctx.shape = losses.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (w,) = ctx.saved_tensors
        d = dout.float().contiguous()
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().svol_weighted_total_bwd(_ptr(w), _ptr(d), w.numel(), _ptr(dx), _stream()), 'svol_weighted_total_bwd')
        return dx, None


class SetCriterionFn(torch.autograd.Function):
    """All decoder layers' matching + losses in three launches (cost, LSAP, loss); no host sync.
    logits [NL,B,N,2], boxes [NL,B,N,4] fp32 -> losses [NL,4] = (label, bbox, giou, class_error)."""

    @staticmethod
    def forward(ctx, logits, boxes, packed, w_bbox, w_giou, w_class, eos_coef):
        NL = logits.shape[0]
        rows = logits.shape[1] * logits.shape[2]
        lg = logits.contiguous().float()
        bx = boxes.contiguous().float()
        match = match_all(lg, bx, packed, w_bbox, w_giou, w_class)
        losses = torch.empty((NL, 4), dtype=torch.float32, device=lg.device)
        g_label = torch.empty_like(lg)
        g_bbox = torch.empty_like(bx)
        g_giou = torch.empty_like(bx)
        rc = _lib.lib().svol_set_loss(_ptr(lg), _ptr(bx), _ptr(packed.tgt_boxes), _ptr(match), _ptr(losses),
                                      _ptr(g_label), _ptr(g_bbox), _ptr(g_giou), NL, rows, float(eos_coef),
                                      _ptr(packed.rebase_vid_off), logits.shape[2], _ptr(packed.status),
                                      _ptr(packed.box_status), packed.problems_per_layer, _stream())
        _lib.check(rc, 'svol_set_loss')
        ctx.save_for_backward(g_label, g_bbox, g_giou)
        ctx.mark_non_differentiable(match)
        return losses, match

    @staticmethod
    def backward(ctx, dl, _dmatch):
        g_label, g_bbox, g_giou = ctx.saved_tensors
        dl = dl.float()
        if FUSED_HEADS and dl.is_contiguous():   # one launch instead of five elementwise ones (csrc/heads.hip)
            dlog, dbox = torch.empty_like(g_label), torch.empty_like(g_bbox)
            NL = g_label.shape[0]
            rc = _lib.lib().svol_set_loss_bwd(_ptr(g_label), _ptr(g_bbox), _ptr(g_giou), _ptr(dl), _ptr(dlog), _ptr(dbox), NL,
                                              g_label.numel() // (2 * NL), _stream())
            _lib.check(rc, 'svol_set_loss_bwd')
            return dlog, dbox, None, None, None, None, None
        dlog = g_label * dl[:, 0].view(-1, 1, 1, 1)
        dbox = g_bbox * dl[:, 1].view(-1, 1, 1, 1) + g_giou * dl[:, 2].view(-1, 1, 1, 1)
        return dlog, dbox, None, None, None, None, None


def match_all(lg, bx, packed, w_bbox, w_giou, w_class):
    """cost blocks + batched LSAP for every problem in `packed`; returns match[R] int32
    (global target row per prediction row, -1 = unmatched)."""
    R = lg.shape[0] * lg.shape[1] * lg.shape[2]
    cost = torch.empty((max(1, packed.cost_numel),), dtype=torch.float32, device=lg.device)
    match = torch.empty((R,), dtype=torch.int32, device=lg.device)
    L = _lib.lib()
    rc = L.svol_match_cost(_ptr(lg), _ptr(bx), _ptr(packed.tgt_boxes), _ptr(packed.pred_off), _ptr(packed.pred_cnt),
                           _ptr(packed.tgt_off), _ptr(packed.tgt_cnt), _ptr(packed.cost_off), _ptr(cost),
                           packed.n_problems, float(w_bbox), float(w_giou), float(w_class), _ptr(packed.box_status),
                           _stream())
    _lib.check(rc, 'svol_match_cost')
    rc = L.svol_lsap_batched(_ptr(cost), _ptr(packed.cost_off), _ptr(packed.pred_off), _ptr(packed.pred_cnt),
                             _ptr(packed.tgt_off), _ptr(packed.tgt_cnt), _ptr(match), _ptr(packed.status),
                             packed.n_problems, packed.max_dim, _stream())
    _lib.check(rc, 'svol_lsap_batched')
    packed.last_cost = cost
    return match
