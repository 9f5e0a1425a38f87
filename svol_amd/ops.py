"""Host-side operators of the SVOL hot path: thin tensor wrappers over the C-ABI
(libsvol_hip.so) and the ``torch.autograd.Function`` s that compose them.

PyTorch here is plumbing only: device memory, the current HIP stream and the
autograd tape.  Every arithmetic op on the hot path is a hand-written gfx950
kernel behind ``include/svol_hip.h``; there is no CPU or eager fallback — a
non-CUDA tensor or a missing library raises.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3
_DT = {torch.float32: 0, torch.bfloat16: 1}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise _lib.SvolHipError(f'unsupported dtype {t.dtype}')


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> int:
    if t is None:
        return 0
    if not t.is_cuda:
        raise _lib.SvolHipError('svol_amd ops need device tensors (no CPU fallback)')
    return t.data_ptr()


def _rows(t: torch.Tensor) -> torch.Tensor:
    """view [..., D] as a 2-D row-major matrix (last dim contiguous)."""
    if t.stride(-1) != 1:
        t = t.contiguous()
    return t


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


def gemm_nt(A, B, bias=None, act=ACT_NONE, residual=None, want_pre=False, out=None, K=None, N=None):
    """out[M,N] = act(A[M,K] @ B[N,K]^T + bias) + residual.  A/B/out/residual may be column slices
    of wider row-major buffers (stride(0) is the leading dimension)."""
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    M = A.shape[0]
    K = A.shape[1] if K is None else K
    N = B.shape[0] if N is None else N
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    pre = torch.empty((M, N), dtype=A.dtype, device=A.device) if want_pre else None
    if pre is not None:
        assert out.stride(0) == pre.stride(0)
    rc = _lib.lib().svol_gemm_nt(_ptr(A), A.stride(0), 0, 0, _ptr(B), B.stride(0), _ptr(out), out.stride(0),
                                 _ptr(bias), act, _ptr(pre), _ptr(residual),
                                 residual.stride(0) if residual is not None else 0, M, N, K, _dt(A), _stream())
    _lib.check(rc, 'svol_gemm_nt')
    return (out, pre) if want_pre else out


def gemm_tn(A, B, out=None):
    """out[N,K] (fp32) += A[Mc,N]^T @ B[Mc,K]; a fresh zeroed `out` is allocated when not given."""
    assert A.dim() == 2 and B.dim() == 2 and A.shape[0] == B.shape[0]
    assert A.stride(1) == 1 and B.stride(1) == 1
    Mc, N = A.shape
    K = B.shape[1]
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=A.device)
    rc = _lib.lib().svol_gemm_tn(_ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), out.stride(0), Mc, N, K,
                                 _dt(A), _stream())
    _lib.check(rc, 'svol_gemm_tn')
    return out


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


def layernorm_fwd(x, gamma, beta, pos=None, p=0.0, seed=0):
    x = x.contiguous()
    M, D = x.shape
    y = torch.empty_like(x)
    ypos = torch.empty_like(x) if pos is not None else None
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    if pos is not None:
        pos = pos.contiguous()
        assert pos.dtype == x.dtype and pos.shape[-1] == D
    rc = _lib.lib().svol_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(ypos), _ptr(pos),
                                       pos.numel() // D if pos is not None else 0, _ptr(mean), _ptr(rstd), M, D,
                                       float(p), int(seed), _dt(x), _stream())
    _lib.check(rc, 'svol_layernorm_fwd')
    return y, ypos, mean, rstd


def layernorm_bwd(dy, dy2, x, gamma, mean, rstd, p=0.0, seed=0):
    dy = dy.contiguous()
    if dy2 is not None:
        dy2 = dy2.contiguous()
    M, D = x.shape
    dx = torch.empty_like(x)
    dg = torch.zeros((D,), dtype=torch.float32, device=x.device)
    db = torch.zeros((D,), dtype=torch.float32, device=x.device)
    rc = _lib.lib().svol_layernorm_bwd(_ptr(dy), _ptr(dy2), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx),
                                       _ptr(dg), _ptr(db), M, D, float(p), int(seed), _dt(x), _stream())
    _lib.check(rc, 'svol_layernorm_bwd')
    return dx, dg, db


def posenc_sine(mask_f32: torch.Tensor, D: int, dtype: torch.dtype) -> torch.Tensor:
    mask_f32 = mask_f32.contiguous()
    B, L = mask_f32.shape
    pos = torch.empty((B, L, D), dtype=dtype, device=mask_f32.device)
    _lib.check(_lib.lib().svol_posenc_sine(_ptr(mask_f32), _ptr(pos), B, L, D, _DT[dtype], _stream()),
               'svol_posenc_sine')
    return pos


def attn_fwd(q, k, v, B, H, Lq, Lk, dh, kbias=None):
    """q/k/v: 2-D [B*L, >=H*dh] views (column slices allowed). Returns o [B*Lq, H*dh], lse2 [B,H,Lq]."""
    o = torch.empty((B * Lq, H * dh), dtype=q.dtype, device=q.device)
    lse2 = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    rc = _lib.lib().svol_attn_fwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o),
                                  o.stride(0), _ptr(lse2), _ptr(kbias), B, H, Lq, Lk, dh, 1.0 / math.sqrt(dh),
                                  _dt(q), _stream())
    _lib.check(rc, 'svol_attn_fwd')
    return o, lse2


def attn_bwd(q, k, v, o, do, lse2, B, H, Lq, Lk, dh, dq, dk, dv, kbias=None):
    """Writes dq/dk/dv (2-D views, column slices allowed)."""
    do = do if do.stride(1) == 1 else do.contiguous()
    delta = torch.empty((B, H, Lq), dtype=torch.float32, device=q.device)
    rc = _lib.lib().svol_attn_bwd(_ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v), v.stride(0), _ptr(o),
                                  o.stride(0), _ptr(do), do.stride(0), _ptr(lse2), _ptr(delta), _ptr(kbias),
                                  _ptr(dq), dq.stride(0), _ptr(dk), dk.stride(0), _ptr(dv), dv.stride(0), B, H, Lq,
                                  Lk, dh, 1.0 / math.sqrt(dh), _dt(q), _stream())
    _lib.check(rc, 'svol_attn_bwd')


# ----------------------------------------------------------------------------
# parameter cache: fp32 master weights -> compute-dtype copies (W and W^T), refreshed when the
# optimizer bumps the tensor version.  One cast per weight per step.
# ----------------------------------------------------------------------------
class _WeightCache:
    def __init__(self):
        self._c = {}

    def get(self, w: torch.Tensor, dtype: torch.dtype):
        key = (w.data_ptr(), dtype, tuple(w.shape))
        ent = self._c.get(key)
        ver = w._version
        if ent is None or ent[0] != ver:
            wd = w.detach()
            if dtype == torch.float32:
                _, wt = cast_transpose(wd, dtype, want=False)
                ent = (ver, wd.contiguous(), wt)
            else:
                wc, wt = cast_transpose(wd, dtype)
                ent = (ver, wc, wt)
            self._c[key] = ent
        return ent[1], ent[2]

    def clear(self):
        self._c.clear()


weights = _WeightCache()


# ----------------------------------------------------------------------------
# autograd Functions
# ----------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    """y = dropout(LN(x)) [, ypos = y + pos].  svanet.py:168-178 / post-norms of
    cross_modal_transformer.py:127-158."""

    @staticmethod
    def forward(ctx, x, gamma, beta, pos, p, seed):
        ctx.set_materialize_grads(False)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        y, ypos, mean, rstd = layernorm_fwd(x2, gamma, beta, pos, p, seed)
        ctx.save_for_backward(x2, gamma, mean, rstd)
        ctx.p, ctx.seed, ctx.shp, ctx.has_pos = p, seed, shp, pos is not None
        if pos is not None:
            ctx.pos_numel, ctx.pos_shape = pos.numel(), pos.shape
        if pos is not None:
            return y.view(shp), ypos.view(shp)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy, dypos=None):
        x2, gamma, mean, rstd = ctx.saved_tensors
        D = x2.shape[1]
        if dy is None:
            dy, dypos = dypos, None
        dpos = None
        if ctx.has_pos and ctx.needs_input_grad[3]:
            src = dypos if dypos is not None else dy  # dy was swapped in when only ypos was used
            rows = ctx.pos_numel // D
            dpos = colsum(src.reshape(-1, rows * D).contiguous()).view(ctx.pos_shape).to(src.dtype)
        dx, dg, db = layernorm_bwd(dy.reshape(-1, D), dypos.reshape(-1, D) if dypos is not None else None, x2, gamma,
                                   mean, rstd, ctx.p, ctx.seed)
        return dx.view(ctx.shp), dg, db, dpos, None, None


def layer_norm(x, gamma, beta, pos=None, p=0.0, seed=0):
    return LayerNormFn.apply(x, gamma, beta, pos, p, seed)


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) (nn.Linear + F.relu / sigmoid)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if act == ACT_GELU:
            raise _lib.SvolHipError('LinearFn: GELU is only available fused in MLPResFn')
        Wc, WcT = weights.get(W, x.dtype)
        y = gemm_nt(x2, Wc, b, act)
        ctx.save_for_backward(x2, y if act != ACT_NONE else None)
        ctx.WcT, ctx.act, ctx.shp, ctx.has_b = WcT, act, shp, b is not None
        ctx.need_dx = ctx.needs_input_grad[0]
        return y.view(*shp[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, y = ctx.saved_tensors
        N = dy.shape[-1]
        d = dy.reshape(-1, N)
        if not d.is_contiguous():
            d = d.contiguous()
        if ctx.act != ACT_NONE:
            d = act_bwd(d, y, ctx.act)
        db = colsum(d) if ctx.has_b else None
        # out-features that are not a multiple of the 16-byte chunk (2-class / 4-coordinate heads):
        # zero-pad the contraction / column dimension
        epc = 8 if d.dtype == torch.bfloat16 else 4
        WcT = ctx.WcT
        if N % epc:
            Np = (N + epc - 1) // epc * epc
            dp = torch.zeros((d.shape[0], Np), dtype=d.dtype, device=d.device)
            dp[:, :N] = d
            wp = torch.zeros((WcT.shape[0], Np), dtype=WcT.dtype, device=WcT.device)
            wp[:, :N] = WcT
            d, WcT = dp, wp
        dW = gemm_tn(d, x2)[:N]
        dx = gemm_nt(d, WcT).view(ctx.shp) if ctx.need_dx else None
        return dx, dW, db, None


def linear(x, W, b=None, act=ACT_NONE):
    return LinearFn.apply(x, W, b, act)


class MLPResFn(torch.autograd.Function):
    """y = x + fc2(gelu(fc1(x)))  — cross_modal_transformer.py:142,157 + MLP :163-179."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        W1c, W1T = weights.get(W1, x.dtype)
        W2c, W2T = weights.get(W2, x.dtype)
        hid, pre = gemm_nt(x2, W1c, b1, ACT_GELU, want_pre=True)
        y = gemm_nt(hid, W2c, b2, ACT_NONE, residual=x2)
        ctx.save_for_backward(x2, pre, hid)
        ctx.W1T, ctx.W2T, ctx.shp = W1T, W2T, shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, pre, hid = ctx.saved_tensors
        d = dy.reshape(-1, dy.shape[-1])
        if not d.is_contiguous():
            d = d.contiguous()
        dW2 = gemm_tn(d, hid)
        db2 = colsum(d)
        dh = gemm_nt(d, ctx.W2T)
        dpre = act_bwd(dh, pre, ACT_GELU)
        del dh
        dW1 = gemm_tn(dpre, x2)
        db1 = colsum(dpre)
        dx = gemm_nt(dpre, ctx.W1T, residual=d)
        return dx.view(ctx.shp), dW1, db1, dW2, db2


def mlp_res(x, W1, b1, W2, b2):
    return MLPResFn.apply(x, W1, b1, W2, b2)


class AttnResFn(torch.autograd.Function):
    """y = xq + out_proj(MHA(q = Wq xq_pos, k = Wk xk_pos, v = Wv xv)) with packed in_proj
    (nn.MultiheadAttention, cross_modal_transformer.py:137-141,145-149,151-156).
    ``self_attn``: xk_pos is xq_pos and xv is xq (one packed projection buffer)."""

    @staticmethod
    def forward(ctx, xq_pos, xq, xk_pos, xv, W_in, b_in, W_o, b_o, H, kbias, self_attn):
        B, Lq, d = xq.shape
        Lk = xv.shape[1]
        dh = d // H
        Wc, WcT = weights.get(W_in, xq.dtype)
        Woc, WoT = weights.get(W_o, xq.dtype)
        a_qp = xq_pos.reshape(B * Lq, d)
        a_q = xq.reshape(B * Lq, d)
        a_kp = a_qp if self_attn else xk_pos.reshape(B * Lk, d)
        a_v = a_q if self_attn else xv.reshape(B * Lk, d)
        if self_attn:
            qkv = torch.empty((B * Lq, 3 * d), dtype=xq.dtype, device=xq.device)
            gemm_nt(a_qp, Wc[:2 * d], b_in[:2 * d], out=qkv[:, :2 * d])
            gemm_nt(a_q, Wc[2 * d:], b_in[2 * d:], out=qkv[:, 2 * d:])
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
        else:
            q = gemm_nt(a_qp, Wc[:d], b_in[:d])
            kv = torch.empty((B * Lk, 2 * d), dtype=xq.dtype, device=xq.device)
            gemm_nt(a_kp, Wc[d:2 * d], b_in[d:2 * d], out=kv[:, :d])
            gemm_nt(a_v, Wc[2 * d:], b_in[2 * d:], out=kv[:, d:])
            k, v = kv[:, :d], kv[:, d:]
        o, lse2 = attn_fwd(q, k, v, B, H, Lq, Lk, dh, kbias)
        y = gemm_nt(o, Woc, b_o, residual=a_q)
        ctx.save_for_backward(a_qp, a_q, a_kp, a_v, q, k, v, o, lse2, kbias)
        ctx.WcT, ctx.WoT, ctx.dims, ctx.self_attn = WcT, WoT, (B, H, Lq, Lk, dh, d), self_attn
        return y.view(B, Lq, d)

    @staticmethod
    def backward(ctx, dy):
        a_qp, a_q, a_kp, a_v, q, k, v, o, lse2, kbias = ctx.saved_tensors
        B, H, Lq, Lk, dh, d = ctx.dims
        WcT, WoT = ctx.WcT, ctx.WoT  # WcT: [d, 3d] ; WoT: [d, d]
        g = dy.reshape(B * Lq, d)
        if not g.is_contiguous():
            g = g.contiguous()
        dWo = gemm_tn(g, o)
        dbo = colsum(g)
        do = gemm_nt(g, WoT)
        dW_in = torch.zeros((3 * d, d), dtype=torch.float32, device=g.device)
        db_in = torch.zeros((3 * d,), dtype=torch.float32, device=g.device)
        if ctx.self_attn:
            dqkv = torch.empty((B * Lq, 3 * d), dtype=g.dtype, device=g.device)
            dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
            attn_bwd(q, k, v, o, do, lse2, B, H, Lq, Lk, dh, dq, dk, dv, kbias)
            gemm_tn(dqkv[:, :2 * d], a_qp, out=dW_in[:2 * d])
            gemm_tn(dv, a_q, out=dW_in[2 * d:])
            colsum(dqkv, out=db_in)
            # d(xq_pos) = [dq dk] W_qk ; d(xq) = dv W_v + dy (residual)
            dxq_pos = gemm_nt(dqkv[:, :2 * d], WcT[:, :2 * d])
            dxq = gemm_nt(dv, WcT[:, 2 * d:], residual=g)
            return (dxq_pos.view(B, Lq, d), dxq.view(B, Lq, d), None, None, dW_in, db_in, dWo, dbo, None, None,
                    None)
        dq = torch.empty((B * Lq, d), dtype=g.dtype, device=g.device)
        dkv = torch.empty((B * Lk, 2 * d), dtype=g.dtype, device=g.device)
        dk, dv = dkv[:, :d], dkv[:, d:]
        attn_bwd(q, k, v, o, do, lse2, B, H, Lq, Lk, dh, dq, dk, dv, kbias)
        gemm_tn(dq, a_qp, out=dW_in[:d])
        gemm_tn(dk, a_kp, out=dW_in[d:2 * d])
        gemm_tn(dv, a_v, out=dW_in[2 * d:])
        colsum(dq, out=db_in[:d])
        colsum(dkv, out=db_in[d:])
        dxq_pos = gemm_nt(dq, WcT[:, :d])
        dxk_pos = gemm_nt(dk, WcT[:, d:2 * d])
        dxv = gemm_nt(dv, WcT[:, 2 * d:])
        return (dxq_pos.view(B, Lq, d), g.view(B, Lq, d), dxk_pos.view(B, Lk, d), dxv.view(B, Lk, d), dW_in, db_in,
                dWo, dbo, None, None, None)


def self_attn_res(x_pos, x, W_in, b_in, W_o, b_o, H):
    return AttnResFn.apply(x_pos, x, None, x, W_in, b_in, W_o, b_o, H, None, True)


def cross_attn_res(xq_pos, xq, xk_pos, xv, W_in, b_in, W_o, b_o, H, kbias):
    return AttnResFn.apply(xq_pos, xq, xk_pos, xv, W_in, b_in, W_o, b_o, H, kbias, False)


class GateFn(torch.autograd.Function):
    """(mem, mem + pos) with mem = LN1(x * (1 + a)), a = head-mean softmax of the 1-query
    sketch->video attention (cross_modal_transformer.py:122-127); u = per-(batch, head) folded
    key projection [B,H,d] fp32."""

    @staticmethod
    def forward(ctx, x, pos, u, gamma, beta, H):
        ctx.set_materialize_grads(False)
        B, L, D = x.shape
        x2 = x.reshape(B * L, D)
        pos2 = pos.reshape(B * L, D)
        u = u.contiguous().float()
        y = torch.empty_like(x2)
        ypos = torch.empty_like(x2)
        a = torch.empty((B * L,), dtype=torch.float32, device=x.device)
        mean = torch.empty_like(a)
        rstd = torch.empty_like(a)
        ws = torch.empty((B * H * (L + 2),), dtype=torch.float32, device=x.device)
        rc = _lib.lib().svol_gate_fwd(_ptr(x2), _ptr(pos2), _ptr(u), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(ypos),
                                      _ptr(a), _ptr(mean), _ptr(rstd), _ptr(ws), B, L, D, H, _dt(x2), _stream())
        _lib.check(rc, 'svol_gate_fwd')
        ctx.save_for_backward(x2, pos2, u, gamma, a, mean, rstd, ws)
        ctx.dims = (B, L, D, H)
        return y.view(B, L, D), ypos.view(B, L, D)

    @staticmethod
    def backward(ctx, dy, dypos):
        x2, pos2, u, gamma, a, mean, rstd, ws = ctx.saved_tensors
        B, L, D, H = ctx.dims
        if dy is None:
            dy, dypos = dypos, None
        dy2 = dy.reshape(B * L, D).contiguous()
        dyp2 = dypos.reshape(B * L, D).contiguous() if dypos is not None else None
        dx = torch.empty_like(x2)
        du = torch.zeros((B, H, D), dtype=torch.float32, device=x2.device)
        dg = torch.zeros((D,), dtype=torch.float32, device=x2.device)
        db = torch.zeros((D,), dtype=torch.float32, device=x2.device)
        ws2 = torch.empty((B * L + B * H,), dtype=torch.float32, device=x2.device)
        rc = _lib.lib().svol_gate_bwd(_ptr(dy2), _ptr(dyp2), _ptr(x2), _ptr(pos2), _ptr(u), _ptr(gamma), _ptr(a),
                                      _ptr(mean), _ptr(rstd), _ptr(ws), _ptr(ws2), _ptr(dx), _ptr(du), _ptr(dg),
                                      _ptr(db), B, L, D, H, _dt(x2), _stream())
        _lib.check(rc, 'svol_gate_bwd')
        return dx.view(B, L, D), None, du, dg, db, None


def gate(x, pos, u, gamma, beta, H):
    return GateFn.apply(x, pos, u, gamma, beta, H)


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
                                      _ptr(g_label), _ptr(g_bbox), _ptr(g_giou), NL, rows, float(eos_coef), _stream())
        _lib.check(rc, 'svol_set_loss')
        ctx.save_for_backward(g_label, g_bbox, g_giou)
        ctx.mark_non_differentiable(match)
        return losses, match

    @staticmethod
    def backward(ctx, dl, _dmatch):
        g_label, g_bbox, g_giou = ctx.saved_tensors
        dl = dl.float()
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
                           packed.n_problems, float(w_bbox), float(w_giou), float(w_class), _stream())
    _lib.check(rc, 'svol_match_cost')
    rc = L.svol_lsap_batched(_ptr(cost), _ptr(packed.cost_off), _ptr(packed.pred_off), _ptr(packed.pred_cnt),
                             _ptr(packed.tgt_off), _ptr(packed.tgt_cnt), _ptr(match), _ptr(packed.status),
                             packed.n_problems, packed.max_dim, _stream())
    _lib.check(rc, 'svol_lsap_batched')
    packed.last_cost = cost
    return match
