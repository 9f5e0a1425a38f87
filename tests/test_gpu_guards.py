"""Red zones around every output and scratch buffer of the kernels that write by computed offsets ("What's weak 1" /
"Next round 5"): the single-pass attention backward shipped in round 4 with fp32 atomics of all-zero partials landing behind its rows
and no parity test could see it.  Every buffer a kernel may write is carved out of ONE arena pre-filled with fp32 -0.0 words, with a
guard zone in FRONT of and BEHIND it: a plain store of anything else changes the pattern, and an fp32 atomic add of +0.0 — the invisible
kind — flips the sign.  The guards must come back bit-identical; the outputs must be finite (i.e. written).  All calls go through the
C-ABI with raw pointers, the way the block programs call it.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

NEG0 = -2147483648  # 0x80000000 as int32: fp32 -0.0


class GuardArena:
    """One int32 allocation of -0.0 words; take() returns tensors of it separated by guard zones (also one in front of the first and
    one behind the last); check() verifies every word outside the taken ranges."""

    def __init__(self, guard_bytes=1 << 18):
        self.guard = guard_bytes // 4
        self.req = []

    def plan(self, name, shape, dtype):
        self.req.append((name, tuple(shape), dtype))

    def build(self):
        off = self.guard
        self.slots = {}
        for name, shape, dtype in self.req:
            n = 1
            for s in shape:
                n *= s
            words = (n * torch.empty((), dtype=dtype).element_size() + 3) // 4
            words = (words + 63) // 64 * 64          # 256-byte aligned starts
            self.slots[name] = (off, words, shape, dtype)
            off += words + self.guard
        self.buf = torch.full((off,), NEG0, dtype=torch.int32, device='cuda')
        out = {}
        for name, (o, w, shape, dtype) in self.slots.items():
            n = 1
            for s in shape:
                n *= s
            out[name] = self.buf[o:o + w].view(dtype)[:n].view(shape)
        return out

    def check(self, what):
        keep = torch.ones_like(self.buf, dtype=torch.bool)
        for name, (o, w, shape, dtype) in self.slots.items():
            n = 1
            for s in shape:
                n *= s
            used = (n * torch.empty((), dtype=dtype).element_size() + 3) // 4
            keep[o:o + used] = False
        bad = (self.buf != NEG0) & keep
        if bool(bad.any()):
            idx = int(torch.nonzero(bad)[0])
            near = [nm for nm, (o, w, _, _) in self.slots.items() if o - self.guard <= idx < o + w + self.guard]
            raise AssertionError(f'{what}: guard word {idx} changed ({int(bad.sum())} words in all); nearest buffer(s): {near}')


def _qkv(B, Lq, Lk, d, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    pm = 1.4426950408889634 / math.sqrt(32)
    q = (torch.randn((B * Lq, d), generator=g) * pm).to(dtype).cuda()
    k = torch.randn((B * Lk, d), generator=g).to(dtype).cuda()
    v = torch.randn((B * Lk, d), generator=g).to(dtype).cuda()
    do = torch.randn((B * Lq, d), generator=g).to(dtype).cuda()
    return q, k, v, do, pm


@pytest.mark.parametrize('B,Lq,Lk,ws,mask', [
    (1, 6272, 6272, True, False),      # single pass, B = 1: the image ends the scratch's first part, a 128-key tail group behind it
    (8, 6272, 6272, True, False),      # the bench launch itself
    (2, 1536, 1536, True, False),      # single pass, no tail group (L % 512 == 0)
    (2, 2048 + 384, 2048 + 384, True, False),   # single pass, 384-key tail (three blocks per wave)
    (2, 1536, 1536, False, False),     # two-pass kernels (no scratch)
    (2, 1000, 1000, True, False),      # ragged length: tile-classified masked kernels (scratch = tile flags)
    (2, 1536, 1536, True, True),       # additive key mask
    (8, 100, 6272, True, True),        # 100 queries against the video: key-split partials + dQ atomics in the scratch
    (8, 100, 6272, True, False),
], ids=['sp-B1-L6272', 'sp-B8-L6272', 'sp-L1536', 'sp-tail384', 'twopass', 'ragged', 'masked', 'ksplit-masked', 'ksplit'])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_attention_backward_stays_inside_its_buffers(B, Lq, Lk, ws, mask, dtype):
    from svol_amd import _lib, ops
    H, dh = 8, 32
    d = H * dh
    lib = _lib.lib()
    q, k, v, do, pm = _qkv(B, Lq, Lk, d, dtype, 17)
    kbias = None
    if mask:
        kbias = torch.zeros((B, Lk), dtype=torch.float32, device='cuda')
        kbias[:, Lk - Lk // 5:] = float('-inf')
        kbias[0, 7] = float('-inf')
    o, lse2 = ops.attn_fwd(q, k, v, B, H, Lq, Lk, dh, kbias, pm)
    need = int(lib.svol_attn_ws_bytes(B, H, Lq, Lk, dh)) if ws else 0
    ar = GuardArena()
    ar.plan('dq', (B * Lq, d), dtype)
    ar.plan('dkv', (B * Lk, 2 * d), dtype)
    ar.plan('delta', (3 * B * H * Lq,), torch.float32)
    if need:
        ar.plan('ws', (need // 4,), torch.float32)
    t = ar.build()
    dq, dkv = t['dq'], t['dkv']
    P = ops._ptr
    rc = lib.svol_attn_bwd(P(q), q.stride(0), P(k), k.stride(0), P(v), v.stride(0), P(o), o.stride(0), P(do), do.stride(0), P(lse2),
                           P(t['delta']), P(kbias) if kbias is not None else None, P(dq), dq.stride(0), P(dkv[:, :d]), dkv.stride(0),
                           P(dkv[:, d:]), dkv.stride(0), B, H, Lq, Lk, dh, 1.0 / math.sqrt(dh), pm, P(t['ws']) if need else None, need,
                           ops._dt(q), ops._stream())
    _lib.check(rc, 'svol_attn_bwd')
    torch.cuda.synchronize()
    ar.check(f'svol_attn_bwd B={B} Lq={Lq} Lk={Lk} ws={ws} mask={mask}')
    assert bool(torch.isfinite(dq.float()).all()) and bool(torch.isfinite(dkv.float()).all()), 'an output was not (fully) written'
    assert float(dq.float().abs().max()) > 0 and float(dkv.float().abs().max()) > 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_attention_forward_stays_inside_its_buffers(dtype):
    """the forward's outputs (o, lse) and its scratch (redo flags / tile classes / key-split partials)."""
    from svol_amd import _lib, ops
    H, dh = 8, 32
    d = H * dh
    lib = _lib.lib()
    for B, Lq, Lk, mask in ((2, 1536, 1536, False), (2, 1000, 1000, False), (8, 100, 6272, True), (1, 6272, 6272, False)):
        q, k, v, _, pm = _qkv(B, Lq, Lk, d, dtype, 23)
        kbias = None
        if mask:
            kbias = torch.zeros((B, Lk), dtype=torch.float32, device='cuda')
            kbias[:, Lk - 1000:] = float('-inf')
        need = int(lib.svol_attn_ws_bytes(B, H, Lq, Lk, dh))
        ar = GuardArena()
        ar.plan('o', (B * Lq, d), dtype)
        ar.plan('lse', (B * H * Lq,), torch.float32)
        ar.plan('ws', (max(need // 4, 1),), torch.float32)
        t = ar.build()
        P = ops._ptr
        rc = lib.svol_attn_fwd(P(q), q.stride(0), P(k), k.stride(0), P(v), v.stride(0), P(t['o']), t['o'].stride(0), P(t['lse']),
                               P(kbias) if kbias is not None else None, B, H, Lq, Lk, dh, 1.0 / math.sqrt(dh), pm, P(t['ws']), need,
                               ops._dt(q), ops._stream())
        _lib.check(rc, 'svol_attn_fwd')
        torch.cuda.synchronize()
        ar.check(f'svol_attn_fwd B={B} Lq={Lq} Lk={Lk}')
        assert bool(torch.isfinite(t['o'].float()).all()) and bool(torch.isfinite(t['lse']).all())


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
def test_grouped_weight_gradient_gemm_stays_inside_its_sinks(dtype):
    """svol_gemm_tn_grouped: split-M workgroups add their tiles with fp32 atomics straight into the gradient sinks, the problem of a
    workgroup is found by a scan over prefix sums — odd sizes, several problems, column sums on and off."""
    from svol_amd import ops
    g = torch.Generator().manual_seed(3)
    M = 50176 if dtype == torch.bfloat16 else 800
    shapes = [(256, 2048, True), (2048, 256, True), (256, 256, False), (256, 512, True), (248, 104, True), (64, 8, False)]   # (N, K multiples of one 16-byte load)
    ar = GuardArena()
    for i, (k_, n_, cs) in enumerate(shapes):
        ar.plan(f'out{i}', (n_, k_), torch.float32)
        if cs:
            ar.plan(f'cs{i}', (n_,), torch.float32)
    t = ar.build()
    probs = []
    refs = []
    for i, (k_, n_, cs) in enumerate(shapes):
        A = (torch.randn((M, n_), generator=g) * 0.1).to(dtype).cuda()      # dY [M, N]
        X = (torch.randn((M, k_), generator=g) * 0.1).to(dtype).cuda()      # X  [M, K]
        out = t[f'out{i}']
        out.zero_()
        c = t[f'cs{i}'] if cs else None
        if c is not None:
            c.zero_()
        probs.append((A, X, out, c))
        refs.append((A, X))
    ops.gemm_tn_grouped(probs)
    torch.cuda.synchronize()
    ar.check('svol_gemm_tn_grouped')
    for (A, X), (_, _, out, c) in zip(refs, probs):
        ref = A.double().t() @ X.double()
        err = float((out.double() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        assert err < 2e-3, err
        if c is not None:
            rc = A.double().sum(0)
            assert float((c.double() - rc).abs().max()) / max(1.0, float(rc.abs().max())) < 2e-3


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_gemm_families_stay_inside_their_outputs(dtype):
    """svol_gemm_nt / svol_gemm_nt_dact through every 16-bit kernel family the step uses — weight-stationary K = 256 (bf16 and fp32 +
    residual outputs, the gelu' copy), the deep-K N = 256 kernel, the 800-row skinny kernels, the fused (dY W2) * aux with its column
    sums — at row counts that are NOT multiples of the 16 / 128-row tiles, with outputs that are column slices of a wider buffer (the
    q | k | v layout): rows past M and columns past the slice must stay untouched."""
    from svol_amd import _lib, ops
    lib = _lib.lib()
    g = torch.Generator().manual_seed(31)
    P = ops._ptr
    D, F = 256, 2048
    for M in (50176 // 8 + 5, 4133, 800):
        x = (torch.randn((M, D), generator=g) * 0.5).to(dtype).cuda()
        hid = (torch.randn((M, F), generator=g) * 0.5).to(dtype).cuda()
        res = torch.randn((M, D), generator=g).cuda()
        W_qk = (torch.randn((2 * D, D), generator=g) * 0.05).to(dtype).cuda()
        W_dd = (torch.randn((D, D), generator=g) * 0.05).to(dtype).cuda()
        W_fd = (torch.randn((F, D), generator=g) * 0.05).to(dtype).cuda()
        W_df = (torch.randn((D, F), generator=g) * 0.05).to(dtype).cuda()
        b = (torch.randn((F,), generator=g) * 0.1).cuda()
        ar = GuardArena()
        ar.plan('qkv', (M, 3 * D), dtype)        # q | k written as a 512-column slice, v as the last 256 columns
        ar.plan('o32', (M, D), torch.float32)
        ar.plan('hid', (M, F), dtype)
        ar.plan('pre', (M, F), dtype)
        ar.plan('y32', (M, D), torch.float32)
        ar.plan('dx', (M, D), dtype)
        ar.plan('dpre', (M, F), dtype)
        ar.plan('cs', (F,), torch.float32)
        t = ar.build()
        t['cs'].zero_()
        qkv = t['qkv']
        dt, s = ops._dt(x), ops._stream()
        calls = [
            ('q|k', lambda: lib.svol_gemm_nt(P(x), D, None, 0, P(W_qk), D, P(qkv), 3 * D, P(b), None, 0, None, 0, None, 0, 0, M, 2 * D, D, dt, s)),
            ('v', lambda: lib.svol_gemm_nt(P(x), D, None, 0, P(W_dd), D, P(qkv[:, 2 * D:]), 3 * D, P(b), None, 0, None, 0, None, 0, 0, M, D, D, dt, s)),
            ('out-proj + residual (fp32)', lambda: lib.svol_gemm_nt(P(x), D, None, 0, P(W_dd), D, P(t['o32']), D, P(b), None, 0, None, 0, P(res), D, 1, M, D, D, dt, s)),
            ("fc1 + gelu'", lambda: lib.svol_gemm_nt(P(x), D, None, 0, P(W_fd), D, P(t['hid']), F, P(b), None, ops.ACT_GELU_D, P(t['pre']), F, None, 0, 0, M, F, D, dt, s)),
            ('fc2 + residual (fp32)', lambda: lib.svol_gemm_nt(P(hid), F, None, 0, P(W_df), F, P(t['y32']), D, P(b), None, 0, None, 0, P(res), D, 1, M, D, F, dt, s)),
            ('dX of the MLP', lambda: lib.svol_gemm_nt(P(hid), F, None, 0, P(W_df), F, P(t['dx']), D, None, None, 0, None, 0, None, 0, 0, M, D, F, dt, s)),
            ("(dY W2) * gelu' + column sums", lambda: lib.svol_gemm_nt_dact(P(x), D, P(W_fd), D, P(t['dpre']), F, P(hid), F, ops.ACT_GELU_D, P(t['cs']), M, F, D, dt, s)),
        ]
        for name, fn in calls:
            _lib.check(fn(), name)
        torch.cuda.synchronize()
        ar.check(f'GEMM families, M = {M}')
        for k_ in ('qkv', 'o32', 'hid', 'pre', 'y32', 'dx', 'dpre', 'cs'):
            assert bool(torch.isfinite(t[k_].float()).all()), k_
        ref = x.double() @ W_dd.double().t() + b[:D].double() + res.double()
        assert float((t['o32'].double() - ref).abs().max()) < 0.05


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16], ids=['bf16', 'fp16'])
def test_layernorm_and_gate_stay_inside_their_buffers(dtype):
    """svol_layernorm_fwd / _bwd and svol_gate_fwd / _bwd: per-wave row loops with computed row offsets, end-of-workgroup atomics into
    [D]-sized gradient sinks; row counts that do not divide by the rows-per-wave split."""
    from svol_amd import _lib, ops
    lib = _lib.lib()
    g = torch.Generator().manual_seed(37)
    P = ops._ptr
    D, H = 256, 8
    for B, L in ((2, 3001), (8, 100)):
        M = B * L
        x32 = torch.randn((M, D), generator=g).cuda()
        pos = torch.randn((M, D), generator=g).to(dtype).cuda()
        gamma = (1.0 + 0.1 * torch.randn((D,), generator=g)).cuda()
        beta = (0.1 * torch.randn((D,), generator=g)).cuda()
        u = (torch.randn((B, H, D), generator=g) * 0.1).cuda()
        dy32 = torch.randn((M, D), generator=g).cuda()
        dy = torch.randn((M, D), generator=g).to(dtype).cuda()
        ar = GuardArena()
        for nm in ('y32', 'dx32', 'gy32', 'gdx32'):
            ar.plan(nm, (M, D), torch.float32)
        for nm in ('y', 'ypos', 'dx', 'gy', 'gypos'):
            ar.plan(nm, (M, D), dtype)
        for nm in ('mean', 'rstd', 'gmean', 'grstd'):
            ar.plan(nm, (M,), torch.float32)
        for nm in ('dgamma', 'dbeta', 'colsum', 'gdgamma', 'gdbeta'):
            ar.plan(nm, (D,), torch.float32)
        ar.plan('a', (B * L,), torch.float32)
        ar.plan('gws', (B * H * (L + 2),), torch.float32)      # exactly what include/svol_hip.h asks for
        ar.plan('gws2', (B * L + B * H,), torch.float32)
        ar.plan('du', (B, H, D), torch.float32)
        t = ar.build()
        for nm in ('dgamma', 'dbeta', 'colsum', 'gdgamma', 'gdbeta', 'du'):
            t[nm].zero_()
        dt, s = ops._dt(pos), ops._stream()
        _lib.check(lib.svol_layernorm_fwd(P(x32), 1, P(gamma), P(beta), P(t['y32']), P(t['y']), P(t['ypos']), P(pos), M, P(t['mean']), P(t['rstd']),
                                          M, D, 0.0, 0, None, dt, s), 'svol_layernorm_fwd')
        _lib.check(lib.svol_layernorm_bwd(P(dy32), P(dy), None, P(x32), 1, P(gamma), P(t['mean']), P(t['rstd']), P(t['dx32']), P(t['dx']),
                                          P(t['dgamma']), P(t['dbeta']), P(t['colsum']), M, D, 0.0, 0, None, dt, s), 'svol_layernorm_bwd')
        _lib.check(lib.svol_gate_fwd(P(x32), P(pos), P(u), P(gamma), P(beta), P(t['gy32']), P(t['gy']), P(t['gypos']), P(t['a']), P(t['gmean']),
                                     P(t['grstd']), P(t['gws']), B, L, D, H, dt, s), 'svol_gate_fwd')
        _lib.check(lib.svol_gate_bwd(P(dy32), P(dy), None, P(x32), P(pos), P(u), P(gamma), P(t['a']), P(t['gmean']), P(t['grstd']), P(t['gws']),
                                     P(t['gws2']), P(t['gdx32']), P(t['du']), P(t['gdgamma']), P(t['gdbeta']), B, L, D, H, dt, s), 'svol_gate_bwd')
        torch.cuda.synchronize()
        ar.check(f'LayerNorm / gate, B = {B}, L = {L}')
        for k_ in ('y32', 'y', 'ypos', 'mean', 'rstd', 'dx32', 'dx', 'dgamma', 'dbeta', 'colsum', 'gy32', 'gy', 'gypos', 'gdx32', 'du', 'gdgamma', 'gdbeta'):
            assert bool(torch.isfinite(t[k_].float()).all()), k_


def test_resnet_training_kernels_stay_inside_their_buffers():
    """csrc/resnet_train.hip + the gathering weight-gradient GEMM: column reductions with end-of-workgroup atomics, gather kernels with
    computed pixel offsets, odd image sizes and row counts that do not divide by the workgroups' row chunks."""
    from svol_amd import _lib, ops
    lib = _lib.lib()
    g = torch.Generator().manual_seed(41)
    P = ops._ptr
    dt = torch.bfloat16
    for (n, H, W, C, Cout, k, s, p) in [(3, 9, 11, 16, 32, 3, 2, 1), (2, 15, 15, 64, 64, 3, 1, 1), (5, 7, 7, 128, 256, 1, 2, 0)]:
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        M, Mi = n * Ho * Wo, n * H * W
        K = k * k * C
        Kp = (K + 31) // 32 * 32
        x = torch.randn((Mi, C), generator=g).to(dt).cuda()
        z = torch.randn((M, Cout), generator=g).to(dt).cuda()
        dy = torch.randn((M, Cout), generator=g).to(dt).cuda()
        dcols = torch.randn((M, Kp), generator=g).to(dt).cuda()
        gamma = torch.ones(Cout).cuda()
        ar = GuardArena()
        ar.plan('stats', (8, Cout), torch.float32)
        ar.plan('y', (M, Cout), dt)
        ar.plan('sums', (2, Cout), torch.float32)
        ar.plan('dz', (M, Cout), dt)
        ar.plan('dres', (M, Cout), dt)
        ar.plan('dx', (Mi, C), dt)
        ar.plan('dwp', (Cout, Kp), torch.float32)
        ar.plan('pool', (M, C), dt)
        ar.plan('idx', (M, C), torch.uint8)
        ar.plan('dpool', (Mi, C), dt)
        t = ar.build()
        for nm in ('stats', 'sums', 'dwp'):
            t[nm].zero_()
        st, d_, s_ = t['stats'], ops._dt(x), ops._stream()
        _lib.check(lib.svol_bn_colstats(P(z), None, 0.0, P(st[0]), P(st[1]), M, Cout, d_, s_), 'colstats')
        _lib.check(lib.svol_bn_colstats(P(z), P(st[0]), 1.0 / M, P(st[2]), P(st[3]), M, Cout, d_, s_), 'colstats')
        _lib.check(lib.svol_bn_finalize(P(st[0]), P(st[3]), None, P(gamma), P(gamma), None, None, 0.1, 1e-5, M, Cout, P(st[4]), P(st[5]), P(st[6]), P(st[7]), s_), 'finalize')
        _lib.check(lib.svol_bn_apply(P(z), P(st[6]), P(st[7]), P(dy), 1, P(t['y']), M, Cout, d_, s_), 'apply')
        _lib.check(lib.svol_bn_bwd_reduce(P(dy), P(t['y']), P(z), P(st[4]), P(st[5]), P(t['sums'][0]), P(t['sums'][1]), M, Cout, d_, s_), 'bwd_reduce')
        _lib.check(lib.svol_bn_bwd_apply(P(dy), P(t['y']), P(z), P(st[4]), P(st[5]), P(gamma), P(t['sums'][0]), P(t['sums'][1]), P(t['dz']), P(t['dres']), M,
                                         Cout, d_, s_), 'bwd_apply')
        _lib.check(lib.svol_col2im_nhwc(P(dcols), Kp, P(t['dx']), n, H, W, C, k, k, s, p, d_, s_), 'col2im')
        _lib.check(lib.svol_conv_wgrad_nhwc(P(t['dz']), P(x), P(t['dwp']), n, H, W, C, Cout, k, k, s, p, Kp, d_, s_), 'conv_wgrad')
        if k == 3:
            _lib.check(lib.svol_maxpool_idx_nhwc(P(x), P(t['pool']), P(t['idx']), n, H, W, C, k, s, p, d_, s_), 'maxpool')
            _lib.check(lib.svol_maxpool_bwd_nhwc(P(t['pool']), P(t['idx']), P(t['dpool']), n, H, W, C, k, s, p, d_, s_), 'maxpool_bwd')
        torch.cuda.synchronize()
        ar.check(f'resnet training kernels {(n, H, W, C, Cout, k, s, p)}')
        for nm in ('y', 'dz', 'dres', 'dx', 'dwp') + (('pool', 'dpool') if k == 3 else ()):
            assert bool(torch.isfinite(t[nm].float()).all()), nm
