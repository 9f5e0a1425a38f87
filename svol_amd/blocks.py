"""Host side of the composite block programs (include/svol_hip.h: svol_video_half_* / svol_query_self_* / svol_query_cross_*).

One ``torch.autograd.Function`` and ONE C call per block and direction of a ``CrossModalTransformerLayer`` (reference
cross_modal_transformer.py:105-160) instead of one Function per op and one ctypes call + ``torch.empty`` per kernel:
the per-op host path needed 19.6 ms to issue a 20.8 ms step.  The arithmetic is the same kernels in the same
order as ``ops.GateFn`` / ``ops.AttnLNFn`` / ``ops.MLPLNFn`` (which stay for the enc/dec heads and as the A/B reference,
``SVOL_NO_BLOCKS=1``); what this module owns is

* the slot tables (one ``void*`` array per block call: parameters and weight copies are filled once per plan, activations per call),
* the arenas: everything a block saves for backward + its outputs live in ONE allocation, backward temporaries in another,
* the weight-gradient deferral (``ops.gemm_tn_sink``'s policy, one queue item per block instead of one per GEMM).

PyTorch remains plumbing: device memory, streams, the autograd tape.
"""
from __future__ import annotations

import ctypes
import math
import os
import weakref

import torch

from . import _lib, ops

WGRAD_SPLIT = os.environ.get('SVOL_WGRAD_SPLIT') is not None   # measured: +0.16 ms/step (more work beside the attention backward than the shorter tail saves)
ENABLED = os.environ.get('SVOL_NO_BLOCKS') is None
# SVOL_GATE_SCORES_FUSE=1: layer i's LN3 also writes layer i + 1's gate scores (svol_layernorm_gate_scores_fwd) and that layer skips its
# own score pass — same bits (tests/test_gpu_head.py).  OFF by default: alone the pair of launches is 25 us per layer shorter
# (LN3 28.5 -> 33.3 us, gate 72.5 -> 42.6 us), in the step it is 0.1-0.2 ms per step SLOWER on three alternating same-box pairs
# (17.35-17.45 against 17.25-17.30 ms): the query stream's block, which used to run beside the gate and the projections, slides
# under the attention forward (430 -> 446 us per layer in the block trace) — profiles/round6_summary.md.
GATE_SCORES_FUSE = os.environ.get('SVOL_GATE_SCORES_FUSE') is not None
VH, QS, QC = 0, 1, 2
_F32, _BF16 = 0, 1
_ESZ = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2}
_SLOTS = {}


def _slots(block):
    """name -> index of a block's slot table, from the library itself (one source of truth: the X-macro lists of svol_hip.h)."""
    m = _SLOTS.get(block)
    if m is None:
        names = _lib.lib().svol_block_slot_names(block).decode().rstrip(',').split(',')
        m = _SLOTS[block] = {n: i for i, n in enumerate(names)}
    return m


class _Table:
    """a ``void*[]`` for one block call; ``template`` rows (parameters, weight copies, gradient targets) are copied in bulk."""

    __slots__ = ('arr', 'idx', 'n')

    def __init__(self, block, template=None):
        self.idx = _slots(block)
        self.n = len(self.idx)
        self.arr = (ctypes.c_void_p * self.n)()
        if template is not None:
            ctypes.memmove(self.arr, template.arr, self.n * 8)

    def set(self, name, ptr):
        self.arr[self.idx[name]] = ptr

    def set_t(self, name, t):
        self.arr[self.idx[name]] = t.data_ptr() if t is not None else None


def _align(n, a=256):
    return (n + a - 1) // a * a


class _Layout:
    """named byte ranges inside one arena allocation."""

    def __init__(self, items):
        self.off = {}
        n = 0
        for name, nbytes in items:
            self.off[name] = (n, nbytes)
            n += _align(nbytes)
        self.total = max(n, 256)

    def fill(self, tbl, base):
        for name, (o, _) in self.off.items():
            tbl.arr[tbl.idx[name]] = base + o

    def view(self, arena, name, dtype, shape):
        o, nb = self.off[name]
        return arena[o:o + nb].view(dtype).view(shape)


_LAYOUTS = {}


def _layout(key, make):
    lay = _LAYOUTS.get(key)
    if lay is None:
        lay = _LAYOUTS[key] = _Layout(make())
    return lay


_SCRATCH = {}


def _scratch(dev, nbytes):
    """per-(device, stream) scratch that only lives inside one block call (fp32 LayerNorm outputs used as the next GEMM's
    residual, attention workspace): stream order makes reuse across layers safe."""
    key = (dev, ops._stream())
    t = _SCRATCH.get(key)
    if t is None or t.numel() < nbytes:
        t = _SCRATCH[key] = torch.empty((_align(nbytes, 1 << 20),), dtype=torch.uint8, device=dev)
    return t


def _dims(B, L, N, D, H, F, dt, qdt, ws_bytes):
    return (ctypes.c_int64 * 9)(B, L, N, D, H, F, ops._DT[dt], ops._DT[qdt], ws_bytes)


def _attn_ws_bytes(B, H, Lq, Lk, dh, dt):
    n = _lib.lib().svol_attn_ws_bytes(B, H, Lq, Lk, dh) if dt != torch.float32 else 0
    return max(int(n), 0)


# ----------------------------------------------------------------------------------------------------------------------
# plans: everything about a layer that does not change from step to step
# ----------------------------------------------------------------------------------------------------------------------
_VH_PARAMS = [  # (grad slot, parameter getter) in the order the Function takes / returns them
    ('DG1', lambda l: l.norm1.weight), ('DBT1', lambda l: l.norm1.bias),
    ('DW_IN', lambda l: l.content_self_attn.in_proj_weight), ('DB_IN', lambda l: l.content_self_attn.in_proj_bias),
    ('DW_O', lambda l: l.content_self_attn.out_proj.weight), ('DB_O', lambda l: l.content_self_attn.out_proj.bias),
    ('DG2', lambda l: l.norm2.weight), ('DBT2', lambda l: l.norm2.bias),
    ('DW_FC1', lambda l: l.mlp1.fc1.weight), ('DB_FC1', lambda l: l.mlp1.fc1.bias),
    ('DW_FC2', lambda l: l.mlp1.fc2.weight), ('DB_FC2', lambda l: l.mlp1.fc2.bias),
    ('DG3', lambda l: l.norm3.weight), ('DBT3', lambda l: l.norm3.bias)]
_QS_PARAMS = [
    ('DW_IN', lambda l: l.token_self_attn.in_proj_weight), ('DB_IN', lambda l: l.token_self_attn.in_proj_bias),
    ('DW_O', lambda l: l.token_self_attn.out_proj.weight), ('DB_O', lambda l: l.token_self_attn.out_proj.bias),
    ('DG4', lambda l: l.norm4.weight), ('DBT4', lambda l: l.norm4.bias)]
_QC_PARAMS = [
    ('DW_IN', lambda l: l.content_token_cross_attn.in_proj_weight), ('DB_IN', lambda l: l.content_token_cross_attn.in_proj_bias),
    ('DW_O', lambda l: l.content_token_cross_attn.out_proj.weight), ('DB_O', lambda l: l.content_token_cross_attn.out_proj.bias),
    ('DG5', lambda l: l.norm5.weight), ('DBT5', lambda l: l.norm5.bias),
    ('DW_FC1', lambda l: l.mlp2.fc1.weight), ('DB_FC1', lambda l: l.mlp2.fc1.bias),
    ('DW_FC2', lambda l: l.mlp2.fc2.weight), ('DB_FC2', lambda l: l.mlp2.fc2.bias),
    ('DG6', lambda l: l.norm6.weight), ('DBT6', lambda l: l.norm6.bias)]
_PARAMS = {VH: _VH_PARAMS, QS: _QS_PARAMS, QC: _QC_PARAMS}


class BlockPlan:
    """static rows of one block's slot table for one layer: parameter pointers, compute-dtype weight copies (ops.weights keeps them
    alive and refreshes them once per forward), gradient sinks.  ``key`` changes when any of those pointers does."""

    def __init__(self, block, layer, dt, qdt):
        self.block, self.dt, self.qdt = block, dt, qdt
        self.layer = weakref.ref(layer)
        self.params = [g(layer) for _, g in _PARAMS[block]]
        self.grad_slots = [n for n, _ in _PARAMS[block]]
        self.tpl = _Table(block)
        self.keep = []   # tensors whose pointers sit in the template
        D = layer.d_model
        self.split_v = ops.SPLIT_V and D % 32 == 0
        t = self.tpl
        W = ops.weights
        dev = self.params[0].device

        self.casts = []    # (weight, dtype, wc, wt) / (weight, row0, rows, ws): the copies whose POINTERS sit in the template
        self.epoch = W.epoch

        def both(name, w, dtype, name_t=None):
            wc, wt = W.get(w, dtype)
            t.set_t(name, wc)
            t.set_t(name_t or (name + '_T'), wt)
            self.keep += [wc, wt]
            self.casts.append((w, dtype, wc, wt))

        def qscale(dtype, n):
            if dtype == torch.float32:
                return None
            q = ops._qscale(D, ops.LOG2E / math.sqrt(D // layer.nhead), dev)
            self.keep.append(q)
            return q

        if block == VH:
            m, mlp = layer.content_self_attn, layer.mlp1
            for n_, p_ in (('G1', layer.norm1.weight), ('BT1', layer.norm1.bias), ('G2', layer.norm2.weight), ('BT2', layer.norm2.bias),
                           ('G3', layer.norm3.weight), ('BT3', layer.norm3.bias), ('B_IN', m.in_proj_bias), ('B_O', m.out_proj.bias),
                           ('B_FC1', mlp.fc1.bias), ('B_FC2', mlp.fc2.bias)):
                t.set_t(n_, p_)
            both('W_IN', m.in_proj_weight, dt)
            both('W_O', m.out_proj.weight, dt)
            both('W_FC1', mlp.fc1.weight, dt)
            both('W_FC2', mlp.fc2.weight, dt)
            if dt == torch.bfloat16 and self.split_v:
                hl = W.get_split(m.in_proj_weight, 2 * D, D)
                t.set_t('WV_HILO', hl)
                self.keep.append(hl)
                self.casts.append((m.in_proj_weight, 2 * D, D, hl))
            t.set_t('QSCALE', qscale(dt, 2 * D))
        elif block == QS:
            m = layer.token_self_attn
            for n_, p_ in (('G4', layer.norm4.weight), ('BT4', layer.norm4.bias), ('B_IN', m.in_proj_bias), ('B_O', m.out_proj.bias)):
                t.set_t(n_, p_)
            both('W_IN', m.in_proj_weight, qdt)
            both('W_O', m.out_proj.weight, qdt)
            if qdt == torch.bfloat16 and self.split_v:
                hl = W.get_split(m.in_proj_weight, 2 * D, D)
                t.set_t('WV_HILO', hl)
                self.keep.append(hl)
                self.casts.append((m.in_proj_weight, 2 * D, D, hl))
            t.set_t('QSCALE', qscale(qdt, 2 * D))
        else:
            m, mlp = layer.content_token_cross_attn, layer.mlp2
            for n_, p_ in (('G5', layer.norm5.weight), ('BT5', layer.norm5.bias), ('G6', layer.norm6.weight), ('BT6', layer.norm6.bias),
                           ('B_IN', m.in_proj_bias), ('B_O', m.out_proj.bias), ('B_FC1', mlp.fc1.bias), ('B_FC2', mlp.fc2.bias)):
                t.set_t(n_, p_)
            both('W_INQ', m.in_proj_weight, qdt)
            both('W_KV', m.in_proj_weight, dt)
            both('W_O', m.out_proj.weight, qdt)
            both('W_FC1', mlp.fc1.weight, qdt)
            both('W_FC2', mlp.fc2.weight, qdt)
            if dt == torch.bfloat16 and self.split_v:
                hl = W.get_split(m.in_proj_weight, 2 * D, D)
                t.set_t('WV_HILO', hl)
                self.keep.append(hl)
                self.casts.append((m.in_proj_weight, 2 * D, D, hl))
            t.set_t('QSCALE', qscale(dt, D))   # the attention core's dtype decides (ops.AttnLNFn: premul follows dkv)
        # gradient targets: the reducer's bucket views where there are sinks
        self.sinks = [ops._claim(p, p.requires_grad) for p in self.params]
        self.nosink = [(i, p.numel()) for i, (p, s) in enumerate(zip(self.params, self.sinks)) if s is None]
        for name, s in zip(self.grad_slots, self.sinks):
            if s is not None:
                t.set_t(name, s.view)
        # every parameter has a sink owned by a reducer: the Function then takes NO parameter inputs — no AccumulateGrad nodes, no
        # per-parameter hooks (150 of each per step) — and tells the reducer itself, once per bucket, when the block's gradients
        # are enqueued
        self.graph_params = self.params if (self.nosink or any(s.owner is None for s in self.sinks)) else []
        cnt = {}
        for s in self.sinks:
            if s is not None and s.owner is not None:
                cnt[(id(s.owner), s.bucket)] = (s.owner, s.bucket, cnt.get((id(s.owner), s.bucket), (None, None, 0))[2] + 1)
        self.notify = list(cnt.values())
        self.key = self._key()

    def _key(self):
        k = []
        for p in self.params:
            g = p.grad
            s = getattr(p, '_svol_sink', None)
            k.append((p.data_ptr(), g.data_ptr() if (g is not None and s is not None) else 0, p.requires_grad))
        return (ops.SPLIT_V, tuple(k))

    def valid(self):
        return self.layer() is not None and self._key() == self.key

    def fresh_copies(self):
        """The compute-dtype / transposed / split weight copies behind the template's pointers are those of the CURRENT parameter
        values: the per-op path re-validated every copy on every call (version counter + cache epoch, ops._WeightCache.get),
        a plan caches raw pointers.  Same check here, once per use: a hit costs a dictionary lookup; a copy that an in-place update
        made stale (a torch optimizer step, load_state_dict, copy_ — anything that bumps ``_version`` — or a new cache epoch) is
        re-cast INTO ITS BUFFER, so the template stays valid; a copy that moved invalidates the plan."""
        W = ops.weights
        for c in self.casts:
            if len(c) == 4 and isinstance(c[1], torch.dtype):
                wc, wt = W.get(c[0], c[1])
                if wc is not c[2] or wt is not c[3]:
                    return False
            else:
                if W.get_split(c[0], c[1], c[2]) is not c[3]:
                    return False
        return True

    def grad_buffers(self, tbl, dev):
        """parameters without a sink: one zeroed flat buffer, views returned to autograd."""
        if not self.nosink:
            return None
        offs, n = [], 0
        for _, numel in self.nosink:
            offs.append(n)
            n += (numel + 3) // 4 * 4
        flat = torch.zeros((n,), dtype=torch.float32, device=dev)
        out = {}
        base = flat.data_ptr()
        for (i, numel), o in zip(self.nosink, offs):
            tbl.set(self.grad_slots[i], base + o * 4)
            out[i] = flat[o:o + numel].view_as(self.params[i])
        return out

    def inputs(self, *tensors):
        """the parameter inputs of this call's Function: none when every gradient goes to a reducer-owned sink and backward is
        certain to run through a tensor input."""
        if self.graph_params or not any(t_.requires_grad for t_ in tensors):
            return self.params
        return ()

    def done(self, in_graph):
        """the block's parameter gradients are enqueued / queued: tell the reducer (only when the parameters are not in the graph —
        otherwise their post-accumulate hooks do)."""
        if not in_graph:
            for owner, bi, n in self.notify:
                owner.params_done(bi, n)

    def grads_out(self, bufs, needs, in_graph):
        """what the Function returns for its parameter inputs."""
        res = []
        if not in_graph:
            return res
        for i, p in enumerate(self.params):
            if bufs is not None and i in bufs and needs[i]:
                res.append(bufs[i])
            else:
                res.append(None)
        return res


_PLANS = weakref.WeakKeyDictionary()


def plan(layer, block, dt, qdt):
    per = _PLANS.get(layer)
    if per is None:
        per = _PLANS[layer] = {}
    k = (block, dt, qdt)
    pl = per.get(k)
    if pl is None or not pl.valid() or not pl.fresh_copies():
        pl = per[k] = BlockPlan(block, layer, dt, qdt)
    return pl


# ----------------------------------------------------------------------------------------------------------------------
# weight-gradient deferral (policy of ops.gemm_tn_sink, one queue item per block)
# ----------------------------------------------------------------------------------------------------------------------
def _issue_wgrad(fn_name, dims, tbl, keep, dev, part=None):
    """the block's weight-gradient GEMMs: on the weight-gradient stream behind an event when every target is a sink, else in line."""
    fn = getattr(_lib.lib(), fn_name)

    def run(stream_handle):
        if part is None:
            _lib.check(fn(dims, tbl.arr, stream_handle), fn_name)
        else:
            _lib.check(fn(dims, tbl.arr, part, stream_handle), fn_name)

    if not ops.WGRAD_ASYNC or (not ops.WGRAD_IN_CAPTURE and torch.cuda.is_current_stream_capturing()):
        run(ops._stream())
        return
    cur = ops._current_stream_obj()
    ev = torch.cuda.Event()
    ev.record(cur)

    def later():
        ws = ops._wgrad_stream(dev)
        ws.wait_event(ev)
        run(ws.cuda_stream)
        for t in keep:
            t.record_stream(ws)

    if ops.WGRAD_DEFER and ops._BIG_ATTN['left'] > 0 and not torch.cuda.is_current_stream_capturing():
        ops._WGRAD_PENDING.append(later)
        ops.arm_wgrad_flush()
    else:
        later()


def _maybe_events(name, meta):
    """bench.py's live roofline: a pair of timing events the C side records around the block's large attention launch."""
    tm = ops.timer
    if not tm.enabled or name not in tm.names:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()   # materialises the handles (torch creates the hipEvent at the first record)
    e1.record()
    tm.records.setdefault((name, meta), []).append((e0, e1))
    return e0, e1


# ----------------------------------------------------------------------------------------------------------------------
# the single-pass attention backward's fp32 dQ image, zeroed off the critical path
# ----------------------------------------------------------------------------------------------------------------------
# The key-stationary kernel adds into a zeroed [B, L, H, 32] fp32 image (51 MB at cfg2).  svol_attn_bwd zeroes it in its prologue
# kernel — a third of that kernel's bytes, on the video stream, in front of an issue-bound launch that leaves the memory side idle
# for 0.86 ms.  With TWO workspaces alternating between the layers, layer i's program zeroes the other one on a side stream BEHIND
# its own prologue, i.e. beside its long kernel, and layer i-1 finds it clean (include/svol_hip.h: svol_attn_bwd_ex, the
# EV_CLEAN_IN / EV_PREP / ATTN_WS_NEXT / ZERO_STREAM / EV_CLEAN_OUT slots).  The last layer of a step cleans the first one's of the
# next step.  Same values either way: only the zero fill moves.
# MEASURED (round 6, same box): the prologue 35.8 -> 13.6 us, the zero launch 12.8 us beside the long kernel; the launch group in the
# step 915 -> 895 us with a 64-workgroup fill (2048 workgroups: +10 us — a wide grid takes dispatch slots from the long kernel's first
# round) — and the STEP does not move: 17.04-17.11 ms with it against 17.02-17.05 without (four alternating pairs).  The video stream's
# chain gets shorter, the step does not: outside the attention launches the three streams are bound by memory THROUGHPUT together, and
# 0.3 GB of 46 GB per step moved under the attention kernel is below what the two extra event packets per layer on the video stream
# cost.  Off by default; SVOL_DQ_PREZERO=1 turns it on.
DQ_PREZERO = os.environ.get('SVOL_DQ_PREZERO') is not None and os.environ.get('SVOL_NO_DQ_PREZERO') is None
_DQ_PREZERO = {}


class _DqPrezero:
    __slots__ = ('bufs', 'side', 'ev_prep', 'clean_ev', 'clean', 'nxt')

    def __init__(self, dev, wsb):
        self.bufs = [torch.empty((_align(wsb, 1 << 20),), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev)
        self.ev_prep = torch.cuda.Event()
        self.clean_ev = [torch.cuda.Event(), torch.cuda.Event()]
        for e_ in [self.ev_prep] + self.clean_ev:
            e_.record()   # materialises the handle (torch creates the hipEvent at the first record)
        self.clean = [False, False]
        self.nxt = 0

    def bind(self, tbl):
        i = self.nxt
        tbl.set('ATTN_WS', self.bufs[i].data_ptr())
        tbl.set('EV_CLEAN_IN', self.clean_ev[i].cuda_event if self.clean[i] else None)
        tbl.set('EV_PREP', self.ev_prep.cuda_event)
        tbl.set('ATTN_WS_NEXT', self.bufs[1 - i].data_ptr())
        tbl.set('ZERO_STREAM', self.side.cuda_stream)
        tbl.set('EV_CLEAN_OUT', self.clean_ev[1 - i].cuda_event)
        self.clean[i] = False   # whatever happens from here on, this image is no longer known to be zero

    def advance(self, tbl):
        i = self.nxt
        self.clean[1 - i] = True
        self.nxt = 1 - i
        for n_ in ('EV_CLEAN_IN', 'EV_PREP', 'ATTN_WS_NEXT', 'ZERO_STREAM', 'EV_CLEAN_OUT'):
            tbl.set(n_, None)


def _dq_prezero(dev, B, H, L, dh, dt, wsb):
    """the two-workspace state for this launch shape, or None (switch off, stream capture, a shape the single pass does not serve)."""
    if not DQ_PREZERO or dt == torch.float32 or torch.cuda.is_current_stream_capturing():
        return None
    key = (dev, ops._stream(), B, H, L, dh, dt, wsb)
    st = _DQ_PREZERO.get(key)
    if st is None:
        img = _lib.lib().svol_attn_bwd_sp_image_bytes(B, H, L, L, dh, wsb, ops._DT[dt])
        st = _DQ_PREZERO[key] = _DqPrezero(dev, wsb) if img > 0 else False
    return st or None


# ----------------------------------------------------------------------------------------------------------------------
# video half
# ----------------------------------------------------------------------------------------------------------------------
def _vh_fwd_layout(B, L, D, H, F, e):
    M = B * L
    return [('Y1', M * D * e), ('Y1POS', M * D * e), ('A', M * 4), ('MEAN1', M * 4), ('RSTD1', M * 4), ('GATE_WS', B * H * (L + 2) * 4),
            ('QKV', M * 3 * D * e), ('O', M * D * e), ('LSE', B * H * L * 4), ('S2', M * D * 4), ('Y2', M * D * e), ('MEAN2', M * 4),
            ('RSTD2', M * 4), ('PRE', M * F * e), ('HID', M * F * e), ('S3', M * D * 4), ('MEAN3', M * 4), ('RSTD3', M * 4),
            ('M32', M * D * 4), ('M', M * D * e), ('MPOS', M * D * e)]


def _vh_bwd_layout(B, L, D, H, F, e):
    M = B * L
    return [('DS32_3', M * D * 4), ('DS3', M * D * e), ('DPRE', M * F * e), ('DY2', M * D * e), ('DS32_2', M * D * 4), ('G2D', M * D * e),
            ('DO', M * D * e), ('DQKV', M * 3 * D * e), ('DELTA', 3 * B * H * L * 4), ('DXQP', M * D * e), ('DXQ', M * D * e),
            ('GATE_WS2', (B * L + B * H) * 4), ('DX32', M * D * 4)]


class VideoHalfFn(torch.autograd.Function):
    """(m32, m, m + pos) = LN3(. + MLP1(.)) o LN2(. + SelfAttn(.)) o LN1(gate(x32))  — cross_modal_transformer.py:122-143."""

    @staticmethod
    def forward(ctx, pl, x32, pos, u, u_next, sc_in, *params):
        """u_next: the NEXT layer's gate vectors or None; sc_in: the gate workspace the layer before filled with THIS layer's scores
        (its fourth output) or None.  Fourth output: the next layer's gate workspace (empty when u_next is None) — plain storage,
        not differentiable: the next layer's gate backward differentiates the scores through ITS x32, as it always did."""
        ctx.set_materialize_grads(False)
        B, L, D = x32.shape
        layer = pl.layer()
        H, F = layer.nhead, layer.mlp1.fc1.weight.shape[0]
        dt = pl.dt
        e = _ESZ[dt]
        dev = x32.device
        M = B * L
        x2 = x32.reshape(M, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        pos2 = pos.reshape(M, D)
        pos2 = pos2 if pos2.is_contiguous() else pos2.contiguous()
        u = u.contiguous()
        assert x2.dtype == torch.float32 and pos2.dtype == dt and u.dtype == torch.float32
        wsb = _attn_ws_bytes(B, H, L, L, D // H, dt)
        key = (VH, B, L, D, H, F, dt)
        lay = _layout(('f',) + key, lambda: _vh_fwd_layout(B, L, D, H, F, e))
        slay = _layout(('s',) + key, lambda: [('Y1_32', M * D * 4), ('Y2_32', M * D * 4), ('ATTN_WS', max(wsb, 16))])
        arena = torch.empty((lay.total,), dtype=torch.uint8, device=dev)
        scr = _scratch(dev, slay.total)
        tbl = _Table(VH, pl.tpl)
        lay.fill(tbl, arena.data_ptr())
        slay.fill(tbl, scr.data_ptr())
        tbl.set('X32', x2.data_ptr())
        tbl.set('POS', pos2.data_ptr())
        tbl.set('U', u.data_ptr())
        if sc_in is not None:
            assert sc_in.dtype == torch.float32 and sc_in.numel() == B * H * (L + 2) and sc_in.is_contiguous()
            tbl.set('GATE_WS', sc_in.data_ptr())
            tbl.set('GATE_PRE', sc_in.data_ptr())
        if u_next is not None:
            u_next = u_next.contiguous()
            assert u_next.dtype == torch.float32 and u_next.shape == u.shape and L % 4 == 0
            sc_next = torch.empty((B * H * (L + 2),), dtype=torch.float32, device=dev)
            tbl.set('U_NEXT', u_next.data_ptr())
            tbl.set('GATE_WS_NEXT', sc_next.data_ptr())
        else:
            sc_next = torch.empty((0,), dtype=torch.float32, device=dev)
        dims = _dims(B, L, 0, D, H, F, dt, dt, wsb)
        big = B * H * L * L >= ops._WGRAD_FLUSH_MIN_SCORES
        if big:
            ops._BIG_ATTN['left'] += 1
        ev = _maybe_events('attn_fwd', (B, H, L, L, D // H))
        if ev:
            tbl.set('EV_A0', ev[0].cuda_event)
            tbl.set('EV_A1', ev[1].cuda_event)
        _lib.check(_lib.lib().svol_video_half_fwd(dims, tbl.arr, ops._stream()), 'svol_video_half_fwd')
        tbl.set('EV_A0', None)
        tbl.set('EV_A1', None)
        tbl.set('U_NEXT', None)
        tbl.set('GATE_WS_NEXT', None)
        ctx.pl, ctx.tbl, ctx.dims, ctx.arena, ctx.shape, ctx.big = pl, tbl, dims, arena, (B, L, D, H, F), big
        ctx.n_par = len(params)
        ctx.sc_in = sc_in   # (GATE_WS of the backward; statistics are written behind the scores: not a saved TENSOR, its version moves)
        ctx.save_for_backward(x2, pos2, u)
        ctx.mark_non_differentiable(sc_next)
        return (lay.view(arena, 'M32', torch.float32, (B, L, D)), lay.view(arena, 'M', dt, (B, L, D)),
                lay.view(arena, 'MPOS', dt, (B, L, D)), sc_next)

    @staticmethod
    def backward(ctx, dm32, dm, dmpos, _dsc=None):
        pl, tbl, dims = ctx.pl, ctx.tbl, ctx.dims
        B, L, D, H, F = ctx.shape
        x2, pos2, u = ctx.saved_tensors
        n_par = ctx.n_par
        if dm32 is None and dm is None and dmpos is None:
            pl.done(n_par > 0)
            return (None,) * (6 + n_par)
        dt = pl.dt
        e = _ESZ[dt]
        dev = x2.device
        M = B * L
        cont = lambda t_: None if t_ is None else (t_.reshape(M, D) if t_.is_contiguous() else t_.reshape(M, D).contiguous())
        dm32, dm, dmpos = cont(dm32), cont(dm), cont(dmpos)
        lay = _layout(('b', VH, B, L, D, H, F, dt), lambda: _vh_bwd_layout(B, L, D, H, F, e))
        tmp = torch.empty((lay.total,), dtype=torch.uint8, device=dev)
        lay.fill(tbl, tmp.data_ptr())
        du = torch.zeros((B, H, D), dtype=torch.float32, device=dev)
        tbl.set('DU', du.data_ptr())
        tbl.set_t('DM32', dm32)
        tbl.set_t('DM', dm)
        tbl.set_t('DMPOS', dmpos)
        slay = _layout(('s', VH, B, L, D, H, F, dt), None)
        scr = _scratch(dev, slay.total)
        slay.fill(tbl, scr.data_ptr())
        bufs = pl.grad_buffers(tbl, dev)
        L_ = _lib.lib()
        s = ops._stream()
        _lib.check(L_.svol_video_half_bwd(dims, tbl.arr, 1, s), 'svol_video_half_bwd')
        keep = [t_ for t_ in (ctx.arena, tmp, dm32, dm, dmpos) if t_ is not None]
        split = bufs is None and ctx.big and WGRAD_SPLIT
        if ctx.big:
            ops._BIG_ATTN['left'] -= 1
            if split:   # this layer's MLP / out-proj weight gradients exist already: they go beside ITS attention backward
                _issue_wgrad('svol_video_half_wgrad_part', dims, tbl, keep, dev, part=1)
            ops.flush_wgrad(gate=True)   # queued weight-gradient GEMMs run beside the attention backward (issue-bound; they are HBM / atomic bound)
        ev = _maybe_events('attn_bwd', (B, H, L, L, D // H))
        if ev:
            tbl.set('EV_A0', ev[0].cuda_event)
            tbl.set('EV_A1', ev[1].cuda_event)
        pz = _dq_prezero(dev, B, H, L, D // H, dt, dims[8]) if ctx.big else None
        if pz is not None:
            pz.bind(tbl)
        _lib.check(L_.svol_video_half_bwd(dims, tbl.arr, 2, s), 'svol_video_half_bwd')
        if pz is not None:
            pz.advance(tbl)
        tbl.set('EV_A0', None)
        tbl.set('EV_A1', None)
        if split:
            _issue_wgrad('svol_video_half_wgrad_part', dims, tbl, keep, dev, part=2)
        elif bufs is None:
            _issue_wgrad('svol_video_half_wgrad', dims, tbl, keep, dev)
        else:
            _lib.check(L_.svol_video_half_wgrad(dims, tbl.arr, s), 'svol_video_half_wgrad')
        pl.done(n_par > 0)
        dx32 = lay.view(tmp, 'DX32', torch.float32, (B, L, D))
        needs = ctx.needs_input_grad
        return (None, dx32, None, du if needs[3] else None, None, None) + tuple(pl.grads_out(bufs, needs[6:], n_par > 0))


def gate_scores_fusable(L, D, H):
    """svol_layernorm_gate_scores_fwd's shapes (the four rows of a workgroup share one batch element's gate vectors)."""
    return GATE_SCORES_FUSE and L % 4 == 0 and D % 4 == 0 and D <= 256 and H <= 8


def video_half(layer, x32, pos, u, dt, u_next=None, sc_in=None):
    """-> (m32, m, m + pos, sc_next): sc_next = the next layer's gate workspace with its scores when u_next was given (pass it to
    that layer as sc_in), else None."""
    pl = plan(layer, VH, dt, dt)
    if u_next is not None and not gate_scores_fusable(x32.shape[1], x32.shape[2], layer.nhead):
        u_next = None
    m32, m, mpos, sc = VideoHalfFn.apply(pl, x32, pos, u, u_next, sc_in, *pl.inputs(x32, u))
    return m32, m, mpos, (sc if u_next is not None else None)


# ----------------------------------------------------------------------------------------------------------------------
# query self-attention
# ----------------------------------------------------------------------------------------------------------------------
def _qpos_target(pl, qpos, tbl, dev, N, D):
    """where the gradient of the broadcast query_pos goes: straight into the query embedding's bucket view when qpos IS that
    parameter (fp32 query stream) and it has a sink, else a zeroed buffer returned to autograd."""
    if not qpos.requires_grad:
        tbl.set('DQPOS', None)
        return None, False
    s = ops._claim(qpos, True) if (qpos.is_leaf and qpos.dtype == torch.float32) else None
    if s is not None:
        tbl.set_t('DQPOS', s.view)
        return None, True
    g = torch.zeros((N, D), dtype=torch.float32, device=dev)
    tbl.set('DQPOS', g.data_ptr())
    return g, False


class QuerySelfFn(torch.autograd.Function):
    """LN4(o32 + SelfAttn(q = k = o + query_pos, v = o)) (+query_pos) — cross_modal_transformer.py:145-147."""

    @staticmethod
    def forward(ctx, pl, o32, o, opos, qpos, *params):
        ctx.set_materialize_grads(False)
        B, N, D = o.shape
        layer = pl.layer()
        H = layer.nhead
        qdt = pl.qdt
        e = _ESZ[qdt]
        dev = o.device
        R = B * N
        c = lambda t_: t_ if t_.is_contiguous() else t_.contiguous()
        o32_, o_, opos_, qpos_ = c(o32), c(o), c(opos), c(qpos)
        assert o32_.dtype == torch.float32 and o_.dtype == qdt and opos_.dtype == qdt and qpos_.dtype == qdt
        wsb = _attn_ws_bytes(B, H, N, N, D // H, qdt)
        key = (QS, B, N, D, H, qdt)
        lay = _layout(('f',) + key, lambda: [('QKV', R * 3 * D * e), ('OA', R * D * e), ('LSE', B * H * N * 4), ('S4', R * D * 4),
                                              ('MEAN4', R * 4), ('RSTD4', R * 4), ('Y32', R * D * 4), ('Y', R * D * e), ('YPOS', R * D * e)])
        slay = _layout(('s',) + key, lambda: [('ATTN_WS', max(wsb, 16))])
        arena = torch.empty((lay.total,), dtype=torch.uint8, device=dev)
        scr = _scratch(dev, slay.total)
        tbl = _Table(QS, pl.tpl)
        lay.fill(tbl, arena.data_ptr())
        slay.fill(tbl, scr.data_ptr())
        tbl.set('O32', o32_.data_ptr())
        tbl.set('O', o_.data_ptr())
        tbl.set('OPOS', opos_.data_ptr())
        tbl.set('QPOS', qpos_.data_ptr())
        dims = _dims(B, 0, N, D, H, 0, qdt, qdt, wsb)
        _lib.check(_lib.lib().svol_query_self_fwd(dims, tbl.arr, ops._stream()), 'svol_query_self_fwd')
        ctx.pl, ctx.tbl, ctx.dims, ctx.arena, ctx.shape = pl, tbl, dims, arena, (B, N, D, H)
        ctx.n_par = len(params)
        ctx.save_for_backward(o32_, o_, opos_, qpos)
        return (lay.view(arena, 'Y32', torch.float32, (B, N, D)), lay.view(arena, 'Y', qdt, (B, N, D)),
                lay.view(arena, 'YPOS', qdt, (B, N, D)))

    @staticmethod
    def backward(ctx, dy32, dy, dypos):
        pl, tbl, dims = ctx.pl, ctx.tbl, ctx.dims
        B, N, D, H = ctx.shape
        o32_, o_, opos_, qpos = ctx.saved_tensors
        n_par = ctx.n_par
        if dy32 is None and dy is None and dypos is None:
            pl.done(n_par > 0)
            return (None,) * (5 + n_par)
        qdt = pl.qdt
        e = _ESZ[qdt]
        dev = o_.device
        R = B * N
        cont = lambda t_: None if t_ is None else (t_ if t_.is_contiguous() else t_.contiguous())
        dy32, dy, dypos = cont(dy32), cont(dy), cont(dypos)
        lay = _layout(('b', QS, B, N, D, H, qdt), lambda: [('G', R * D * e), ('DOA', R * D * e), ('DQKV', R * 3 * D * e),
                                                            ('DELTA', 3 * B * H * N * 4), ('DO32', R * D * 4), ('DXQ', R * D * e),
                                                            ('DXQP', R * D * e)])
        tmp = torch.empty((lay.total,), dtype=torch.uint8, device=dev)
        lay.fill(tbl, tmp.data_ptr())
        tbl.set_t('DY32', dy32)
        tbl.set_t('DY', dy)
        tbl.set_t('DYPOS', dypos)
        slay = _layout(('s', QS, B, N, D, H, qdt), None)
        scr = _scratch(dev, slay.total)
        slay.fill(tbl, scr.data_ptr())
        needs = ctx.needs_input_grad
        gq, _ = _qpos_target(pl, qpos, tbl, dev, N, D) if needs[4] else (None, False)
        if not needs[4]:
            tbl.set('DQPOS', None)
        bufs = pl.grad_buffers(tbl, dev)
        L_ = _lib.lib()
        s = ops._stream()
        _lib.check(L_.svol_query_self_bwd(dims, tbl.arr, s), 'svol_query_self_bwd')
        keep = (ctx.arena, tmp, o_, opos_)
        if bufs is None:
            _issue_wgrad('svol_query_self_wgrad', dims, tbl, list(keep), dev)
        else:
            _lib.check(L_.svol_query_self_wgrad(dims, tbl.arr, s), 'svol_query_self_wgrad')
        pl.done(n_par > 0)
        if gq is not None and qdt != torch.float32:
            gq = ops.cast(gq, qdt)
        v = lambda n_, dt_: lay.view(tmp, n_, dt_, (B, N, D))
        return (None, v('DO32', torch.float32), v('DXQ', qdt), v('DXQP', qdt), gq) + tuple(pl.grads_out(bufs, needs[5:], n_par > 0))


def query_self(layer, out, qpos, dt, qdt):
    pl = plan(layer, QS, dt, qdt)
    o32, o, opos = out
    return QuerySelfFn.apply(pl, o32, o, opos, qpos, *pl.inputs(o32, o, opos))


# ----------------------------------------------------------------------------------------------------------------------
# query -> video cross-attention + MLP2
# ----------------------------------------------------------------------------------------------------------------------
def _qc_fwd_layout(B, N, L, D, H, F, e, qe):
    R, M = B * N, B * L
    return [('Q', R * D * qe), ('QC', R * D * e), ('KV', M * 2 * D * e), ('OA', R * D * e), ('OAQ', R * D * qe), ('LSE', B * H * N * 4),
            ('S5', R * D * 4), ('Y5_32', R * D * 4), ('Y5', R * D * qe), ('MEAN5', R * 4), ('RSTD5', R * 4), ('PRE', R * F * qe),
            ('HID', R * F * qe), ('S6', R * D * 4), ('MEAN6', R * 4), ('RSTD6', R * 4), ('Y32', R * D * 4), ('Y', R * D * qe),
            ('YPOS', R * D * qe)]


def _qc_bwd_layout(B, N, L, D, H, F, e, qe):
    R, M = B * N, B * L
    return [('DS32_6', R * D * 4), ('DS6', R * D * qe), ('DPRE', R * F * qe), ('DY5', R * D * qe), ('G5D', R * D * qe), ('DOAQ', R * D * qe),
            ('DOA', R * D * e), ('DQC', R * D * e), ('DQ', R * D * qe), ('DKV', M * 2 * D * e), ('DELTA', 3 * B * H * N * 4),
            ('DO32', R * D * 4), ('DXQP', R * D * qe), ('DMPOS', M * D * e), ('DMV', M * D * e)]


class QueryCrossFn(torch.autograd.Function):
    """LN6(. + MLP2(.)) o LN5(o32 + CrossAttn(q = o + query_pos, k = m + pos, v = m, key_padding_mask)) (+query_pos) —
    cross_modal_transformer.py:151-158."""

    @staticmethod
    def forward(ctx, pl, o32, o, opos, mv, mpos, kbias, qpos, *params):
        ctx.set_materialize_grads(False)
        B, N, D = o.shape
        L = mv.shape[1]
        layer = pl.layer()
        H, F = layer.nhead, layer.mlp2.fc1.weight.shape[0]
        dt, qdt = pl.dt, pl.qdt
        e, qe = _ESZ[dt], _ESZ[qdt]
        dev = o.device
        c = lambda t_: t_ if t_.is_contiguous() else t_.contiguous()
        o32_, o_, opos_, mv_, mpos_, kb_, qpos_ = c(o32), c(o), c(opos), c(mv), c(mpos), c(kbias), c(qpos)
        assert o32_.dtype == torch.float32 and o_.dtype == qdt and opos_.dtype == qdt and qpos_.dtype == qdt
        assert mv_.dtype == dt and mpos_.dtype == dt and kb_.dtype == torch.float32
        wsb = _attn_ws_bytes(B, H, N, L, D // H, dt)
        key = (QC, B, N, L, D, H, F, dt, qdt)
        lay = _layout(('f',) + key, lambda: _qc_fwd_layout(B, N, L, D, H, F, e, qe))
        slay = _layout(('s',) + key, lambda: [('ATTN_WS', max(wsb, 16))])
        arena = torch.empty((lay.total,), dtype=torch.uint8, device=dev)
        scr = _scratch(dev, slay.total)
        tbl = _Table(QC, pl.tpl)
        lay.fill(tbl, arena.data_ptr())
        slay.fill(tbl, scr.data_ptr())
        for n_, t_ in (('O32', o32_), ('O', o_), ('OPOS', opos_), ('MV', mv_), ('MPOS', mpos_), ('KBIAS', kb_), ('QPOS', qpos_)):
            tbl.set(n_, t_.data_ptr())
        dims = _dims(B, L, N, D, H, F, dt, qdt, wsb)
        _lib.check(_lib.lib().svol_query_cross_fwd(dims, tbl.arr, ops._stream()), 'svol_query_cross_fwd')
        ctx.pl, ctx.tbl, ctx.dims, ctx.arena, ctx.shape = pl, tbl, dims, arena, (B, N, L, D, H, F)
        ctx.n_par = len(params)
        ctx.save_for_backward(o32_, o_, opos_, mv_, mpos_, kb_, qpos)
        return (lay.view(arena, 'Y32', torch.float32, (B, N, D)), lay.view(arena, 'Y', qdt, (B, N, D)),
                lay.view(arena, 'YPOS', qdt, (B, N, D)))

    @staticmethod
    def backward(ctx, dy32, dy, dypos):
        pl, tbl, dims = ctx.pl, ctx.tbl, ctx.dims
        B, N, L, D, H, F = ctx.shape
        o32_, o_, opos_, mv_, mpos_, kb_, qpos = ctx.saved_tensors
        n_par = ctx.n_par
        if dy32 is None and dy is None and dypos is None:
            pl.done(n_par > 0)
            return (None,) * (8 + n_par)
        dt, qdt = pl.dt, pl.qdt
        e, qe = _ESZ[dt], _ESZ[qdt]
        dev = o_.device
        cont = lambda t_: None if t_ is None else (t_ if t_.is_contiguous() else t_.contiguous())
        dy32, dy, dypos = cont(dy32), cont(dy), cont(dypos)
        lay = _layout(('b', QC, B, N, L, D, H, F, dt, qdt), lambda: _qc_bwd_layout(B, N, L, D, H, F, e, qe))
        tmp = torch.empty((lay.total,), dtype=torch.uint8, device=dev)
        lay.fill(tbl, tmp.data_ptr())
        tbl.set_t('DY32', dy32)
        tbl.set_t('DY', dy)
        tbl.set_t('DYPOS', dypos)
        slay = _layout(('s', QC, B, N, L, D, H, F, dt, qdt), None)
        scr = _scratch(dev, slay.total)
        slay.fill(tbl, scr.data_ptr())
        needs = ctx.needs_input_grad
        gq = None
        if needs[7]:
            gq, _ = _qpos_target(pl, qpos, tbl, dev, N, D)
        else:
            tbl.set('DQPOS', None)
        bufs = pl.grad_buffers(tbl, dev)
        L_ = _lib.lib()
        s = ops._stream()
        _lib.check(L_.svol_query_cross_bwd(dims, tbl.arr, s), 'svol_query_cross_bwd')
        keep = (ctx.arena, tmp, opos_, mv_, mpos_)
        if bufs is None:
            _issue_wgrad('svol_query_cross_wgrad', dims, tbl, list(keep), dev)
        else:
            _lib.check(L_.svol_query_cross_wgrad(dims, tbl.arr, s), 'svol_query_cross_wgrad')
        pl.done(n_par > 0)
        if gq is not None and qdt != torch.float32:
            gq = ops.cast(gq, qdt)
        vq = lambda n_, dt_: lay.view(tmp, n_, dt_, (B, N, D))
        vm = lambda n_: lay.view(tmp, n_, dt, (B, L, D))
        return (None, vq('DO32', torch.float32), None, vq('DXQP', qdt), vm('DMV'), vm('DMPOS'), None, gq) + \
            tuple(pl.grads_out(bufs, needs[8:], n_par > 0))


def query_cross(layer, out, mv, mpos, kbias, qpos, dt, qdt):
    pl = plan(layer, QC, dt, qdt)
    o32, o, opos = out
    return QueryCrossFn.apply(pl, o32, o, opos, mv, mpos, kbias, qpos, *pl.inputs(o32, opos, mv, mpos))


def trace_dump() -> str:
    """In-step durations per call site of the block programs (``SVOL_BLOCK_TRACE=1``; include/svol_hip.h svol_block_trace_dump)."""
    buf = ctypes.create_string_buffer(1 << 20)
    _lib.check(_lib.lib().svol_block_trace_dump(buf, len(buf)), 'svol_block_trace_dump')
    return buf.value.decode()
