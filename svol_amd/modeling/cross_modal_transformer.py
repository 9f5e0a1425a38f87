"""Cross-modal transformer of the SVANet head on the MI355X kernels.

Mirrors the reference's module tree so that state-dict keys (and, for a given
seed, the initial weights) are identical —
lib/modeling/cross_modal_transformer.py:9-25,84-100,163-179,196-202 — but the
``nn.MultiheadAttention`` / ``nn.Linear`` / ``nn.LayerNorm`` children are used
purely as PARAMETER CONTAINERS: their ``forward`` is never called.  All
arithmetic goes through ``svol_amd.ops`` (hand-written HIP behind the C-ABI):

  per layer (reference :105-160)
    gate        : GateFn    (folded 1-query attention + x*(1+a) + LN1, one fused op)
    video SA    : AttnLNFn  (packed QKV GEMMs, flash attention, out_proj + residual, LN2)
    MLP1        : MLPLNFn   (fc1+GELU, fc2 + residual, LN3 (+pos))
    query SA    : AttnLNFn  (... LN4 (+query_pos))
    cross-attn  : AttnLNFn  (key_padding_mask as additive bias, LN5)
    MLP2        : MLPLNFn   (... LN6 (+query_pos))

The post-norm residual stream (every LN input and output) is carried in fp32 even in bf16 mode;
each block also emits the bf16 copies (x, x + pos) that feed the MFMA GEMMs.

The [B,L,L] attention-weight stacks the reference returns (never consumed by
SVANet.forward, svanet.py:91) are not materialised.
"""
from __future__ import annotations

import copy
import os

import torch
from torch import nn

from .. import blocks, ops

D_FF = 2048  # the reference hard-codes dim_feedforward=2048 here and ignores --dim_feedforward


class MLP(nn.Module):
    """fc1 / fc2 parameter container (reference MLP, :163-179); the arithmetic is fused with the
    residual add and the following post-norm in ``ops.mlp_ln``."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


OVERLAP_QUERY_STREAM = os.environ.get('SVOL_NO_STREAM_OVERLAP') is None
# bf16 mode: the N object queries (a few hundred rows: < 1 % of the step's FLOPs) run their projections, MLP2, query
# self-attention, residuals and norms in exact fp32 GEMMs; only the K / V projections of the L video tokens and the
# query -> video attention core stay on the bf16 MFMA path (ops.AttnLNFn "mixed").  The final logits / boxes are linear
# heads of the query stream, so its bf16 operand roundings were the larger half of the output error: measured on the
# reference goldens 1.0e-2 -> ~5e-3 max |logit error| (profiles/round2_bf16_output_error.md).  SVOL_QUERY_BF16=1 restores
# the all-bf16 query stream (A/B).
QUERY_FP32 = os.environ.get('SVOL_QUERY_BF16') is None
_SIDE = {}


def _side_stream(dev):
    s = _SIDE.get(dev)
    if s is None:
        s = _SIDE[dev] = torch.cuda.Stream(device=dev)
        # parameters of the query half get their gradients from side-stream nodes by design; a caller that still holds
        # last step's outputs keeps that step's AccumulateGrad nodes (and their streams) alive, which is harmless here
        # (BucketedGradAllReduce / the next forward join the streams) but makes torch warn on every backward
        try:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:  # pragma: no cover  (older torch)
            pass
    return s


def side_streams(dev):
    """side streams this module has used on `dev` (svol_amd.parallel joins them before reading gradient buckets)."""
    return [s for d, s in _SIDE.items() if d == dev]


class _GateVectorsFn(torch.autograd.Function):
    """u[b,h,:] = d_h^-1/2 * W_k,h^T (W_q,h s_b + b_q,h)  (B*d-sized algebra: svol_gate_vectors_fwd / _bwd, one launch forward and
    two backward).  The packed in_proj parameters get ONE gradient each, accumulated straight into the reducer's bucket when
    there is one."""

    @staticmethod
    def forward(ctx, skch, W_in, b_in, h):
        from .. import _lib
        B, d = skch.shape
        skch = skch.contiguous().float()
        q = torch.empty((B, d), dtype=torch.float32, device=skch.device)
        u = torch.empty((B, h, d), dtype=torch.float32, device=skch.device)
        Wd, bd = W_in.detach(), b_in.detach()
        rc = _lib.lib().svol_gate_vectors_fwd(ops._ptr(skch), ops._ptr(Wd), ops._ptr(bd), ops._ptr(q), ops._ptr(u), B, d, h, ops._stream())
        _lib.check(rc, 'svol_gate_vectors_fwd')
        ctx.save_for_backward(skch, W_in, q)
        ctx.h = h
        ctx.sinks = (ops._claim(W_in, ctx.needs_input_grad[1]), ops._claim(b_in, ctx.needs_input_grad[2]))
        return u

    @staticmethod
    def backward(ctx, du):
        from .. import _lib
        skch, W_in, q = ctx.saved_tensors
        h = ctx.h
        B, d = skch.shape
        sW, sb = ctx.sinks
        du = du.contiguous().float()
        dev = du.device
        dW = sW.view if sW is not None else torch.zeros((3 * d, d), dtype=torch.float32, device=dev)
        db = sb.view if sb is not None else torch.zeros((3 * d,), dtype=torch.float32, device=dev)
        dskch = torch.empty((B, d), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        ws = torch.empty((B, d), dtype=torch.float32, device=dev)
        rc = _lib.lib().svol_gate_vectors_bwd(ops._ptr(du), ops._ptr(skch), ops._ptr(W_in.detach()), ops._ptr(q), ops._ptr(ws),
                                              ops._ptr(dskch), ops._ptr(dW), ops._ptr(db), B, d, h, ops._stream())
        _lib.check(rc, 'svol_gate_vectors_bwd')
        return dskch, (None if sW is not None else dW), (None if sb is not None else db), None


def _ptr_array(tensors):
    """host array of device pointers (None -> NULL) for the *_multi entry points"""
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class _GateVectorsAllFn(torch.autograd.Function):
    """_GateVectorsFn for ALL layers at once (they read the same sketch token): svol_gate_vectors_fwd_multi / _bwd_multi, one launch
    forward and two backward instead of one and two PER LAYER.  The node is the oldest of the step, so the engine runs its backward
    last, when every layer's du exists.  Inputs: skch, h, then (in_proj_weight, in_proj_bias) per layer; outputs: one u per layer."""

    @staticmethod
    def forward(ctx, skch, h, *params):
        from .. import _lib
        B, d = skch.shape
        n = len(params) // 2
        skch = skch.contiguous().float()
        dev = skch.device
        Ws, bs = params[0::2], params[1::2]
        q = torch.empty((n, B, d), dtype=torch.float32, device=dev)
        us = [torch.empty((B, h, d), dtype=torch.float32, device=dev) for _ in range(n)]
        rc = _lib.lib().svol_gate_vectors_fwd_multi(ops._ptr(skch), _ptr_array([w.detach() for w in Ws]), _ptr_array([b.detach() for b in bs]),
                                                    _ptr_array(list(q.unbind(0))), _ptr_array(us), n, B, d, h, ops._stream())
        _lib.check(rc, 'svol_gate_vectors_fwd_multi')
        ctx.save_for_backward(skch, q, *Ws)
        ctx.h, ctx.n = h, n
        ctx.sinks = [(ops._claim(Ws[l], ctx.needs_input_grad[2 + 2 * l]), ops._claim(bs[l], ctx.needs_input_grad[3 + 2 * l])) for l in range(n)]
        return tuple(us)

    @staticmethod
    def backward(ctx, *dus):
        from .. import _lib
        skch, q = ctx.saved_tensors[:2]
        Ws = ctx.saved_tensors[2:]
        h, n = ctx.h, ctx.n
        B, d = skch.shape
        dev = skch.device
        dus = [torch.zeros((B, h, d), dtype=torch.float32, device=dev) if g is None else g.contiguous().float() for g in dus]
        dWs = [sW.view if sW is not None else torch.zeros((3 * d, d), dtype=torch.float32, device=dev) for sW, _ in ctx.sinks]
        dbs = [sb.view if sb is not None else torch.zeros((3 * d,), dtype=torch.float32, device=dev) for _, sb in ctx.sinks]
        ws = torch.empty((n, B, d), dtype=torch.float32, device=dev)
        dsk = torch.empty((n, B, d), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        rc = _lib.lib().svol_gate_vectors_bwd_multi(_ptr_array(dus), ops._ptr(skch), _ptr_array([w.detach() for w in Ws]),
                                                    _ptr_array(list(q.unbind(0))), _ptr_array(list(ws.unbind(0))),
                                                    None if dsk is None else _ptr_array(list(dsk.unbind(0))), _ptr_array(dWs),
                                                    _ptr_array(dbs), n, B, d, h, ops._stream())
        _lib.check(rc, 'svol_gate_vectors_bwd_multi')
        grads = [None if dsk is None else dsk.sum(0), None]
        for l, (sW, sb) in enumerate(ctx.sinks):
            grads += [None if sW is not None else dWs[l], None if sb is not None else dbs[l]]
        return tuple(grads)


GATE_VECTORS_MULTI_MAX = 8      # SVOL_GATE_VEC_MAX_LAYERS (include/svol_hip.h)


def all_gate_vectors(layers, skch):
    """[layer.gate_vectors(skch) for layer in layers] in one launch each way when the layers allow it"""
    import os
    if len(layers) > GATE_VECTORS_MULTI_MAX or len(layers) < 2 or os.environ.get('SVOL_GATE_VEC_PER_LAYER') is not None:
        return [layer.gate_vectors(skch) for layer in layers]
    params = []
    for layer in layers:
        a = layer.sketch_video_cross_attn
        params += [a.in_proj_weight, a.in_proj_bias]
    return list(_GateVectorsAllFn.apply(skch, layers[0].nhead, *params))


class CrossModalTransformerLayer(nn.Module):
    def __init__(self, d_model=512, nhead=8, dim_feedforward=D_FF):
        super().__init__()
        self.sketch_video_cross_attn = nn.MultiheadAttention(d_model, nhead)
        self.norm1 = nn.LayerNorm(d_model)
        self.content_self_attn = nn.MultiheadAttention(d_model, nhead)
        self.norm2 = nn.LayerNorm(d_model)
        self.mlp1 = MLP(d_model, dim_feedforward)
        self.norm3 = nn.LayerNorm(d_model)
        self.token_self_attn = nn.MultiheadAttention(d_model, nhead)
        self.norm4 = nn.LayerNorm(d_model)
        self.content_token_cross_attn = nn.MultiheadAttention(d_model, nhead)
        self.norm5 = nn.LayerNorm(d_model)
        self.mlp2 = MLP(d_model, dim_feedforward)
        self.norm6 = nn.LayerNorm(d_model)
        self.d_model, self.nhead = d_model, nhead

    def gate_vectors(self, skch):
        """u[b,h,:] = d_h^-1/2 * W_k,h^T (W_q,h skch_b + b_q,h): the 1-query attention's key
        projection folded into one d-vector per (batch, head).  [B,d] fp32 -> [B,H,d] fp32.
        (B*d-sized host-graph arithmetic; the L-sized work is in GateFn.)"""
        m = self.sketch_video_cross_attn
        return _GateVectorsFn.apply(skch, m.in_proj_weight, m.in_proj_bias, self.nhead)

    def forward(self, mem32, skch32, out, pos, qpos, kbias, u=None):
        """mem32: fp32 video stream [B,L,d]; out = (out32, out, out + query_pos) query stream triple; u: this layer's gate
        vectors when the caller computed them ahead of the layer loop."""
        m32, m, mpos = self.video_half(mem32, skch32, pos, u)
        return m32, self.query_half(out, m, mpos, qpos, kbias)

    def video_half(self, mem32, skch32, pos, u=None, u_next=None, sc_in=None):
        """the "encoder-like" half, :122-143 -> (fp32 stream, compute-dtype copy, copy + pos).
        With u_next (the NEXT layer's gate vectors; block programs only) -> (.., .., .., sc_next): this layer's last LayerNorm also
        writes the next layer's gate scores; hand sc_next to that layer's call as sc_in (None when the shape is not fusable)."""
        if blocks.ENABLED:   # one C call for the whole half (svol_video_half_fwd), same kernels in the same order
            uu = self.gate_vectors(skch32) if u is None else u
            m32, m, mpos, sc = blocks.video_half(self, mem32, pos, uu, pos.dtype, u_next, sc_in)
            return (m32, m, mpos) if u_next is None else (m32, m, mpos, sc)
        assert sc_in is None
        h = self.nhead
        n = lambda m: (m.weight, m.bias)
        mha = lambda m: (m.in_proj_weight, m.in_proj_bias, m.out_proj.weight, m.out_proj.bias)
        mlp = lambda m: (m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
        m32, m, mpos = ops.gate(mem32, pos, self.gate_vectors(skch32) if u is None else u, *n(self.norm1), h)
        m32, m = ops.self_attn_ln(m32, m, mpos, *mha(self.content_self_attn), *n(self.norm2), None, h)
        return ops.mlp_ln(m32, m, *mlp(self.mlp1), *n(self.norm3), pos)

    def query_half(self, out, m, mpos, qpos, kbias):
        """the "decoder-like" half, :145-158: object queries attend to themselves, then to the video tokens."""
        return self.query_cross(self.query_self(out, qpos, m.dtype), m, mpos, qpos, kbias)

    def query_self(self, out, qpos, dt=None):
        """:145-147: query self-attention + norm4 — needs only the previous layer's queries, not this layer's video tokens.
        dt: element type of the video tokens (the block plans are keyed by it)."""
        if blocks.ENABLED:
            return blocks.query_self(self, out, qpos, dt if dt is not None else qpos.dtype, qpos.dtype)
        n = lambda m_: (m_.weight, m_.bias)
        mha = lambda m_: (m_.in_proj_weight, m_.in_proj_bias, m_.out_proj.weight, m_.out_proj.bias)
        o32, o, opos = out
        return ops.self_attn_ln(o32, o, opos, *mha(self.token_self_attn), *n(self.norm4), qpos, self.nhead)

    def query_cross(self, out, m, mpos, qpos, kbias):
        """:149-158: query -> video cross-attention + norm5, MLP2 + norm6."""
        n = lambda m_: (m_.weight, m_.bias)
        mha = lambda m_: (m_.in_proj_weight, m_.in_proj_bias, m_.out_proj.weight, m_.out_proj.bias)
        mlp = lambda m_: (m_.fc1.weight, m_.fc1.bias, m_.fc2.weight, m_.fc2.bias)
        if blocks.ENABLED:
            return blocks.query_cross(self, out, m, mpos, kbias, qpos, m.dtype, qpos.dtype)
        o32, o, opos = out
        o32, o = ops.cross_attn_ln(o32, o, opos, mpos, m, *mha(self.content_token_cross_attn), *n(self.norm5), None, self.nhead,
                                   kbias)
        return ops.mlp_ln(o32, o, *mlp(self.mlp2), *n(self.norm6), qpos)


class CrossModalTransformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_layers=6, dim_feedforward=D_FF):
        super().__init__()
        layer = CrossModalTransformerLayer(d_model, nhead, dim_feedforward)
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        for p in self.parameters():  # reference _reset_parameters, :22-25
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead, self.num_layers = d_model, nhead, num_layers
        self._epoch_seen = -1

    def forward(self, src_vid32, src_skch32, kbias, vid_pos, query_embed, us=None):
        """src_vid32 [B,L,d] fp32 (projected video tokens = start of the fp32 residual stream),
        src_skch32 [B,d] fp32, kbias [B,L] fp32 additive key mask, vid_pos [B,L,d] compute dtype,
        query_embed [N,d] fp32 parameter; us: every layer's gate vectors when the caller computed them (all_gate_vectors; SVANet does,
        on the query stream beside the video projection).  Returns hs [num_layers,B,N,d] fp32."""
        # The compute-dtype weight copies are refreshed once per cache EPOCH; the heads that own this module (SVANet, the
        # svanet_variants) open one per forward.  Driven on its own (a training loop around the bare transformer), nobody does:
        # open it here, or an optimizer that rewrites parameters without bumping their version counters (torch's fused AdamW on ROCm)
        # would leave every copy stale
        if ops.weights.epoch == self._epoch_seen:
            ops.weights.new_epoch()
        self._epoch_seen = ops.weights.epoch
        B = src_vid32.shape[0]
        dt = torch.float32 if QUERY_FP32 else vid_pos.dtype   # element type of the QUERY stream's GEMM operands
        qpos = ops.cast_ag(query_embed, dt)
        N, d = qpos.shape
        def initial_queries():  # reference :56: tgt = zeros; (fp32 stream, compute copy, copy + query_pos)
            return (torch.zeros((B, N, d), dtype=torch.float32, device=vid_pos.device),
                    torch.zeros((B, N, d), dtype=dt, device=vid_pos.device),
                    qpos.unsqueeze(0).expand(B, -1, -1).contiguous())
        mem32 = src_vid32
        outputs = []
        # The gate vectors of every layer depend only on the sketch token and the layer's weights (B*d-sized algebra): all of them
        # are computed HERE, before the first big kernel is queued, instead of one tiny launch in front of each layer's gate where it
        # waits for a CU beside the other streams' kernels (15 us each).  Their autograd nodes get the lowest sequence numbers of the
        # transformer, so the engine runs their backward (two tiny launches per layer, 65 us each in the middle of the backward)
        # after everything else: 6 x 14 us at the end.  (svol_amd.parallel.arrival_order puts these parameters last accordingly.)
        if us is None:
            us = all_gate_vectors(list(self.layers), src_skch32)
        if not OVERLAP_QUERY_STREAM:
            out = initial_queries()
            for layer, u in zip(self.layers, us):
                mem32, out = layer(mem32, src_skch32, out, vid_pos, qpos, kbias, u)
                outputs.append(out[0])
            return torch.stack(outputs)
        # The query half of layer i (N = 100 object queries: skinny GEMMs, 64-512 workgroup attention launches, small
        # LayerNorms — none of them fills 256 CUs) only needs layer i's video tokens; the video half of layer i+1 does
        # not need the queries at all.  The query half therefore runs on a side stream under the next layer's video
        # kernels; autograd replays each node on its forward stream, so the backward overlaps the same way, and a
        # captured hipGraph gets the fork / join edges.
        main = torch.cuda.current_stream()
        side = _side_stream(main.device)
        side.wait_stream(main)  # qpos
        qpos.record_stream(side)
        kbias.record_stream(side)
        with torch.cuda.stream(side):
            # The query state is ALLOCATED on the side stream.  A tensor allocated on the main stream and dropped by the host
            # while a side-stream kernel still has it queued (the fp32 zeros are only a GEMM residual, nothing keeps them
            # alive) goes back to the main stream's pool and is handed to the next main-stream allocation: round 1 cloned
            # main-stream zeros here, which only moved the problem to the clone's SOURCE — dropped right after the clone was
            # queued, overwritten by the video half before a lagging side stream had read it (a 1-in-15 garbage forward in
            # the test suite, in streaks; never seen in the bench, where the side stream does not lag at that point)
            out = initial_queries()
        sc = None   # the gate scores the layer before computed for this one (blocks.video_half)
        for li, (layer, u) in enumerate(zip(self.layers, us)):
            # the query self-attention of layer i needs only layer i-1's queries: it goes out before this layer's video half is
            # issued and runs under it; only the cross-attention waits for the video tokens (the last layer's tail is shorter
            # by that block)
            with torch.cuda.stream(side):
                out_sa = layer.query_self(out, qpos, vid_pos.dtype)
            if blocks.ENABLED and blocks.GATE_SCORES_FUSE and li + 1 < len(us):
                m32, m, mpos, sc = layer.video_half(mem32, src_skch32, vid_pos, u, us[li + 1], sc)
            elif sc is not None:
                m32, m, mpos = layer.video_half(mem32, src_skch32, vid_pos, u, None, sc)
            else:
                m32, m, mpos = layer.video_half(mem32, src_skch32, vid_pos, u)
            side.wait_stream(main)
            m.record_stream(side)
            mpos.record_stream(side)
            with torch.cuda.stream(side):
                out = layer.query_cross(out_sa, m, mpos, qpos, kbias)
                outputs.append(out[0])
            mem32 = m32
        main.wait_stream(side)
        for t in outputs:
            t.record_stream(main)
        return torch.stack(outputs)


def build_cross_modal_transformer(args):
    return CrossModalTransformer(d_model=args.hidden_dim, nhead=args.nheads, num_layers=args.num_layers,
                                 dim_feedforward=D_FF)
