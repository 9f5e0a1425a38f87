"""Data-parallel gradient exchange for the SVOL training step: one process per GPU, bucketed
all-reduce over RCCL / xGMI (``torch.distributed`` backend "nccl" IS RCCL on ROCm), overlapped with
the backward pass.

The reference wraps the model in apex ``DistributedDataParallel(delay_allreduce=True)`` (train.py:124):
ONE flattened all-reduce of every gradient after backward has finished, no overlap.  Semantics kept:
the result is the MEAN over ranks of the per-rank gradients (each rank's loss is a mean over its LOCAL
batch / matched pairs, i.e. mean-of-means — SURVEY.md §8e), parameters that receive no gradient
(``sketch_video_cross_attn.out_proj.*``, ``class_head.*``) are skipped on every rank.

MI355X-first differences:
* gradients live in a few large flat fp32 buckets (``param.grad`` is a VIEW into its bucket), so a
  bucket is reduced in place with no gather/scatter copies;
* buckets are filled in the order backward PRODUCES gradients (``arrival_order``: box / class heads, then
  the transformer layers last to first, then the query embedding and the input projections, whose
  gradients only exist at the very end of backward) and each bucket's all-reduce is issued on a side
  stream the moment its last gradient has been accumulated, so the exchange of layer k overlaps the
  backward kernels of layers < k and only the last, small bucket is exposed;
* xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 74.4 MiB fp32 gradient of the benchmark
  head is cut into ~16 MiB buckets — large enough that each ring step is bandwidth- not latency-bound
  (per-link ring time ~ 2*(7/8)*16 MiB / 153 GB/s ~ 0.19 ms), small enough that the last bucket exposes
  little un-overlapped time.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

_LATE = ('query_embed', 'input_', 'backbone', 'pos_embed', 'position_embedding')   # first used in forward = last in backward
_GATE = ('sketch_video_cross_attn',)   # the gate-vector algebra of every layer runs ahead of the layer loop (cross_modal_transformer.py):
                                       # its backward is the last thing the transformer does, just before the input projections'
_EARLY = ('bbox_embed', 'class_embed')                                            # the heads: last in forward


def arrival_order(model: torch.nn.Module) -> List[torch.nn.Parameter]:
    """The trainable parameters of ``model`` in the order backward is expected to produce their gradients:
    reverse registration order (layers L-1 .. 0, inside a layer the last sub-block first), except that the heads
    (registered before the transformer in SVANet, svanet.py:42-44, but applied last) come first and what the forward
    touches first although it is registered last (input projections svanet.py:49-60, query embedding, backbones) comes
    last.  A wrong guess costs overlap, never correctness."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]

    def rank(name):
        if any(t in name for t in _EARLY):
            return 0
        if any(t in name for t in _LATE):
            return 3
        if any(t in name for t in _GATE):
            return 2
        return 1
    idx = sorted(range(len(named)), key=lambda i: (rank(named[i][0]), -i))
    return [named[i][1] for i in idx]


class BucketedGradAllReduce:
    """``params`` is taken in the order given — pass ``arrival_order(model)`` (a plain ``model.parameters()`` list is
    re-ordered by ``reversed`` as a fallback guess when ``ordered=False``)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20,
                 skip: Optional[Iterable[torch.nn.Parameter]] = None, process_group=None, ordered: bool = False,
                 tail_bytes: int = 2 << 20):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        skip_ids = {id(p) for p in (skip or [])}
        plist = [p for p in params if p.requires_grad and id(p) not in skip_ids]
        if not plist:
            raise ValueError('no parameters to reduce')
        self.device = plist[0].device
        self.on_gpu = self.device.type == 'cuda'
        order = plist if ordered else list(reversed(plist))
        self.buckets: List[dict] = []
        groups, cur, cur_bytes = [], [], 0
        for p in order:
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        # The LAST bucket's all-reduce cannot overlap with anything: its last gradient is the last thing backward produces.
        # Keep that exposed exchange small: the final `tail_bytes` of the arrival order (the input projections, the query
        # embedding, the end of layer 0) get a bucket of their own, the rest of the old last bucket reduces under layer 0's backward.
        if tail_bytes and groups:
            last, tail, tb = groups[-1], [], 0
            while len(last) > 1 and tb + last[-1].numel() * 4 <= tail_bytes:
                tb += last[-1].numel() * 4
                tail.insert(0, last.pop())
            if tail:
                groups.append(tail)
        for g in groups:
            self._make_bucket(g)
        self._handles = []
        self._hooks = []
        # SVOL_FORCE_ALLREDUCE=1: run the collective path at world size 1 too (a 1-rank RCCL communicator is a real communicator: the
        # all-reduce is enqueued on the communication stream behind the same producer-stream waits, finish() joins it) — how the
        # stream choreography is exercised against RCCL on a one-GPU box (tests/test_gpu_parallel.py)
        import os
        self.force = self.world == 1 and dist.is_initialized() and os.environ.get('SVOL_FORCE_ALLREDUCE') == '1'
        self.spans = []            # per launched bucket: (bucket index, start event, end event) on the communication stream (GPU only)
        self.exposed = None        # (event, event) around finish()'s wait for the communication stream: the exchange backward did not hide
        self.record_spans = self.force or os.environ.get('SVOL_DP_SPANS') == '1'
        # SVOL_ALLREDUCE_SPIN=ring|direct (with SVOL_FORCE_ALLREDUCE=1 only): a one-rank communicator's all-reduce is a copy — to
        # rehearse a collective that COSTS TIME on one GPU, a spin kernel of the exchange's modelled duration over xGMI at 8 ranks
        # follows every bucket's collective on the communication stream (7 links x ~153 GB/s per GPU,
        # point to point): ring 2 (7/8) bytes / 153 GB/s (per-link bound), direct reduce-scatter + all-gather 2 (bytes / 8) / 153 GB/s
        # + 20 us.  The spin holds one CU and moves no bytes: it tests the ORDER and EXPOSURE of the exchange (which buckets hide
        # behind backward, what finish() waits for), not the bandwidth RCCL's copies take from the weight-gradient stream.
        self.spin = os.environ.get('SVOL_ALLREDUCE_SPIN') if self.force else None
        if self.spin not in (None, 'ring', 'direct'):
            raise ValueError('SVOL_ALLREDUCE_SPIN must be ring or direct')
        self._spin_cycles_per_ms = None
        self.pending_scale = 1.0   # see finish(mean=False)
        self.on_bucket_reduced = None   # FlatAdamW(step_in_backward=True): callable(bucket index, stream handle), see _launch
        self._early_any = False
        self.fire_order: List[int] = []   # diagnostics: bucket index of every hook of the current step, in firing order
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        # the stream the caller computes on (the video half of the model, the criterion, the optimizer); re-read
        # at every zero_grad() on the caller's thread — hooks run on the autograd thread under the NODE's stream
        self.main_stream = torch.cuda.current_stream(self.device) if self.on_gpu else None
        from .ops import GradSink
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
                # the svol kernels accumulate weight / bias / LayerNorm gradients straight into the bucket view
                # (svol_amd.ops "gradient sinks"); the hook above still fires for them
                p._svol_sink = GradSink(p.grad, self, bi)

    def _make_bucket(self, params):
        # every tensor starts on a 16-byte boundary (the kernels take 16-byte vector accesses on gradients and — FlatAdamW —
        # on the parameters that share this layout); the few padding floats stay zero
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        for p, off in zip(params, offs):
            assert p.dtype == torch.float32, 'master parameters are fp32'
            p.grad = flat[off:off + p.numel()].view_as(p)  # autograd accumulates in place into the view
        self.buckets.append({'params': params, 'offsets': offs, 'flat': flat, 'pending': len(params), 'n': len(params)})

    def _make_hook(self, bi):
        def hook(_p):
            b = self.buckets[bi]
            self.fire_order.append(bi)
            b['pending'] -= 1
            if b['pending'] == 0:
                self._launch(b)
        return hook

    def params_done(self, bi, n=1):
        """`n` parameters of bucket `bi` have their complete gradient enqueued (or queued for the weight-gradient stream): what the
        per-parameter hooks do, for producers that write into the bucket without an autograd edge to the parameter (svol_amd.blocks)."""
        b = self.buckets[bi]
        self.fire_order.extend([bi] * n)
        b['pending'] -= n
        if b['pending'] < 0:
            # a second backward() without zero_grad() in between (gradient accumulation): the bucket was already handed to the
            # exchange once — reducing it again would double-count
            raise RuntimeError('BucketedGradAllReduce: more gradients arrived for a bucket than it has parameters — call zero_grad() '
                               'before every backward() (gradient accumulation over several backward passes is not supported by the '
                               'block programs\' gradient sinks)')
        if b['pending'] == 0:
            self._launch(b)

    def _producer_streams(self):
        """every stream a gradient of a bucket can have been written on: the caller's compute stream (video half,
        heads, input projections), the model's side stream (query half: gradient-sink kernels and AccumulateGrad nodes
        created under ``torch.cuda.stream(side)``), and the stream of the node whose hook is running."""
        from . import ops
        from .modeling import cross_modal_transformer as cmt
        ss = ([self.main_stream, torch.cuda.current_stream(self.device)] + list(cmt.side_streams(self.device)) +
              list(ops.wgrad_streams(self.device)))   # weight-gradient GEMMs run on a stream of their own (ops.gemm_tn_sink)
        out = []
        for s in ss:
            if s is not None and all(s != t for t in out):
                out.append(s)
        return out

    def _launch(self, b):
        collective = not (self.world == 1 and not self.force)
        early = self.on_bucket_reduced is not None and self.on_gpu
        if not collective:
            if early:
                self._early_step_local(b)
            return
        if self.on_gpu:
            # The hook runs on the autograd thread under the stream guard of the LAST-arriving parameter's node.  That
            # can be the side stream (query half), which runs far ahead of the main stream: waiting only on it would
            # start the all-reduce before the main stream has written the video-half gradients of the same bucket.
            # By the time the last hook of a bucket fires every producing kernel of the bucket has been ENQUEUED on
            # its stream (autograd issues nodes in order on the host), so waiting on all producer streams here is
            # sufficient.
            from . import ops
            ops.flush_wgrad()   # weight-gradient GEMMs still queued on the host (ops.gemm_tn_sink) may belong to this bucket
            for s in self._producer_streams():
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                if self.record_spans:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                h = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self.spin is not None:
                    h.wait()
                    torch.cuda._sleep(int(self.modelled_exchange_ms(b['flat'].numel() * 4) * self._spin_rate()))
                if self.record_spans:
                    h.wait()       # (stream-side wait: orders the communication stream behind the collective, does not block the host)
                    e1.record(self.comm_stream)
                    self.spans.append((next(i for i, x in enumerate(self.buckets) if x is b), e0, e1))
                if early:          # the bucket's optimizer update right behind its exchange, on the communication stream
                    h.wait()
                    self._early_any = True
                    self.on_bucket_reduced(next(i for i, x in enumerate(self.buckets) if x is b), self.comm_stream.cuda_stream)
        else:
            h = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append((h, b))

    def _early_step_local(self, b):
        """world size 1, FlatAdamW(step_in_backward=True): the bucket's update as soon as its gradients are final, on the communication
        stream behind every producer stream's work so far.  Weight-gradient launches still QUEUED on the host (ops: deferred to
        the next large attention backward) belong in front of it: the update joins that queue behind them, so the deferral keeps its
        schedule and the update lands beside the attention backward — an issue-bound launch that leaves the memory side idle."""
        from . import ops
        bi = next(i for i, x in enumerate(self.buckets) if x is b)

        def issue():
            for s in self._producer_streams():
                self.comm_stream.wait_stream(s)
            self._early_any = True
            self.on_bucket_reduced(bi, self.comm_stream.cuda_stream)

        if ops._WGRAD_PENDING:
            ops._WGRAD_PENDING.append(issue)
            ops.arm_wgrad_flush()
        else:
            issue()

    def modelled_exchange_ms(self, nbytes: int, ranks: int = 8, link_gbs: float = 153.0) -> float:
        """duration of one bucket's all-reduce over xGMI in the SVOL_ALLREDUCE_SPIN model (see __init__)."""
        if self.spin == 'direct':
            return 2.0 * (nbytes / ranks) / (link_gbs * 1e9) * 1e3 + 0.02
        return 2.0 * (ranks - 1) / ranks * nbytes / (link_gbs * 1e9) * 1e3

    def _spin_rate(self) -> float:
        """torch.cuda._sleep cycles per millisecond on this device (calibrated once, synchronising; first use only)."""
        if self._spin_cycles_per_ms is None:
            s_ = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(s_):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(1_000_000)
                e0.record(s_)
                torch.cuda._sleep(20_000_000)
                e1.record(s_)
            e1.synchronize()
            self._spin_cycles_per_ms = 20_000_000 / max(e0.elapsed_time(e1), 1e-3)
        return self._spin_cycles_per_ms

    def zero_grad(self):
        """Zero the flat buckets (replaces optimizer.zero_grad(); keeps the grad views alive)."""
        if self.on_gpu:
            self.main_stream = torch.cuda.current_stream(self.device)
            from . import ops
            ops.drop_pending_wgrad()   # weight-gradient launches a failed backward left queued must not land in the zeroed buckets
        self.fire_order = []
        self.spans = []
        self._early_any = False
        for b in self.buckets:
            if b.pop('clean', False):
                pass   # FlatAdamW(zero_grads=True) zeroed the range behind its read of it (svol_adamw_flat_zero)
            else:
                b['flat'].zero_()
            b['pending'] = b['n']
            for p in b['params']:
                if p.grad is None or p.grad.data_ptr() < b['flat'].data_ptr() or \
                        p.grad.data_ptr() >= b['flat'].data_ptr() + b['flat'].numel() * 4:
                    raise RuntimeError('a parameter gradient was re-bound (zero_grad(set_to_none=True)?); use '
                                       'BucketedGradAllReduce.zero_grad() instead of optimizer.zero_grad()')

    def finish(self, mean: bool = True):
        """Wait for every in-flight bucket and turn sums into means.  Call after backward() — MANDATORY before anything reads
        ``param.grad`` (a gradient-norm log, clipping, a torch optimizer), also at world size 1: the block programs write the gradient
        buckets from the query-half and weight-gradient streams without autograd edges to the parameters, so ``backward()``
        returning does not order those streams against the caller's; this call does.
        mean=False: leave the SUMS in the buckets and remember the factor (``pending_scale`` = 1 / world) for ``FlatAdamW.step()``,
        which applies it inside its one pass over the gradients (no multiply launch per bucket)."""
        if self.on_gpu:
            # kernels that accumulate straight into the buckets (gradient sinks) may have run on the model's side
            # stream (query-stream / video-stream overlap): join it before anyone reads the buckets
            from . import ops
            from .modeling import cross_modal_transformer as cmt
            ops.flush_wgrad()
            for s in list(cmt.side_streams(self.device)) + list(ops.wgrad_streams(self.device)):
                torch.cuda.current_stream().wait_stream(s)
        collective = self.world > 1 or self.force
        for b in self.buckets:
            if b['pending'] != 0 and collective:
                # a parameter of this bucket got no gradient this step (should not happen for a fixed
                # architecture); reduce what we have so that ranks stay in lock-step
                self._launch(b)
        for h, b in self._handles:
            h.wait()
        if self.on_gpu and self._early_any and not collective:
            torch.cuda.current_stream().wait_stream(self.comm_stream)   # bucket updates issued during backward (FlatAdamW(step_in_backward=True))
        if self.on_gpu and collective:
            cur = torch.cuda.current_stream()
            if self.record_spans:   # what the caller's stream waits here is the exchange that backward did not cover
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(cur)
            cur.wait_stream(self.comm_stream)
            if self.record_spans:
                eb.record(cur)
                self.exposed = (ea, eb)
        if self.world > 1:
            inv = 1.0 / self.world
            if mean:
                for b in self.buckets:
                    b['flat'].mul_(inv)
            else:
                self.pending_scale = inv
        self._handles.clear()

    def allreduce_report(self):
        """after a synchronised step with span recording on (SVOL_FORCE_ALLREDUCE / SVOL_DP_SPANS): per launched bucket (index, MiB, ms
        from the first recorded start, duration ms) on the communication stream, and the ms the caller's stream waited in finish()."""
        if not self.spans:
            return {'buckets': [], 'exposed_ms': 0.0}
        t0 = self.spans[0][1]
        out = [{'bucket': bi, 'mib': round(self.buckets[bi]['flat'].numel() * 4 / 2 ** 20, 2), 'start_ms': round(t0.elapsed_time(e0), 3),
                'ms': round(e0.elapsed_time(e1), 3)} for bi, e0, e1 in self.spans]
        exposed = self.exposed[0].elapsed_time(self.exposed[1]) if self.exposed else 0.0
        rep = {'buckets': out, 'exposed_ms': round(exposed, 4)}
        if self.spin is not None:
            rep['spin_model'] = self.spin
            rep['modelled_ms'] = [round(self.modelled_exchange_ms(self.buckets[bi]['flat'].numel() * 4), 3) for bi, _e0, _e1 in self.spans]
        return rep

    def bucket_fire_spans(self):
        """diagnostics after a backward: per bucket (first, last) position of its hooks in the step's firing order —
        with a good ``arrival_order`` the spans do not interleave and bucket k completes before bucket k+1 starts."""
        spans = {}
        for pos, bi in enumerate(self.fire_order):
            lo, _ = spans.get(bi, (pos, pos))
            spans[bi] = (lo, pos)
        return [spans.get(bi) for bi in range(len(self.buckets))]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        for b in self.buckets:
            for p in b['params']:
                if hasattr(p, '_svol_sink'):
                    del p._svol_sink


class DynamicLossScaler:
    """apex-amp style dynamic loss scaling for fp16 operands (the reference's fp16 mode: configs.py:52-61, ``amp.scale_loss`` at
    train.py:231-232) kept entirely on the device: ``scale(loss)`` multiplies by the scale tensor, ``FlatAdamW.step()`` checks the
    flat gradient buckets for inf / NaN, SKIPS the whole update when one is found (parameters, moments and the bias-correction step
    count keep their values), and halves / grows the scale — no host synchronisation, no ``.item()``.  ``state_dict()`` speaks the
    ``amp`` entry of the reference's checkpoints (``{'loss_scaler0': {'loss_scale', 'unskipped'}}``)."""

    def __init__(self, device, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, min_scale=1.0,
                 max_scale=2.0 ** 24):
        self.state = torch.tensor([float(init_scale), 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)

    def scale(self, loss):
        return loss * self.state[0]

    def state_dict(self):   # (a host read: checkpoint time only)
        st = self.state.tolist()
        return {'loss_scaler0': {'loss_scale': st[0], 'unskipped': int(st[2])}}

    def load_state_dict(self, sd):
        e = sd.get('loss_scaler0', sd)
        self.state[0] = float(e['loss_scale'])
        self.state[2] = float(e.get('unskipped', 0))


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW (decoupled weight decay, amsgrad off; the reference's optimizer, train.py:98-99) over the reducer's
    flat buckets: the parameters of a bucket are re-homed into ONE flat fp32 buffer (``param.data`` becomes a view, like
    ``param.grad`` already is), the moments are flat too, and a step is one streaming kernel per bucket (``svol_adamw_flat``:
    28 bytes per parameter) instead of torch's multi-tensor launches over ~150 separate tensors (0.44 ms -> 0.15 ms at the
    benchmark size).  Parameters the reducer skips (they never receive a gradient) are not touched, as in torch.

    It IS a ``torch.optim.Optimizer``: one ``param_groups`` entry whose ``lr`` / ``betas`` / ``eps`` / ``weight_decay`` the step
    reads (so ``StepLR`` / ``MultiStepLR`` of the reference loop, train.py:127-130, wrap it), and ``state_dict()`` /
    ``load_state_dict()`` speak torch AdamW's schema — ``{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups':
    [...]}`` with ``i`` indexing ``params`` — so a checkpoint written by the reference (train.py:267-284) resumes here and vice
    versa.  ``params``: the optimizer's parameter list in the REFERENCE's order (train.py:72: every trainable parameter in
    ``named_parameters()`` order, including the ones that never get a gradient and therefore have no state entry); defaults
    to the reducer's own parameters in bucket order."""

    def __init__(self, reducer: 'BucketedGradAllReduce', lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 params: Optional[Iterable[torch.nn.Parameter]] = None, zero_grads: bool = False, step_in_backward: bool = False):
        """zero_grads: the update kernel zeroes each gradient bucket behind its read (``svol_adamw_flat_zero``) and the reducer's next
        ``zero_grad()`` skips its fill — the reference loop's ``optimizer.zero_grad()`` (train.py:222) folded into ``step()``; for a
        loop that reads no ``param.grad`` between ``step()`` and ``zero_grad()``.
        step_in_backward: a bucket's update is issued as soon as its gradients are final (behind its all-reduce at world size > 1),
        on the reducer's communication stream, instead of all of them in ``step()`` behind the whole backward; ``step()`` then only
        updates what is left (the last bucket) and closes the step.  Same arithmetic on the same values: the reference has no
        gradient clipping or any other cross-bucket dependency between backward and the update (train.py:229-234).  NOT for a loop
        that may skip ``step()`` after ``backward()``; ignored while a loss scaler is attached (its overflow check spans all buckets)."""
        owned = [p for b in reducer.buckets for p in b['params']]
        plist = list(params) if params is not None else owned
        self._params_given = params is not None
        known = {id(p) for p in plist}
        missing = [p for p in owned if id(p) not in known]
        if missing:
            raise ValueError(f'{len(missing)} parameter(s) of the gradient buckets are not in `params`')
        defaults = dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                        weight_decay=float(weight_decay), amsgrad=False, maximize=False, foreach=None, capturable=False,
                        differentiable=False, fused=None)
        super().__init__(plist, defaults)
        self.reducer = reducer
        self.t = 0
        # the loss was multiplied by this before backward (fp16 operands: gradients of ~1e-6 underflow otherwise); divided back out
        # inside the update kernel
        self.loss_scale = 1.0
        self._scaler: Optional[DynamicLossScaler] = None   # set for fp16 training: overflow check + skip + dynamic scale (see step())
        self.flat = []      # per bucket: flat parameters / first / second moments
        self._slot = {}     # id(param) -> (bucket index, offset, numel)
        for bi, b in enumerate(reducer.buckets):
            flat_g = b['flat']
            flat_p = torch.zeros_like(flat_g)
            for p, off in zip(b['params'], b['offsets']):
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + n].view_as(p)  # same layout as the gradient views of the bucket
                self._slot[id(p)] = (bi, off, n)
            self.flat.append({'p': flat_p, 'm': torch.zeros_like(flat_p), 'v': torch.zeros_like(flat_p)})
        self.zero_grads = bool(zero_grads)
        self.step_in_backward = bool(step_in_backward)
        self._stepped = [False] * len(self.flat)
        if self.step_in_backward:
            reducer.on_bucket_reduced = self._step_bucket_early

    # With a scaler attached the bias-correction step count is the DEVICE count of updates really taken (scaler.state[3]: an
    # overflow-skipped step does not advance it, as apex + torch AdamW do not advance 'step').  It is what checkpoints carry
    # (a resume used to restart it at 1 under warm moments), so attaching a scaler or loading a state dict keeps the
    # two in step: whichever happens second pushes the loaded count to the device.
    @property
    def scaler(self) -> Optional['DynamicLossScaler']:
        return self._scaler

    @scaler.setter
    def scaler(self, sc: Optional['DynamicLossScaler']):
        # seed the new scaler's device count from the updates really TAKEN so far, read before the swap: with a scaler attached
        # self.t counts step() CALLS (overflow-skipped ones included), and replacing the scaler mid-run from it would inflate the
        # bias corrections
        taken = self.steps_taken()
        self._scaler = sc
        if sc is not None:
            sc.state[3] = float(taken)
        else:
            self.t = taken

    def steps_taken(self) -> int:
        """Updates really applied (what torch calls 'step'); a host read of the device count when a scaler is attached."""
        return int(self._scaler.state[3].item()) if self._scaler is not None else self.t

    # the hyper-parameters live in param_groups[0] (what torch's schedulers edit); attribute access kept for callers
    lr = property(lambda self: self.param_groups[0]['lr'], lambda self, v: self.param_groups[0].__setitem__('lr', float(v)))
    betas = property(lambda self: self.param_groups[0]['betas'])
    eps = property(lambda self: self.param_groups[0]['eps'])
    weight_decay = property(lambda self: self.param_groups[0]['weight_decay'])

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib
        from .ops import _ptr, _stream
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        if self.scaler is not None:
            # dynamic loss scaling (a static scale with no overflow check lets one inf gradient poison p, m and v for good):
            # the buckets are checked first, every update kernel is a no-op on an overflowed step, then the scale moves
            sc = self.scaler
            gmul = float(self.reducer.pending_scale)
            self.reducer.pending_scale = 1.0
            self.t += 1    # (host-side count of step() calls; the bias corrections use the DEVICE count of steps really taken)
            for b in self.reducer.buckets:
                _lib.check(_lib.lib().svol_grad_finite(_ptr(b['flat']), b['flat'].numel(), _ptr(sc.state), _stream()), 'svol_grad_finite')
            for b, st in zip(self.reducer.buckets, self.flat):
                rc = _lib.lib().svol_adamw_flat_scaled(_ptr(st['p']), _ptr(b['flat']), _ptr(st['m']), _ptr(st['v']), st['p'].numel(),
                                                       float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                                       float(g['weight_decay']), gmul, _ptr(sc.state), _stream())
                _lib.check(rc, 'svol_adamw_flat_scaled')
            _lib.check(_lib.lib().svol_loss_scaler_update(_ptr(sc.state), sc.growth_factor, sc.backoff_factor, sc.growth_interval,
                                                          sc.min_scale, sc.max_scale, _stream()), 'svol_loss_scaler_update')
            return loss
        self.t += 1
        gscale = float(self.reducer.pending_scale) / float(self.loss_scale)
        self.reducer.pending_scale = 1.0
        for bi, (b, st) in enumerate(zip(self.reducer.buckets, self.flat)):
            if self._stepped[bi]:      # updated during backward (_step_bucket_early), with this step's count
                self._stepped[bi] = False
                continue
            self._launch_bucket(b, st, g, self.t, gscale, _stream())
        return loss

    def _launch_bucket(self, b, st, g, t, gscale, stream):
        from . import _lib
        from .ops import _ptr
        fn = _lib.lib().svol_adamw_flat_zero if self.zero_grads else _lib.lib().svol_adamw_flat
        rc = fn(_ptr(st['p']), _ptr(b['flat']), _ptr(st['m']), _ptr(st['v']), st['p'].numel(), float(g['lr']), float(g['betas'][0]),
                float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), t, gscale, stream)
        _lib.check(rc, 'svol_adamw_flat_zero' if self.zero_grads else 'svol_adamw_flat')
        if self.zero_grads:
            b['clean'] = True

    @torch.no_grad()
    def _step_bucket_early(self, bi, stream):
        """reducer callback (step_in_backward): bucket ``bi``'s gradients are final and ``stream`` is ordered behind every kernel
        that wrote them or read its parameters in this step."""
        if self._scaler is not None or self._stepped[bi]:
            return
        g = self.param_groups[0]
        world = self.reducer.world
        gscale = (1.0 / world if world > 1 else 1.0) / float(self.loss_scale)
        self._launch_bucket(self.reducer.buckets[bi], self.flat[bi], g, self.t + 1, gscale, stream)
        self._stepped[bi] = True

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    def _views(self, p):
        bi, off, n = self._slot[id(p)]
        st = self.flat[bi]
        return st['m'][off:off + n].view_as(p), st['v'][off:off + n].view_as(p)

    def state_dict(self):
        """torch.optim.AdamW's schema.  A parameter has a state entry iff it has a slot in a gradient bucket and at least
        one step was taken (torch creates the entry at the first step of a parameter that has a gradient)."""
        plist = self.param_groups[0]['params']
        state = {}
        taken = self.steps_taken()   # (not the count of step() calls: overflow-skipped steps are not steps)
        if taken > 0:
            for i, p in enumerate(plist):
                if id(p) in self._slot:
                    m, v = self._views(p)
                    state[i] = {'step': torch.tensor(float(taken)), 'exp_avg': m.clone(), 'exp_avg_sq': v.clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(plist)))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        if 't' in sd and 'm' in sd:   # round-1 files of this build: flat moments per bucket, layout-dependent
            if len(sd['m']) != len(self.flat) or any(a.numel() != st['m'].numel() for a, st in zip(sd['m'], self.flat)):
                raise ValueError('flat optimizer state does not match this bucket layout')
            self.t, self.lr = int(sd['t']), float(sd['lr'])
            for st, m, v in zip(self.flat, sd['m'], sd['v']):
                st['m'].copy_(m)
                st['v'].copy_(v)
            if self._scaler is not None:
                self._scaler.state[3] = float(self.t)
            return
        if not self._params_given:
            # torch's schema maps state to parameters BY POSITION in the optimizer's list (train.py:72: named_parameters() order);
            # this optimizer was built in bucket (arrival) order, so the moments of the many equal-shaped d x d weights would be
            # silently permuted
            raise ValueError('FlatAdamW.load_state_dict: build the optimizer with params=<the parameter list in the order the checkpoint '
                             'was written with (the reference: every trainable parameter in named_parameters() order)>')
        groups = sd['param_groups']
        plist = self.param_groups[0]['params']
        ids = [i for g in groups for i in g['params']]
        if len(ids) != len(plist):
            raise ValueError(f'loaded state dict holds {len(ids)} parameters, the optimizer {len(plist)}')
        g0 = groups[0]
        for g in groups[1:]:
            if any(g.get(k) != g0.get(k) for k in ('lr', 'betas', 'eps', 'weight_decay')):
                raise ValueError('FlatAdamW keeps one parameter group: the loaded groups differ in their hyper-parameters')
        if g0.get('amsgrad') or g0.get('maximize'):
            raise ValueError('amsgrad / maximize are not supported')
        steps = set()
        todo = []
        for pos, i in enumerate(ids):
            ent = sd['state'].get(i)
            p = plist[pos]
            if ent is None:
                continue
            if id(p) not in self._slot:
                raise ValueError(f'parameter {pos} has optimizer state but no gradient bucket here')
            if ent['exp_avg'].numel() != p.numel() or ent['exp_avg_sq'].numel() != p.numel():
                raise ValueError(f'parameter {pos}: state of {ent["exp_avg"].numel()} elements for {p.numel()} weights')
            steps.add(int(float(ent['step'])))
            todo.append((p, ent))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}); FlatAdamW keeps one')
        for st in self.flat:
            st['m'].zero_()
            st['v'].zero_()
        for p, ent in todo:
            m, v = self._views(p)
            m.copy_(ent['exp_avg'].reshape(p.shape))
            v.copy_(ent['exp_avg_sq'].reshape(p.shape))
        self.t = steps.pop() if steps else 0
        if self._scaler is not None:
            self._scaler.state[3] = float(self.t)
        grp = self.param_groups[0]
        for k in ('lr', 'eps', 'weight_decay', 'initial_lr'):
            if k in g0:
                grp[k] = float(g0[k])
        if 'betas' in g0:
            grp['betas'] = (float(g0['betas'][0]), float(g0['betas'][1]))


class StepFence:
    """Bounds how far the host may run ahead of the GPU: ``tick()`` at the end of a training step records an event and waits (spinning,
    no interrupt) for the step ``max_inflight`` steps back.  With the block programs a step takes ~6 ms of host time against ~20 ms
    on the GPU, so an unfenced loop queues a dozen steps (>5000 packets over three streams) within a few iterations; at that depth
    the HIP runtime's own back-pressure (kernel-argument pool / signal recycling) takes over, and that wait was measured to stall
    host AND GPU for 0.1 - 1.7 s at a time (profiles/round3_summary.md, "host run-ahead").  Two steps in flight keep the queue full
    without ever reaching it.  The reference's loop is fenced the same way by its per-step ``loss.item()`` logging
    (lib/engine/train.py:233-240)."""

    def __init__(self, max_inflight: int = 2):
        from collections import deque
        self.max_inflight = max(1, int(max_inflight))
        self._q = deque()

    def tick(self) -> None:
        ev = torch.cuda.Event()
        ev.record()
        self._q.append(ev)
        while len(self._q) > self.max_inflight:
            self._q.popleft().synchronize()

    def drain(self) -> None:
        while self._q:
            self._q.popleft().synchronize()


def unused_parameters(model: torch.nn.Module):
    """Parameters that never receive a gradient (reference: grad None): the SVANet gate MHA's out_proj
    (only its attention WEIGHTS are used), the unused ``class_head``, and — svanet_variants — the input projections of
    the fusion modes the model was not built for."""
    mode = getattr(model, 'mode', None) or getattr(getattr(model, 'head', None), 'mode', None)
    dead = {'concat_to_seq': ('input_sketch_proj', 'input_video_proj', 'input_query_proj'),
            'append_to_seq': ('input_proj.', 'input_query_proj'),
            'concat_to_qry': ('input_proj.', 'input_sketch_proj')}.get(mode, ())
    out = []
    for name, p in model.named_parameters():
        if 'sketch_video_cross_attn.out_proj' in name or 'class_head' in name or any(d in name for d in dead):
            out.append(p)
    return out


def reduce_scalar_mean(t: torch.Tensor) -> torch.Tensor:
    """reduce_tensor of the reference (lib/utils/comm.py:21-25): mean of a logged scalar over ranks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t.clone()
    rt = t.detach().clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= dist.get_world_size()
    return rt
