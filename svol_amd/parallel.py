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
* buckets are filled in reverse parameter order (= backward order: heads and the last layer first) and
  each bucket's all-reduce is issued on a side stream the moment its last gradient has been accumulated,
  so the exchange of layer k overlaps the backward kernels of layers < k;
* xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 74.4 MiB fp32 gradient of the benchmark
  head is cut into ~16 MiB buckets — large enough that each ring step is bandwidth- not latency-bound
  (per-link ring time ~ 2*(7/8)*16 MiB / 153 GB/s ~ 0.19 ms), small enough that the last bucket (the
  input projections, ready only at the very end of backward) exposes little un-overlapped time.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class BucketedGradAllReduce:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 16 << 20,
                 skip: Optional[Iterable[torch.nn.Parameter]] = None, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        skip_ids = {id(p) for p in (skip or [])}
        plist = [p for p in params if p.requires_grad and id(p) not in skip_ids]
        if not plist:
            raise ValueError('no parameters to reduce')
        self.device = plist[0].device
        self.on_gpu = self.device.type == 'cuda'
        # reverse order ~ the order in which backward produces gradients
        order = list(reversed(plist))
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in order:
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._make_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._make_bucket(cur)
        self._handles = []
        self._hooks = []
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        from .ops import GradSink
        for bi, b in enumerate(self.buckets):
            for p in b['params']:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
                # the svol kernels accumulate weight / bias / LayerNorm gradients straight into the bucket view
                # (svol_amd.ops "gradient sinks"); the hook above still fires for them
                p._svol_sink = GradSink(p.grad)

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
            b['pending'] -= 1
            if b['pending'] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if self.world == 1:
            return
        if self.on_gpu:
            # the bucket's gradients were produced on the current (compute) stream — or, for the parameters of the
            # query half, by gradient-sink kernels on the model's side stream (cross_modal_transformer.py)
            from .modeling import cross_modal_transformer as cmt
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for s in cmt.side_streams(self.device):
                self.comm_stream.wait_stream(s)
            with torch.cuda.stream(self.comm_stream):
                h = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            h = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append((h, b))

    def zero_grad(self):
        """Zero the flat buckets (replaces optimizer.zero_grad(); keeps the grad views alive)."""
        for b in self.buckets:
            b['flat'].zero_()
            b['pending'] = b['n']
            for p in b['params']:
                if p.grad is None or p.grad.data_ptr() < b['flat'].data_ptr() or \
                        p.grad.data_ptr() >= b['flat'].data_ptr() + b['flat'].numel() * 4:
                    raise RuntimeError('a parameter gradient was re-bound (zero_grad(set_to_none=True)?); use '
                                       'BucketedGradAllReduce.zero_grad() instead of optimizer.zero_grad()')

    def finish(self):
        """Wait for every in-flight bucket and turn sums into means.  Call after backward()."""
        if self.on_gpu:
            # kernels that accumulate straight into the buckets (gradient sinks) may have run on the model's side
            # stream (query-stream / video-stream overlap): join it before anyone reads the buckets
            from .modeling import cross_modal_transformer as cmt
            for s in cmt.side_streams(self.device):
                torch.cuda.current_stream().wait_stream(s)
        for b in self.buckets:
            if b['pending'] != 0 and self.world > 1:
                # a parameter of this bucket got no gradient this step (should not happen for a fixed
                # architecture); reduce what we have so that ranks stay in lock-step
                self._launch(b)
        for h, b in self._handles:
            h.wait()
        if self.on_gpu and self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self.world > 1:
            inv = 1.0 / self.world
            for b in self.buckets:
                b['flat'].mul_(inv)
        self._handles.clear()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        for b in self.buckets:
            for p in b['params']:
                if hasattr(p, '_svol_sink'):
                    del p._svol_sink


class FlatAdamW:
    """torch.optim.AdamW (decoupled weight decay, amsgrad off; the reference's optimizer, train.py:98-99) over the reducer's
    flat buckets: the parameters of a bucket are re-homed into ONE flat fp32 buffer (``param.data`` becomes a view, like
    ``param.grad`` already is), the moments are flat too, and a step is one streaming kernel per bucket (``svol_adamw_flat``:
    28 bytes per parameter) instead of torch's multi-tensor launches over ~150 separate tensors (0.44 ms -> 0.15 ms at the
    benchmark size).  Parameters the reducer skips (they never receive a gradient) are not touched, as in torch.

    ``lr`` may be changed between steps (``opt.lr = ...``: what torch's StepLR does to ``param_groups``)."""

    def __init__(self, reducer: 'BucketedGradAllReduce', lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.reducer = reducer
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.t = 0
        self.state = []
        for b in reducer.buckets:
            flat_g = b['flat']
            flat_p = torch.zeros_like(flat_g)
            for p, off in zip(b['params'], b['offsets']):
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + n].view_as(p)  # same layout as the gradient views of the bucket
            self.state.append({'p': flat_p, 'm': torch.zeros_like(flat_p), 'v': torch.zeros_like(flat_p)})

    @torch.no_grad()
    def step(self):
        from . import _lib
        from .ops import _ptr, _stream
        self.t += 1
        for b, st in zip(self.reducer.buckets, self.state):
            rc = _lib.lib().svol_adamw_flat(_ptr(st['p']), _ptr(b['flat']), _ptr(st['m']), _ptr(st['v']), st['p'].numel(), self.lr,
                                            self.betas[0], self.betas[1], self.eps, self.weight_decay, self.t, 1.0, _stream())
            _lib.check(rc, 'svol_adamw_flat')

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    def state_dict(self):
        return {'t': self.t, 'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': self.weight_decay,
                'm': [st['m'].clone() for st in self.state], 'v': [st['v'].clone() for st in self.state]}

    def load_state_dict(self, sd):
        self.t, self.lr = int(sd['t']), float(sd['lr'])
        for st, m, v in zip(self.state, sd['m'], sd['v']):
            st['m'].copy_(m)
            st['v'].copy_(v)


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
