"""Whole-step hipGraph capture of the SVOL training step (HIP graphs instead of a tracing compiler).

One eager training step enqueues ~1300 kernels from Python (autograd nodes, ctypes calls): ~23 ms of
host time per step at the benchmark size, the same order as the GPU time.  All kernels behind the
C-ABI are capture-safe (no allocation, no sync, sizes of ragged data are read from device memory), so
the whole step — gradient-bucket zeroing, per-step weight casts, forward, device-side matching of
every decoder layer, losses, backward, fused AdamW — is captured ONCE and replayed.

Per step the host only (1) copies the new inputs into the static input tensors, (2) flattens the new
targets into the fixed-capacity static target buffers (`StaticPackedTargets.load`) and (3) replays.
The SVANet input-projection dropout masks stay fresh because their seed is offset by a device-side step
counter that the graph itself increments.  The enc/dec Transformer's attention / residual / FFN dropouts
take their seeds from the host and REFUSE capture in training mode (svol_amd/modeling/transformer.py::_drop).
"""
from __future__ import annotations

import torch

from .modeling.matcher import StaticPackedTargets


class GraphedTrainStep:
    def __init__(self, model, criterion, optimizer, reducer, inputs: dict, targets, max_boxes_per_video: int = 128,
                 warmup: int = 3):
        """model: SVANet head (training mode), criterion: SetCriterion, optimizer: a CAPTURABLE torch optimizer
        (e.g. AdamW(fused=True, capturable=True)), reducer: BucketedGradAllReduce (world size 1)."""
        if reducer.world != 1:
            raise NotImplementedError('graph capture is used for single-process steps; multi-GPU runs stay eager so '
                                      'that the RCCL all-reduce overlaps with backward')
        self.model, self.criterion, self.optimizer, self.reducer = model, criterion, optimizer, reducer
        dev = inputs['src_video'].device
        self.static_in = {k: v.clone() for k, v in inputs.items()}
        B, N = inputs['src_video'].shape[0], model.num_queries
        m = criterion.matcher
        nl = model.transformer.num_layers
        self.packed = StaticPackedTargets(m.kind, nl, B, N, m.num_frames, m.num_queries_per_frame, dev,
                                          max_boxes_per_video)
        self.packed.load(targets)
        criterion.static_packed = self.packed
        model.step_dev = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.weights = criterion.weight_dict
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # the query-stream / video-stream overlap (cross_modal_transformer.py) becomes fork / join edges in a capture,
        # which the replay executes slower than one serial chain (measured 25.5 vs 24.6 ms/step): capture it serial
        # (SVOL_GRAPH_OVERLAP=1 keeps the side streams in the capture — to re-measure that on a new ROCm)
        import os
        from . import ops
        from .modeling import cross_modal_transformer as cmt
        overlap = os.environ.get('SVOL_GRAPH_OVERLAP') is not None
        keep, cmt.OVERLAP_QUERY_STREAM = cmt.OVERLAP_QUERY_STREAM, cmt.OVERLAP_QUERY_STREAM and overlap
        keep_w, ops.WGRAD_IN_CAPTURE = ops.WGRAD_IN_CAPTURE, overlap
        try:
            with torch.cuda.graph(self.graph):
                self.static_loss, self.static_losses = self._step()
        finally:
            cmt.OVERLAP_QUERY_STREAM = keep
            ops.WGRAD_IN_CAPTURE = keep_w
        torch.cuda.synchronize()

    def _step(self):
        self.reducer.zero_grad()
        self.model.step_dev.add_(1)
        out = self.model(self.static_in['src_sketch'], self.static_in['src_sketch_mask'], self.static_in['src_video'],
                         self.static_in['src_video_mask'])
        ld = self.criterion(out, None)
        loss = sum(ld[k] * self.weights[k] for k in ld.keys() if k in self.weights)
        loss.backward()
        self.reducer.finish()
        self.optimizer.step()
        return loss.detach(), {k: v.detach() for k, v in ld.items()}

    def __call__(self, inputs: dict = None, targets=None):
        """Run one training step; returns (loss, loss_dict) as STATIC tensors (overwritten by the next call)."""
        if inputs is not None:
            for k, v in inputs.items():
                self.static_in[k].copy_(v, non_blocking=True)
        if targets is not None:
            self.packed.load(targets)
        self.graph.replay()
        return self.static_loss, self.static_losses
