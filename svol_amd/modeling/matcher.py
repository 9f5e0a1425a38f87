"""Set matchers (reference lib/modeling/matcher.py:12-119 PerFrameMatcher, :122-159
HungarianMatcher, :162-178 build_matcher) on the device-side batched LSAP.

The reference builds the full cross-batch cost matrix on the GPU, copies it to
the host and calls scipy per block in a Python loop.  Here the nested target
dicts are flattened ONCE per batch into :class:`PackedTargets` (host work the
reference also does, matcher.py:62-70) and the diagonal blocks are costed and
solved on the device in two launches for all decoder layers together.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch
from torch import nn

from .. import _lib, ops


class PackedTargets:
    """Flattened targets + the list of LSAP problems for `n_layers` decoder layers.

    Problem order: layer-major, then video (video_matcher) or (video, frame) (per_frame_matcher),
    i.e. exactly the order in which the reference's loops visit blocks."""

    def __init__(self, targets, matcher: str, n_layers: int, B: int, N: int, num_frames: int, q_per_frame: int,
                 device):
        boxes, per_video, per_frame = [], [], []
        for tv in targets:  # same traversal as matcher.py:62-70 / :140-148
            per_frame.extend(int(x) for x in tv['num_boxes_per_frame'])
            cnt = 0
            for frame in tv['bboxes'].values():
                cnt += len(frame)
                for inst in frame:
                    boxes.append(inst['bbox'])
            per_video.append(cnt)
        assert len(targets) == B
        tgt = torch.stack(boxes).to(torch.float32) if boxes else torch.zeros((0, 4))
        self.per_video = per_video
        self.per_frame = per_frame
        self.matcher = matcher
        self.B, self.N, self.T, self.q, self.n_layers = B, N, num_frames, q_per_frame, n_layers
        vid_off = np.concatenate([[0], np.cumsum(per_video)[:-1]]).astype(np.int64)
        if matcher == 'video_matcher':
            p_off = (np.arange(B, dtype=np.int64) * N)
            p_cnt = np.full(B, N, np.int64)
            t_off, t_cnt = vid_off, np.asarray(per_video, np.int64)
        elif matcher == 'per_frame_matcher':
            assert N == num_frames * q_per_frame  # matcher.py:56
            assert len(per_frame) == B * num_frames
            bt = np.arange(B * num_frames, dtype=np.int64)
            p_off = (bt // num_frames) * N + (bt % num_frames) * q_per_frame
            p_cnt = np.full(B * num_frames, q_per_frame, np.int64)
            t_cnt = np.asarray(per_frame, np.int64)
            t_off = np.concatenate([[0], np.cumsum(t_cnt)[:-1]]).astype(np.int64)
        else:
            raise NotImplementedError  # matcher.py:178
        P1 = len(p_off)
        lay = np.repeat(np.arange(n_layers, dtype=np.int64), P1)
        pred_off = np.tile(p_off, n_layers) + lay * (B * N)
        pred_cnt = np.tile(p_cnt, n_layers)
        tgt_off = np.tile(t_off, n_layers)
        tgt_cnt = np.tile(t_cnt, n_layers)
        sizes = pred_cnt * tgt_cnt
        cost_off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.cost_numel = int(sizes.sum())
        self.n_problems = int(len(pred_off))
        self.problems_per_layer = P1
        self.max_dim = int(max(pred_cnt.max(initial=0), tgt_cnt.max(initial=0)))
        self.total_boxes = int(sum(per_video))
        i32 = np.stack([pred_off, pred_cnt, tgt_off, tgt_cnt]).astype(np.int32)
        i32_d = torch.from_numpy(i32).to(device, non_blocking=True)
        self.pred_off, self.pred_cnt, self.tgt_off, self.tgt_cnt = i32_d[0], i32_d[1], i32_d[2], i32_d[3]
        self.cost_off = torch.from_numpy(cost_off).to(device, non_blocking=True)
        self.tgt_boxes = tgt.to(device, non_blocking=True).contiguous()
        self._flags = torch.zeros((2, self.n_problems), dtype=torch.int32, device=device)
        self.status, self.box_status = self._flags[0], self._flags[1]  # LSAP status | degenerate-box flag, per problem
        self.vid_off = vid_off
        # PerFrameMatcher hands the criterion re-based target ids (matcher.py:114-115); the loss kernel
        # reproduces that with the per-video first-box offsets
        self.rebase_vid_off = (torch.from_numpy(vid_off.astype(np.int32)).to(device, non_blocking=True)
                               if matcher == 'per_frame_matcher' else None)
        self.last_cost = None

    def check_status(self):
        """Raise what the reference raises, from the flags the kernels left (synchronises; one copy):
        AssertionError when a prediction or target box fails generalized_box_iou's early check x1 >= x0, y1 >= y0
        (box_utils.py:51-52 — the reference hits it while building the cost matrix, i.e. BEFORE scipy runs, so it
        takes precedence), then scipy's ValueError for NaN / -inf costs or an infeasible block."""
        fl = self._flags.cpu()
        if self.n_problems == 0:
            return
        if int(fl[1].max()) != 0:
            raise AssertionError('degenerate boxes: generalized_box_iou needs x1 >= x0 and y1 >= y0 for every '
                                 'prediction and target box (NaN coordinates fail the check too)')
        st = int(fl[0].max())
        if st == 1:
            raise ValueError('matrix contains invalid numeric entries')
        if st == 2:
            raise ValueError('cost matrix is infeasible')

    def indices_from_match(self, match: torch.Tensor, layer: int):
        """Reference-format result for one layer: list over videos of (pred_idx, tgt_idx) int64
        tensors (host).  Prediction ids are video-local, ascending (scipy returns rows sorted).
        video_matcher: target ids are video-local.  per_frame_matcher: batch-global box ids re-based by
        the per-video minimum MATCHED id (matcher.py:114-115, quirk preserved)."""
        m = match.view(self.n_layers, self.B, self.N)[layer].cpu().numpy()
        out = []
        for b in range(self.B):
            pi = np.nonzero(m[b] >= 0)[0].astype(np.int64)
            ti = m[b][pi].astype(np.int64)
            if self.matcher == 'video_matcher':
                ti = ti - self.vid_off[b]
            elif len(ti):
                ti = ti - ti.min()
            out.append((torch.as_tensor(pi, dtype=torch.int64), torch.as_tensor(ti, dtype=torch.int64)))
        return out


class StaticPackedTargets:
    """PackedTargets with FIXED-CAPACITY device buffers that are refilled in place (``load``), so that the
    matcher / criterion kernels can be captured in a hipGraph: the launch geometry (number of LSAP problems,
    LDS size, cost-buffer size) depends only on (matcher, layers, B, N, frames) and the capacity, while the
    per-batch box counts / offsets / coordinates live in device memory the kernels read at run time."""

    def __init__(self, matcher: str, n_layers: int, B: int, N: int, num_frames: int, q_per_frame: int, device,
                 max_boxes_per_video: int = 128):
        self.matcher, self.n_layers, self.B, self.N, self.T, self.q = matcher, n_layers, B, N, num_frames, q_per_frame
        self.device = device
        self.cap_video = int(max_boxes_per_video)
        P1 = B if matcher == 'video_matcher' else B * num_frames
        self.problems_per_layer = P1
        self.n_problems = P1 * n_layers
        if matcher == 'video_matcher':
            self.max_dim = max(N, self.cap_video)
            self.cost_numel = self.n_problems * N * self.cap_video
        else:
            self.max_dim = max(q_per_frame, self.cap_video)
            self.cost_numel = n_layers * B * N * self.cap_video  # sum over frames of q * m_t <= N * boxes per video
        self._i32 = torch.zeros((4, self.n_problems), dtype=torch.int32, device=device)
        self.pred_off, self.pred_cnt, self.tgt_off, self.tgt_cnt = self._i32[0], self._i32[1], self._i32[2], self._i32[3]
        self.cost_off = torch.zeros((self.n_problems,), dtype=torch.int64, device=device)
        self.tgt_boxes = torch.zeros((B * self.cap_video, 4), dtype=torch.float32, device=device)
        self._flags = torch.zeros((2, self.n_problems), dtype=torch.int32, device=device)
        self.status, self.box_status = self._flags[0], self._flags[1]
        self.rebase_vid_off = (torch.zeros((B,), dtype=torch.int32, device=device)
                               if matcher == 'per_frame_matcher' else None)
        self.vid_off = np.zeros(B, np.int64)
        self.last_cost = None

    def load(self, targets):
        """Flatten `targets` on the host (same traversal as PackedTargets) and refill the device buffers."""
        tmp = PackedTargets(targets, self.matcher, self.n_layers, self.B, self.N, self.T, self.q, 'cpu')
        if max(tmp.per_video) > self.cap_video:
            raise ValueError(f'a video has {max(tmp.per_video)} boxes, capacity is {self.cap_video}')
        assert tmp.n_problems == self.n_problems and tmp.cost_numel <= self.cost_numel
        self._i32.copy_(torch.stack([tmp.pred_off, tmp.pred_cnt, tmp.tgt_off, tmp.tgt_cnt]), non_blocking=True)
        self.cost_off.copy_(tmp.cost_off, non_blocking=True)
        nb = tmp.tgt_boxes.shape[0]
        self.tgt_boxes[:nb].copy_(tmp.tgt_boxes, non_blocking=True)
        if self.rebase_vid_off is not None:
            self.rebase_vid_off.copy_(tmp.rebase_vid_off, non_blocking=True)
        self.vid_off = tmp.vid_off
        self.per_video, self.per_frame, self.total_boxes = tmp.per_video, tmp.per_frame, tmp.total_boxes

    check_status = PackedTargets.check_status
    indices_from_match = PackedTargets.indices_from_match


class _DeviceMatcher(nn.Module):
    kind = None

    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1, num_frames: int = 32,
                 num_queries_per_frame: int = 10):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou
        self.num_frames, self.num_queries_per_frame = num_frames, num_queries_per_frame
        self.foreground_label = 0
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, 'all costs cant be 0'

    def pack(self, targets, n_layers, B, N, device) -> PackedTargets:
        return PackedTargets(targets, self.kind, n_layers, B, N, self.num_frames, self.num_queries_per_frame, device)

    @torch.no_grad()
    def forward(self, outputs, targets) -> List:
        """Reference contract: list over the batch of (index_i, index_j) int64 tensors."""
        lg, bx = outputs['pred_logits'], outputs['pred_boxes']
        if not lg.is_cuda:
            raise RuntimeError('svol_amd matchers run on the MI355X only (no CPU fallback)')
        B, N = bx.shape[:2]
        packed = self.pack(targets, 1, B, N, lg.device)
        match = ops.match_all(lg.float().contiguous().view(1, B, N, 2), bx.float().contiguous().view(1, B, N, 4),
                              packed, self.cost_bbox, self.cost_giou, self.cost_class)
        packed.check_status()
        return packed.indices_from_match(match, 0)


class PerFrameMatcher(_DeviceMatcher):
    kind = 'per_frame_matcher'


class HungarianMatcher(_DeviceMatcher):
    kind = 'video_matcher'

    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__(cost_class, cost_bbox, cost_giou)


def build_matcher(args):
    if args.matcher == 'per_frame_matcher':
        return PerFrameMatcher(cost_bbox=args.set_cost_bbox, cost_giou=args.set_cost_giou,
                               cost_class=args.set_cost_class, num_frames=args.num_frames,
                               num_queries_per_frame=args.num_queries_per_frame)
    elif args.matcher == 'video_matcher':
        return HungarianMatcher(cost_bbox=args.set_cost_bbox, cost_giou=args.set_cost_giou,
                                cost_class=args.set_cost_class)
    raise NotImplementedError
