"""Worker of tests/test_gpu_parallel.py: one of WORLD_SIZE ranks that share ONE MI355X (gloo backend: RCCL refuses two
ranks on the same device).  Checks that BucketedGradAllReduce — gradient sinks written from two streams, bucket
all-reduces launched from autograd hooks on a communication stream — yields exactly the mean over ranks of the
per-rank gradients computed without it."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svol_amd import parallel  # noqa: E402
from svol_amd import synthetic as syn  # noqa: E402
from svol_amd.modeling.loss import build_loss  # noqa: E402
from svol_amd.modeling.svanet import build_svanet  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    full = os.environ.get('SVOL_DP_CASE') == 'full'
    if full:
        # The benchmark's widths, depth, sequence length and per-rank batch, bf16, the bench's 16 MiB buckets: the GPU lags
        # the host by many kernels, the query half runs on the side stream far ahead of the main stream, and a bucket
        # boundary falls inside a query half — the configuration in which an all-reduce launched from a side-stream hook
        # without waiting for the main stream sums a bucket before its video-half gradients have been written.
        args = syn.head_args(hidden_dim=256, nheads=8, num_layers=6, num_queries=100, num_frames=32, input_vid_dim=512,
                             input_skch_dim=512, matcher='video_matcher', compute_dtype='bf16')
        B, T, P = 8, 32, 196   # the bench's per-rank batch: ~22 ms of GPU work per step against ~15 ms of host issue, so the GPU lags
        bucket_bytes, tol = 16 << 20, 1e-2  # bf16: a reordered fp32 atomic sum flips bf16 roundings downstream (2^-9); a missed wait is O(1)
    else:
        args = syn.head_args(hidden_dim=64, nheads=8, num_layers=2, num_queries=10, num_frames=4, input_vid_dim=64,
                             input_skch_dim=64, matcher='video_matcher', compute_dtype='fp32')
        B, T, P = 2, 4, 49
        bucket_bytes, tol = 64 << 10, 1e-4
    sd = syn.synth_state_dict(args, seed=5)
    inp = {k: v.cuda() for k, v in syn.synth_inputs(args, B, T, P, seed=10 + rank, pad_frames=1).items()}
    tg = syn.synth_targets(B, T, seed=10 + rank)

    def run(use_reducer):
        model = build_svanet(args)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        crit = build_loss(args).cuda()
        red = None
        if use_reducer:
            red = parallel.BucketedGradAllReduce(parallel.arrival_order(model), bucket_bytes=bucket_bytes,
                                                 skip=parallel.unused_parameters(model), ordered=True)
            if os.environ.get('SVOL_DP_BREAK'):  # negative control (run by hand): the round-1 behaviour, wait on the hook's stream only
                red._producer_streams = lambda: [torch.cuda.current_stream()]
            red.zero_grad()
        out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
        ld = crit(out, tg)
        loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        loss.backward()
        if red is not None:
            assert len(red.buckets) > 3
            red.finish()
            # buckets complete in order: the gradient-arrival guess of parallel.arrival_order holds for this model
            spans = red.bucket_fire_spans()
            assert all(s_ is not None for s_ in spans)
            if full:  # a bucket boundary inside a query half: some bucket holds parameters of both halves of one layer
                def half(n_):
                    return 'q' if any(t in n_ for t in ('token_self_attn', 'content_token_cross_attn', 'mlp2', 'norm4',
                                                         'norm5', 'norm6')) else 'v'
                name_of = {id(p_): n_ for n_, p_ in model.named_parameters()}
                mixed = [sorted({half(name_of[id(p_)]) for p_ in b_['params'] if 'transformer.layers.' in name_of[id(p_)]})
                         for b_ in red.buckets]
                assert ['q', 'v'] in mixed, mixed
                # at the bench's bucket size the buckets complete in the order they were laid out (at the toy case's
                # 64 KiB granularity single tensors of one autograd node complete in node-internal order)
                assert all(spans[i][1] < spans[i + 1][1] for i in range(len(spans) - 1)), spans
        torch.cuda.synchronize()
        return {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}

    got = run(True)
    ref = run(False)
    worst = 0.0
    for n, g in ref.items():
        if g is None:
            continue
        r = g.clone()
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        r /= world
        scale = max(float(r.abs().max()), 1e-6)
        worst = max(worst, float((got[n] - r).abs().max()) / scale)
    # fp32 atomics reorder sums: agreement to rounding, far below any real synchronisation error (O(1))
    assert worst < tol, worst
    # and every rank ends up with the same averaged gradient
    chk = torch.stack([g.double().sum() for g in got.values() if g is not None]).sum().reshape(1)
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    assert all(abs(float(x - lst[0])) <= 1e-9 * max(1.0, abs(float(lst[0]))) for x in lst)
    print(f'rank {rank}: reducer == mean of per-rank gradients (worst rel diff {worst:.2e})', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
