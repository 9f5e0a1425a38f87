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
    args = syn.head_args(hidden_dim=64, nheads=8, num_layers=2, num_queries=10, num_frames=4, input_vid_dim=64,
                         input_skch_dim=64, matcher='video_matcher', compute_dtype='fp32')
    B, T, P = 2, 4, 49
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
            red = parallel.BucketedGradAllReduce(list(model.parameters()), bucket_bytes=64 << 10,
                                                 skip=parallel.unused_parameters(model))
            red.zero_grad()
        out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
        ld = crit(out, tg)
        loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        loss.backward()
        if red is not None:
            assert len(red.buckets) > 3
            red.finish()
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
    assert worst < 1e-4, worst
    # and every rank ends up with the same averaged gradient
    chk = torch.stack([g.double().sum() for g in got.values() if g is not None]).sum().reshape(1)
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    assert all(abs(float(x - lst[0])) <= 1e-9 * max(1.0, abs(float(lst[0]))) for x in lst)
    print(f'rank {rank}: reducer == mean of per-rank gradients (worst rel diff {worst:.2e})', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
