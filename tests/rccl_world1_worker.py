"""Worker of tests/test_gpu_parallel.py::test_rccl_world1_forced_allreduce: ONE rank on the MI355X with a real RCCL communicator
(backend nccl, world size 1) and SVOL_FORCE_ALLREDUCE=1, so that BucketedGradAllReduce enqueues every bucket's all-reduce on its
communication stream — behind the waits on the main, query-half and weight-gradient streams — exactly as at N > 1 (train.py:120-124,
362-366 of the reference: apex DDP over NCCL).  A 1-rank sum is the identity: the gradients must equal the un-reduced run's BIT FOR BIT
wherever the producing kernels are deterministic, and to fp32-atomic reordering elsewhere; buckets complete in layout order; the
per-bucket spans on the communication stream are recorded and finish() reports what it had to wait for."""
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
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % os.environ['SVOL_PORT'], rank=0, world_size=1, device_id=dev)
    args = syn.head_args(hidden_dim=256, nheads=8, num_layers=6, num_queries=100, num_frames=32, input_vid_dim=512,
                         input_skch_dim=512, matcher='video_matcher', compute_dtype='bf16')
    B, T, P = 8, 32, 196   # BASELINE configs[1]: the step the bench times
    sd = syn.synth_state_dict(args, seed=5)
    inp = {k: v.cuda() for k, v in syn.synth_inputs(args, B, T, P, seed=10, pad_frames=1).items()}
    tg = syn.synth_targets(B, T, seed=10)

    def run(force):
        os.environ['SVOL_FORCE_ALLREDUCE'] = '1' if force else '0'
        model = build_svanet(args)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        crit = build_loss(args).cuda()
        red = parallel.BucketedGradAllReduce(parallel.arrival_order(model), bucket_bytes=16 << 20,
                                             skip=parallel.unused_parameters(model), ordered=True)
        assert red.force == force and red.world == 1
        reports = []
        for _ in range(2):          # twice: the second step runs with a warm allocator and the host a step ahead of the device
            red.zero_grad()
            out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
            ld = crit(out, tg)
            loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
            loss.backward()
            red.finish()
            torch.cuda.synchronize()
            reports.append(red.allreduce_report())
        spans = red.bucket_fire_spans()
        assert all(s_ is not None for s_ in spans) and all(spans[i][1] < spans[i + 1][1] for i in range(len(spans) - 1)), spans
        grads = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in model.named_parameters()}
        return grads, reports, len(red.buckets)

    got, rep, nb = run(True)
    ref, rep0, _ = run(False)
    assert rep0[-1]['buckets'] == [] and len(rep[-1]['buckets']) == nb, (rep0[-1], rep[-1])
    starts = [b['start_ms'] for b in rep[-1]['buckets']]
    assert [b['bucket'] for b in rep[-1]['buckets']] == list(range(nb)) and starts == sorted(starts), rep[-1]
    worst, exact, total = 0.0, 0, 0
    for n, g in ref.items():
        if g is None:
            assert got[n] is None
            continue
        total += 1
        exact += int(torch.equal(got[n], g))
        scale = max(float(g.abs().max()), 1e-6)
        worst = max(worst, float((got[n] - g).abs().max()) / scale)
    # fp32 atomics (split-M weight gradients, bias / LayerNorm reductions, the single-pass dQ image) reorder sums from run to run:
    # agreement to bf16 rounding downstream of them; a missed stream wait is O(1)
    assert worst < 1e-2, worst
    print('rccl world-1: %d buckets all-reduced on the communication stream, %d / %d tensors bit-identical, worst rel diff %.2e, '
          'exposed %.3f ms, spans %s' % (nb, exact, total, worst, rep[-1]['exposed_ms'], rep[-1]['buckets']), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
