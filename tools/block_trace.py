#!/usr/bin/env python3
"""In-step duration of every launch group of the block programs at the bench workload, no profiler attached:
    SVOL_BLOCK_TRACE=1 python tools/block_trace.py [steps]
(svol_block_trace_dump, include/svol_hip.h).  The events cost ~2 us each on the stream; the step runs ~1 % slower with them."""
import os
import sys
import time

os.environ.setdefault('SVOL_BLOCK_TRACE', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from svol_amd import blocks, parallel  # noqa: E402
from svol_amd import synthetic as syn  # noqa: E402
from svol_amd.modeling.loss import build_loss  # noqa: E402
from svol_amd.modeling.svanet import build_svanet  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda', 0)
    args = syn.cfg2_args('video_matcher')
    args.compute_dtype = os.environ.get('SVOL_TRACE_DTYPE', 'bf16')
    B, T, P = 8, 32, 196
    torch.manual_seed(1)
    model = build_svanet(args).to(dev).train()
    crit = build_loss(args).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    reducer = parallel.BucketedGradAllReduce(parallel.arrival_order(model), skip=parallel.unused_parameters(model), ordered=True)
    plain = os.environ.get('SVOL_TRACE_OPT_IN_BACKWARD') is None   # (bench.py --opt-in-backward)
    opt = parallel.FlatAdamW(reducer, lr=1e-4, weight_decay=1e-4, params=params, zero_grads=not plain, step_in_backward=not plain)
    inp = {k: v.to(dev) for k, v in syn.synth_inputs(args, B, T, P, seed=1).items()}
    tg = syn.synth_targets(B, T, seed=1)
    fence = parallel.StepFence(2)

    def step():
        reducer.zero_grad()
        crit.prepack(tg, args.num_layers, B, args.num_queries, dev)
        out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
        crit(out, tg)
        crit.weighted_total().backward()
        reducer.finish(mean=False)
        opt.step()
        fence.tick()

    for _ in range(4):
        step()
    blocks.trace_dump()   # forget the warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    txt = blocks.trace_dump()
    if os.environ['SVOL_BLOCK_TRACE'] == '2':   # timeline of the traced steps: start / end (ms) of every call site, all streams
        print(f'# {steps} steps, {ms:.2f} ms/step; start end (ms from the first record) program call-site')
        print(txt)
        rows = []
        for line in txt.splitlines():
            f = line.split()
            if len(f) >= 4:
                rows.append((float(f[0]), float(f[1]), f[2], ' '.join(f[3:])))
        vh = sorted(r for r in rows if r[2].startswith('video_half_fwd') or r[2].startswith('video_half_bwd'))
        if vh:
            span, busy = vh[-1][1] - vh[0][0], sum(r[1] - r[0] for r in vh)
            gaps = sorted(((vh[i + 1][0] - vh[i][1], vh[i][1], vh[i][3], vh[i + 1][3]) for i in range(len(vh) - 1)), reverse=True)
            print(f'# video-stream block programs: span {span:.3f} ms, busy {busy:.3f} ms, idle {span - busy:.3f} ms in {len(gaps)} gaps; largest:')
            for g in gaps[:5]:
                print(f'#   {g[0]:.3f} ms at {g[1]:.3f}: after {g[2][:44]} before {g[3][:44]}')
            turns = [g[0] for g in gaps if 'svol_layernorm' in g[2] and 'svol_layernorm_bwd' in g[3]]
            bounds = [g[0] for g in gaps if 'svol_gate_bwd' in g[2] and 'svol_gate_fwd' in g[3]]
            if turns and bounds:
                med = lambda v: sorted(v)[len(v) // 2]
                print(f'# forward -> backward turn (LN3 end to LN3\' start): median {med(turns):.3f} ms over {len(turns)}; '
                      f'step boundary (gate\' end to gate start): median {med(bounds):.3f} ms over {len(bounds)}')
        return
    tot = 0.0
    print(f'# {steps} steps, {ms:.2f} ms/step with the trace events; per step:')
    for line in txt.splitlines():
        head, rest = line.rsplit('calls', 1)
        f = rest.split()
        calls, avg, total = int(f[0]), float(f[2]), float(f[4])
        tot += total / steps
        print(f'{head.strip():100s} x{calls / steps:5.1f}  avg {avg:8.1f} us   {total / steps:7.3f} ms/step')
    print(f'# sum over the traced call sites (all streams): {tot:.2f} ms/step')


if __name__ == '__main__':
    main()
