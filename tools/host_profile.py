#!/usr/bin/env python3
"""Where does the HOST spend its time issuing one training step?  cProfile over free-running steps of bench.py's cfg2 workload
(no synchronisation inside the profiled region, so the numbers are issue time, not GPU time).  `python tools/host_profile.py [steps]`."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from svol_amd import parallel  # noqa: E402
from svol_amd import synthetic as syn  # noqa: E402
from svol_amd.modeling.loss import build_loss  # noqa: E402
from svol_amd.modeling.svanet import build_svanet  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda', 0)
    args = syn.cfg2_args('video_matcher')
    args.compute_dtype = 'bf16'
    B, T, P = 8, 32, 196
    torch.manual_seed(1)
    model = build_svanet(args).to(dev).train()
    crit = build_loss(args).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    reducer = parallel.BucketedGradAllReduce(parallel.arrival_order(model), skip=parallel.unused_parameters(model), ordered=True)
    opt = parallel.FlatAdamW(reducer, lr=1e-4, weight_decay=1e-4, params=params)
    inp = {k: v.to(dev) for k, v in syn.synth_inputs(args, B, T, P, seed=1).items()}
    tg = syn.synth_targets(B, T, seed=1)

    def step():
        reducer.zero_grad()
        crit.prepack(tg, args.num_layers, B, args.num_queries, dev)
        out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
        crit(out, tg)
        loss = crit.weighted_total()
        loss.backward()
        reducer.finish()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # per-Function host time, forward and backward (backward runs on the autograd thread, which cProfile does not see)
    from svol_amd import blocks, ops
    from svol_amd.modeling import cross_modal_transformer as cmt
    acc = {}

    def wrap(cls, name):
        fn = getattr(cls, name)

        def timed(*a_, **k_):
            t_ = time.perf_counter()
            try:
                return fn(*a_, **k_)
            finally:
                e_ = acc.setdefault(f'{cls.__name__}.{name}', [0, 0.0])
                e_[0] += 1
                e_[1] += time.perf_counter() - t_
        setattr(cls, name, staticmethod(timed))

    for mod in (ops, blocks, cmt):
        for v in list(vars(mod).values()):
            if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
                wrap(v, 'forward')
                wrap(v, 'backward')
    hook_t = [0, 0.0]
    orig_make = reducer._make_hook
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    acc.clear()
    tb = [0.0]
    orig_bwd = torch.Tensor.backward

    def timed_bwd(self, *a_, **k_):
        t_ = time.perf_counter()
        r_ = orig_bwd(self, *a_, **k_)
        tb[0] += time.perf_counter() - t_
        return r_
    torch.Tensor.backward = timed_bwd
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    torch.Tensor.backward = orig_bwd
    print(f'per-Function host time over {steps} steps (ms per step | calls per step | us per call); loss.backward() total {tb[0] / steps * 1e3:.2f} ms/step')
    tot_b = 0.0
    for k_, (n_, t_) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f'  {k_:38s} {t_ / steps * 1e3:7.3f} ms  {n_ / steps:6.1f}  {t_ / n_ * 1e6:7.1f} us')
        if k_.endswith('backward'):
            tot_b += t_
    print(f'  sum of Function.backward bodies: {tot_b / steps * 1e3:.2f} ms/step -> engine + AccumulateGrad + hooks + native nodes: {(tb[0] - tot_b) / steps * 1e3:.2f} ms/step')
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'unprofiled host issue: {(t1 - t0) / steps * 1e3:.2f} ms/step   (wall incl. GPU: {(time.perf_counter() - t0) / steps * 1e3:.2f})')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    for key in ('tottime', 'cumulative'):
        print(f'\n==== top by {key} (totals over {steps} steps; divide by {steps}) ====')
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats(key).print_stats(45)


if __name__ == '__main__':
    main()
