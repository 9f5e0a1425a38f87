#!/usr/bin/env python3
"""Headline benchmark of the SVOL hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as  python -m torch.distributed.run --nnodes=1 --nproc-per-node N
     --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...; one rank per GPU over RCCL)

metric  : frames/sec of forward + matcher/criterion + backward (BASELINE.json), whole job over N GPUs
workload: BASELINE.json configs[1]  — B=8 videos per GPU, T=32 frames, P=196 tokens/frame (L=6272),
          d=256, 8 heads, 6 layers (6 video + 6 query blocks), N=100 queries,
          video_matcher, Din=512 features at the head boundary, bf16 MFMA
          compute with an fp32 residual stream, train mode (input dropout on), synthetic data,
          weights: module default init.  One step = zero_grad, forward, criterion (all 6 layers matched on
          device), backward (+ bucketed RCCL gradient all-reduce overlapped with backward when N > 1), and
          the AdamW step (torch.optim.AdamW's update as one kernel per gradient bucket, svol_amd.parallel.FlatAdamW; the
          per-step fp32->bf16 weight refresh is inside the timed region too).  Launches are eager (host issue ~5-6 ms/step,
          hidden behind ~18 ms of GPU work: `host_issue_ms_per_step` in the line) at every N, so the N = 1 and N > 1 numbers are the same program; --graph
          replays the step as one hipGraph instead.  --workload selects the other measured configurations (cfg4: ViT
          extractor online, cfg5: long video, encdec: enc/dec Transformer head, resnet: ResNet extractors online — frozen, or with
          --train-backbone in training mode and in the optimiser as the reference's train.py has them).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
FWD_BWD_GF_PER_FRAME = 34.07    # BASELINE.md §3 (algorithmic, fwd + 2x bwd, no recompute credit)


def _host_cores():
    """(cores this job may really use, how that was found).  The affinity mask of a pool box spans the whole host (256 logical CPUs)
    while the job's CPU share is a cgroup quota (16 for a one-GPU box): threads beyond the quota only time-slice each other
    (measured: 264 s per oracle step with 256 threads, 6 s with 16)."""
    n = os.cpu_count() or 1
    how = 'os.cpu_count'
    try:
        n, how = len(os.sched_getaffinity(0)), 'affinity mask'
    except (AttributeError, OSError):
        pass
    quota = None
    try:   # cgroup v2
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        try:   # cgroup v1
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        return quota, 'cgroup cpu quota'
    if n > 32:   # no quota visible and a mask that spans a whole host: the pool's documented share of a one-GPU job
        return 16, f'pool share (affinity mask spans {n} CPUs, no cgroup quota visible)'
    return max(1, n), how


def cpu_baseline(seconds_budget=40.0):
    """Reference CPU path timed beside the GPU number: the oracle (CPU restatement proven equal to the
    reference by the golden vectors) on a BOUNDED sample of the same workload — one of the 8 videos of
    configs[1] (B=1, T=32, P=196, d=256, 6 layers, N=100, fp32, fwd + matcher + bwd).  Videos are
    independent, so frames/s does not depend on B."""
    import torch
    from oracle import svol_oracle as O
    from svol_amd import synthetic as syn
    cores, how = _host_cores()
    torch.set_num_threads(cores)
    args = syn.cfg2_args('video_matcher')
    B, T, P = 1, 32, 196
    sd = {k: v.clone().requires_grad_(True) for k, v in syn.synth_state_dict(args, seed=1).items()}
    inp = syn.synth_inputs(args, B, T, P, seed=1)
    tg = syn.synth_targets(B, T, seed=1)
    times = []
    t_start = time.time()
    for it in range(6):   # one warm-up + five timed steps (~6 s each on the pool's 16 cores)
        for p in sd.values():
            p.grad = None
        t0 = time.time()
        O.train_step(sd, args, inp, tg)
        dt = time.time() - t0
        print(f'[bench] cpu_baseline step {it}: {dt:.1f} s', file=sys.stderr, flush=True)
        if it > 0 or dt > seconds_budget:
            times.append(dt)  # the first step also warms the allocator; it only counts when it alone spends the budget
        if time.time() - t_start > seconds_budget and times:
            break
    t = sorted(times)[len(times) // 2]
    return {'value': B * T / t, 'unit': 'frames/sec', 'cores': cores, 'cores_from': how, 'kind': 'port',
            'sample': f'oracle (CPU restatement, fp32, torch {torch.__version__}) on 1 of the 8 videos of configs[1]: '
                      f'B=1,T=32,P=196,d=256,6 layers,N=100, fwd+matcher+bwd, median of {len(times)} steps '
                      f'({t * 1e3:.0f} ms/step)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--blocks', type=int, default=5,
                    help='timed blocks of exactly --steps steps run back to back; the line reports the median block (ms_per_step_blocks lists all)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--loss-scale', type=float, default=None,
                    help='static loss scale (default: 1024 with --dtype fp16 — fp16 gradients of ~1e-6 underflow otherwise; the reference '
                         'uses apex amp dynamic scaling for fp16 — else 1); folded back out inside the AdamW kernel')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--max-inflight', type=int, default=2,
                    help='steps the host may run ahead of the GPU (svol_amd.parallel.StepFence; 0 = unbounded)')
    ap.add_argument('--graph', action='store_true',
                    help='N=1 only: replay the whole step as one captured hipGraph (removes the ~15 ms/step of host issue time, '
                         'but a captured graph serialises the query-stream / video-stream overlap: measured slower)')
    ap.add_argument('--no-graph', action='store_true', help='accepted for compatibility (eager is the default)')
    ap.add_argument('--opt-in-backward', action='store_true',
                    help='lab: FlatAdamW(zero_grads=True, step_in_backward=True) — bucket updates issued during backward, buckets zeroed by the update kernel')
    ap.add_argument('--batch', type=int, default=None, help='videos per GPU (default: 8, cfg5: 1)')
    ap.add_argument('--dropout', type=float, default=None,
                    help="--workload encdec: the Transformer's dropout (reference default 0.1; the bench default stays 0 = round 3's line)")
    ap.add_argument('--train-backbone', action='store_true',
                    help='--workload resnet: the ResNet-34 / ResNet-18 extractors in TRAINING mode (batch-statistics BatchNorm, convolution '
                         'gradients) and in the optimiser, as the reference\'s train.py:72 has them; default: frozen extractors')
    ap.add_argument('--workload', default='cfg2', choices=['cfg2', 'cfg4', 'cfg5', 'encdec', 'resnet'],
                    help='cfg2 = the BASELINE metric workload (default); cfg4 = cfg2 with the ViT-B/16 frame + sketch feature '
                         'extractor run online in front of the head (BASELINE configs[3], end-to-end frames/s); '
                         'cfg5 = long-video stress case T=128, P=256 (bf16); encdec = the cfg2 shapes through the 6+6 enc/dec Transformer '
                         '(svanet_variants, append_to_seq) instead of the cross-modal transformer; resnet = the reference\'s default backbone '
                         '(frozen ResNet-34 on the frames -> 49 tokens each, ResNet-18 on the sketch) online in front of the head')
    a = ap.parse_args()

    if a.gpus > 1 and 'RANK' not in os.environ and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, through
        # torch.distributed.run as a CHILD process.  Nothing in this parent has touched the GPU (no HIP call, no
        # torch.cuda.is_available()), it never re-execs, it only forwards the child's output and exit code.
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
        s_.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        rc = subprocess.run(cmd, env=env).returncode
        sys.exit(rc if rc else 0)

    import torch
    import torch.distributed as dist
    from svol_amd import ops, parallel
    from svol_amd import synthetic as syn
    from svol_amd.modeling.loss import build_loss
    from svol_amd.modeling.svanet import build_svanet

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU fallback)'
    # one rank per GPU; SVOL_DIST_BACKEND=gloo lets several ranks share one card for a functional rehearsal of the
    # N > 1 path on a one-GPU box (RCCL refuses two ranks on the same device)
    backend = os.environ.get('SVOL_DIST_BACKEND', 'nccl')
    local_dev = local_rank % torch.cuda.device_count() if backend != 'nccl' else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    force_ar = world == 1 and os.environ.get('SVOL_FORCE_ALLREDUCE') == '1'
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('SVOL_DP_SPANS', '1')   # per-bucket (launch -> done) spans on the communication stream, exposed wait in finish()
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif force_ar:
        # a ONE-rank RCCL communicator: the bucketed all-reduce really runs on the communication stream (svol_amd.parallel: force), so the
        # stream choreography of the N > 1 path executes against RCCL on a one-GPU box; the numbers of this mode are not the N = 1 line
        import socket
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
        sk.close()
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1, device_id=dev)
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}'
    cores_rank = None
    if world > 1:
        # N Python issuers share one host: give every rank its own slice of the job's cores (main thread + autograd thread), so that
        # a rank's launches are not descheduled behind another rank's — and report each rank's issue time (host_issue_ms_per_rank)
        try:
            aff = sorted(os.sched_getaffinity(0))
            per = max(1, len(aff) // world)
            mine = aff[(local_rank * per) % len(aff):(local_rank * per) % len(aff) + per]
            os.sched_setaffinity(0, set(mine))
            cores_rank = len(mine)
        except (AttributeError, OSError):
            pass
    n_ranks_seen = 1
    if world > 1:  # every rank of the job really is in the communicator (RCCL / gloo): sum of ones
        ones = torch.ones((1,), device=dev)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
        assert n_ranks_seen == world, (n_ranks_seen, world)

    T, P = (128, 256) if a.workload == 'cfg5' else (32, 49 if a.workload == 'resnet' else 196)
    B = a.batch if a.batch is not None else (1 if a.workload == 'cfg5' else 8)
    args = syn.head_args(num_frames=T) if a.workload == 'cfg5' else syn.cfg2_args('video_matcher')
    if a.workload == 'cfg4':
        args.input_vid_dim = args.input_skch_dim = 768  # ViT-B/16 features (backbone.py:124-125)
    if a.workload == 'encdec':
        # DETR-style 6 + 6 encoder / decoder, post-norm, ReLU FFN (--dim_feedforward default 1024); the sketch token is
        # prepended to the video tokens.  --dropout 0.1 = the reference's default training step (attention-probability, residual and
        # FFN dropouts in the kernels); the default stays 0 so that this line is comparable with round 3's
        from svol_amd.modeling.svanet_variants import build_svanet as build_svanet
        args = syn.encdec_args(hidden_dim=256, nheads=8, num_queries=100, num_frames=T, enc_layers=6, dec_layers=6,
                               dim_feedforward=1024, dropout=(a.dropout if a.dropout is not None else 0.0), pre_norm=False, mode='append_to_seq', feat_dim=512,
                               matcher='video_matcher', num_layers=6)
        args.input_vid_dim = args.input_skch_dim = args.feat_dim
    args.compute_dtype = a.dtype
    torch.manual_seed(1)  # reference default seed (configs.py:17): identical initial weights on every rank
    model = build_svanet(args).to(dev).train()
    crit = build_loss(args).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]   # the optimizer's list, reference order (train.py:72)
    backbone = None
    bb_params = []
    if a.workload == 'resnet':
        # randomly initialised torchvision-layout ResNets (IMAGENET1K_V1 cannot be downloaded here), synthetic pixels; the extractors run
        # inside the timed step (backbone.py:133-152: ResNet-34 frames, ResNet-18 + avgpool sketch); frozen unless --train-backbone
        from svol_amd.modeling.resnet import ResNetBackbone, resnet18, resnet34
        backbone = ResNetBackbone(resnet34(compute_dtype=a.dtype, trainable=a.train_backbone),
                                  resnet18(avgpool=True, compute_dtype=a.dtype, trainable=a.train_backbone))
        backbone.video_backbone.load_state_dict(syn.synth_resnet_state_dict(syn.resnet_param_shapes((3, 4, 6, 3)), seed=1))
        backbone.sketch_backbone.load_state_dict(syn.synth_resnet_state_dict(syn.resnet_param_shapes((2, 2, 2, 2)), seed=2))
        backbone = backbone.to(dev)
        backbone = backbone.train() if a.train_backbone else backbone.eval()
        bb_params = [p for p in backbone.parameters() if p.requires_grad]
        params = bb_params + params   # (backbone first: model.py:14 registers it before the head)
    # gradient buckets in the order backward produces them (heads, layers 5..0, query embedding + input projections, then the
    # backbone from its last stage to the stem)
    reducer = parallel.BucketedGradAllReduce(parallel.arrival_order(model) + list(reversed(bb_params)), skip=parallel.unused_parameters(model),
                                             ordered=True)
    use_graph = world == 1 and a.graph and not a.no_graph
    if use_graph:  # (the captured step keeps torch's capturable optimizer: its step counter lives on the device)
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-4, fused=True, capturable=True)  # train.py:98-99
    else:  # the same update as one streaming kernel per gradient bucket (svol_amd.parallel.FlatAdamW)
        # --opt-in-backward (lab, off by default): zero_grads = optimizer.zero_grad() of the next iteration (train.py:222) folded
        # into the update kernel; step_in_backward = a bucket's update right behind its gradients (and its all-reduce) instead of
        # all six behind the whole backward — same arithmetic (bit-identical parameters in the deterministic mode), the step
        # boundary 0.95 -> 0.85 ms on one clock and the STEP +-0: the updates then run beside memory-bound kernels of the
        # backward, whose time grows by what the boundary lost
        opt = parallel.FlatAdamW(reducer, lr=1e-4, weight_decay=1e-4, params=params, zero_grads=a.opt_in_backward,
                                 step_in_backward=a.opt_in_backward)
    # weak scaling: every rank gets its own B videos (different seeds = a properly sharded global batch)
    inp = {k: v.to(dev) for k, v in syn.synth_inputs(args, B, T, P, seed=1 + rank).items()}
    tg = syn.synth_targets(B, T, seed=1 + rank)
    wd = crit.weight_dict
    n_dec = args.dec_layers if a.workload == 'encdec' else args.num_layers  # decoder layers whose outputs the criterion matches
    if a.workload == 'cfg4':
        # frozen, randomly initialised ViT-B/16 extractors (the pretrained weights cannot be downloaded here); synthetic
        # normalised pixel values; the extractor runs inside the timed step
        from svol_amd.modeling.backbone import ViTBackbone, ViTExtractor, vit_base_config
        backbone = ViTBackbone(ViTExtractor(vit_base_config()), ViTExtractor(vit_base_config())).to(dev).eval()
        g = torch.Generator(device=dev).manual_seed(1 + rank)
        pix_video = torch.randn((B, T, 3, 224, 224), device=dev, generator=g)
        pix_sketch = torch.randn((B, 1, 3, 224, 224), device=dev, generator=g)
    if a.workload == 'resnet':
        g = torch.Generator(device=dev).manual_seed(1 + rank)
        pix_video = torch.randn((B, T, 3, 224, 224), device=dev, generator=g)
        pix_sketch = torch.randn((B, 1, 3, 224, 224), device=dev, generator=g)

    # fp16 operands: dynamic loss scaling with overflow skip on the device (svol_amd.parallel.DynamicLossScaler; the reference's fp16 mode
    # is apex amp); --loss-scale X pins a static scale instead.  bf16 / fp32: none.
    scaler = None
    loss_scale = a.loss_scale if a.loss_scale is not None else 1.0
    if not use_graph:
        if a.dtype == 'fp16' and a.loss_scale is None:
            scaler = opt.scaler = parallel.DynamicLossScaler(dev, init_scale=2.0 ** 12)
        else:
            opt.loss_scale = loss_scale
    elif loss_scale != 1.0:
        raise SystemExit('--graph keeps torch.optim.AdamW, which does not unscale: no --loss-scale with --graph')

    fence = parallel.StepFence(a.max_inflight) if a.max_inflight > 0 else None

    turn_sleep_cycles = int(float(os.environ.get('SVOL_BENCH_TURN_SLEEP_US', '0')) * 2309.0)   # (torch.cuda._sleep spins on the shader clock: 1e6 cycles = 433 us measured, tools/lab_power_bound.sh)

    def step():
        reducer.zero_grad()
        if backbone is not None:
            inp['src_sketch'], inp['src_video'] = backbone(pix_sketch, pix_video)
        crit.prepack(tg, n_dec, B, args.num_queries, dev)  # target flattening + H2D copies issued before the forward's kernels
        out = model(inp['src_sketch'], inp['src_sketch_mask'], inp['src_video'], inp['src_video_mask'])
        ld = crit(out, tg)
        loss = crit.weighted_total()  # = sum(ld[k] * wd[k] for k in ld if k in wd) (train.py:227-228), one multiply + one reduction
        if turn_sleep_cycles:   # (lab: an idle gap of known length on the forward -> backward dependency chain)
            torch.cuda._sleep(turn_sleep_cycles)
        (scaler.scale(loss) if scaler is not None else (loss * loss_scale if loss_scale != 1.0 else loss)).backward()
        reducer.finish(mean=use_graph)   # the 1 / world of the gradient mean rides FlatAdamW's update kernel
        opt.step()
        return loss

    if use_graph:
        # N = 1: the whole step is captured once in a hipGraph (svol_amd/graph.py); every timed step still
        # re-stages the inputs and re-flattens the targets on the host, then replays
        from svol_amd.graph import GraphedTrainStep
        gstep = GraphedTrainStep(model, crit, opt, reducer, inp, tg)
        eager_step = step

        def step():  # noqa: F811
            return gstep(inp, tg)[0]

    raw_step = step

    def step():  # noqa: F811
        out_ = raw_step()
        if fence is not None:
            fence.tick()
        return out_

    if os.environ.get('SVOL_MAIN_PRIO') is not None:   # lab: the step's main stream above the side streams in queue priority
        hp = torch.cuda.Stream(priority=int(os.environ['SVOL_MAIN_PRIO']))
        hp.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(hp)
    for _ in range(a.warmup):
        loss = step()
    if not use_graph:
        ops.timer.enable(['attn_fwd', 'attn_bwd'])
    trace = os.environ.get('SVOL_BENCH_TRACE') == '1'   # dev aid: stack of any step whose host side takes > 60 ms
    if trace:
        import faulthandler

    def timed_block():
        """EXACTLY --steps steps between barrier + synchronize on both sides; returns (seconds — max over ranks —, host seconds until
        the last step was ISSUED, per-step host wall times)."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = []
        for _ in range(a.steps):
            ts = time.perf_counter()
            if trace:
                faulthandler.dump_traceback_later(0.06, repeat=False)
            step()
            if trace:
                faulthandler.cancel_dump_traceback_later()
            host.append((time.perf_counter() - ts) * 1e3)
        issued = time.perf_counter() - t0   # the host has ISSUED every step; the GPU is still running them
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, issued, host

    # the timed region of the contract, --blocks times back to back (default 5): one 20-step block is 0.35 s, and box-to-box /
    # run-to-run spread (+-3 %) is larger than a round's gains — the line reports the MEDIAN block and lists all of them
    timed = [timed_block() for _ in range(max(1, a.blocks))]
    order = sorted(range(len(timed)), key=lambda i: timed[i][0])
    elapsed, t_issued, host_ms = timed[order[len(order) // 2]]
    if trace:
        print('host_ms per step:', [round(x, 1) for x in host_ms], file=sys.stderr)
    ops.timer.disable()

    def clock_block(probe):
        """the shader clock the chip holds under this step: --steps more steps, UNTIMED, with a one-wave probe on
        its own stream sampling shader-clock ticks against the constant wall counter in 200 us windows (csrc/clock_probe.hip).
        Box-to-box spread of the headline on this pool is +-2 %, mostly this number."""
        import ctypes
        from svol_amd import _lib
        if not probe:                # (the other ranks of a multi-GPU run: the same steps — they hold collectives —, no probe)
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            return None
        cap = 8192
        samples = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        stop = torch.zeros(1, dtype=torch.int32, device=dev)
        khz = ctypes.c_int32(0)
        ps = torch.cuda.Stream()
        torch.cuda.synchronize()
        _lib.check(_lib.lib().svol_clock_probe(samples.data_ptr(), count.data_ptr(), cap, stop.data_ptr(), 200, 20000, ctypes.byref(khz),
                                               ps.cuda_stream), 'svol_clock_probe')
        try:
            for _ in range(a.steps):
                step()
        finally:
            stop.fill_(1)            # stream-ordered behind the steps: the probe leaves at its next window
            torch.cuda.synchronize()
        n = int(count.item())
        if n < 8:
            return None
        w = samples[:2 * n].view(n, 2).cpu().double()
        ghz = (w[:, 0] / w[:, 1] * (khz.value / 1e6)).sort().values
        pick = lambda q: round(float(ghz[min(n - 1, int(q * n))]), 3)
        return {'mean': round(float(ghz.mean()), 3), 'p10': pick(0.1), 'median': pick(0.5), 'p90': pick(0.9), 'windows': n, 'window_us': 200,
                'how': 's_memtime / s_memrealtime in one probe wave beside an untimed block of --steps steps right after the timed blocks'}

    sclk = None
    if not use_graph and os.environ.get('SVOL_NO_CLOCK_PROBE') is None:
        if rank == 0 and world == 1:
            try:
                sclk = clock_block(True)
            except Exception as e:   # a measurement aid must not cost the bench line
                sclk = {'error': str(e)[:200]}
        else:                        # (multi-GPU: no swallowing — a rank that skipped steps would leave the others in a collective)
            sclk = clock_block(rank == 0)
    # host time to ISSUE one step, measured on an EMPTY queue right after the timed region (3 free-running steps): inside the timed
    # region the host runs ahead until the HIP queue pushes back (a few steps), after which a step's host wall time is the GPU's
    free_ms = []
    for _ in range(3):
        ts = time.perf_counter()
        raw_loss = raw_step()   # (unfenced: this measures issue time alone)
        free_ms.append((time.perf_counter() - ts) * 1e3)
    torch.cuda.synchronize()
    if use_graph:
        # per-kernel durations cannot be bracketed inside a graph replay: time the SAME kernels with HIP events on
        # the launch stream over eager steps run right after the timed region (same process, same shapes, same data)
        crit.static_packed = None
        model.step_dev = None
        ops.timer.enable(['attn_fwd', 'attn_bwd'])
        for _ in range(max(3, min(a.steps, 10))):
            eager_step()
        torch.cuda.synchronize()
        ops.timer.disable()
    # the same two attention launches ALONE (nothing else on the device), right after the timed region: inside the step the query
    # stream's and the weight-gradient stream's kernels run beside them
    alone = None
    if rank == 0 and a.dtype != 'fp32':
        try:
            L_ = T * P + (1 if a.workload == 'encdec' else 0)
            Hh, Dm = args.nheads, args.hidden_dim
            dh_ = Dm // Hh
            tdt = {'bf16': torch.bfloat16, 'fp16': torch.float16}[a.dtype]
            qkv = (torch.randn(B * L_, 3 * Dm, device=dev) * 0.5).to(tdt)
            q_, k_, v_ = qkv[:, :Dm], qkv[:, Dm:2 * Dm], qkv[:, 2 * Dm:]
            pm = 1.4426950408889634 / dh_ ** 0.5
            o_, lse_ = ops.attn_fwd(q_, k_, v_, B, Hh, L_, L_, dh_, None, pm)
            do_ = (torch.randn(B * L_, Dm, device=dev) * 0.5).to(tdt)
            dqkv = torch.empty_like(qkv)

            def _time(fn, n=10):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n
            alone = {'attn_fwd_ms': _time(lambda: ops.attn_fwd(q_, k_, v_, B, Hh, L_, L_, dh_, None, pm)),
                     'attn_bwd_ms': _time(lambda: ops.attn_bwd(q_, k_, v_, o_, do_, lse_, B, Hh, L_, L_, dh_, dqkv[:, :Dm], dqkv[:, Dm:2 * Dm],
                                                               dqkv[:, 2 * Dm:], None, pm))}
            del qkv, o_, lse_, do_, dqkv
        except Exception as e:  # a measurement aid must not take the bench line down
            alone = {'error': repr(e)}
    ar_report = reducer.allreduce_report() if (world > 1 or force_ar) else None   # last issued step (the device is idle: events are final)
    issue_per_rank = None
    if world > 1:
        mine_t = torch.tensor([sorted(free_ms)[1]], device=dev, dtype=torch.float64)
        allt = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(allt, mine_t)
        issue_per_rank = [round(float(x.item()), 2) for x in allt]
    final_loss = float(raw_loss.detach())
    assert final_loss == final_loss, 'loss is NaN'

    ms_per_step = elapsed / a.steps * 1e3
    frames = B * T * world
    fps = frames / (elapsed / a.steps)

    # ---- roofline of the dominant kernel: video self-attention (L x L, 67 % of the layer's FLOPs) ----
    L = T * P + (1 if a.workload == 'encdec' else 0)
    dh = args.hidden_dim // args.nheads
    summ = ops.timer.summary()
    fwd = summ.get(('attn_fwd', (B, args.nheads, L, L, dh)))
    bwd = summ.get(('attn_bwd', (B, args.nheads, L, L, dh)))
    attn_fwd_flop = 4.0 * L * L * args.hidden_dim * B          # QK^T + PV  (4 L^2 d per sample)
    # SURVEY.md §8d table; cfg4 adds the extractor forward: 2*197*(768*768*4 + 2*768*3072)*12 + attention ~ 35.1 GF per
    # image, (B*T + B) images per B*T frames
    gf_frame = {'cfg2': FWD_BWD_GF_PER_FRAME, 'resnet': 0.0,  # (no closed form quoted: conv FLOPs 7.3 GF/frame + the head at L = 1568)
                'cfg5': 169.3, 'cfg4': FWD_BWD_GF_PER_FRAME * 1.24 + 35.1 * (T + 1) / T,
                # enc/dec, per sample forward: input proj + 6 x (8Ld^2 + 4L^2d + 4LdF) encoder + 6 x (4Ld^2 K/V proj + 4NLd) decoder
                # memory side (query-side terms < 1 %), x3 for fwd+bwd, / T frames; L = T*P + 1, F = 1024
                'encdec': 3.0 * (2 * L * 512 * 256 + 2 * L * 256 * 256 + 6 * (8 * L * 256 ** 2 + 4 * L * L * 256 + 4 * L * 256 * 1024)
                                 + 6 * (4 * L * 256 ** 2 + 4 * 100 * L * 256)) / T / 1e9}[a.workload]
    roof = None
    if fwd and bwd:
        # backward = 2x forward algorithmically (dV, dP, dQ, dK products; the S recompute gets no credit)
        which = 'attn_bwd (row constants + single-pass key-stationary kernel + dQ rounding)' if bwd[1] >= fwd[1] else 'attn_fwd'
        flop, ms = (2.0 * attn_fwd_flop, bwd[1]) if bwd[1] >= fwd[1] else (attn_fwd_flop, fwd[1])
        ach = flop / (ms * 1e-3) / 1e12
        # memory-side bytes per launch of the dominant kernel(s): from the NEWEST committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # passes of this very command (profiles/roundN_hbm_traffic.json, which records the git head it was measured at; separate
        # counter runs cannot happen inside the timed process)
        traffic = traffic_src = None
        if a.workload == 'cfg2' and B == 8:
            try:
                import glob
                import re
                pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')
                files = sorted((f_ for f_ in glob.glob(os.path.join(pdir, 'round*_hbm_traffic.json'))
                                if re.fullmatch(r'round\d+_hbm_traffic\.json', os.path.basename(f_))),   # (not the cfg5 file)
                               key=lambda f_: int(re.search(r'round(\d+)_', os.path.basename(f_)).group(1)))
                tj = json.load(open(files[-1]))
                meta = tj.pop('_meta', {})
                pick = (['attn_bwd_dq_bf16_pre|', 'attn_bwd_dq_bf16_rot|', 'attn_bwd_dkdv_bf16_pre|', 'attn_bwd_dkdv_bf16_pre_dma|', 'attn_delta_bf16|',
                         'attn_bwd_sp_prep_bf16|', 'attn_bwd_sp_bf16|', 'attn_dq_round_bf16|', 'attn_sp_zero_image|'] if bwd[1] >= fwd[1]
                        else ['attn_fwd_bf16_pre|', 'attn_fwd_bf16_fast|'])
                tot = sum((v['read_MB'] + v['write_MB']) * 1048576.0 for k_, v in tj.items() if any(k_.startswith(q_) for q_ in pick))
                if tot > 0:
                    traffic = tot
                    traffic_src = ('profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch; measured at git %s)'
                                   % (os.path.basename(files[-1]), meta.get('git_head', 'n/a: round-1 file')))
            except (OSError, ValueError, KeyError, IndexError, AttributeError):
                pass
        # at d_h = 32 the attention core is bound by instruction ISSUE (softmax exp / VALU beside the MFMAs), not by the
        # matrix pipe.  Round 5 settled the model with a fillers-per-gap micro-benchmark (profiles/round5_mfma_fillers.md): an MFMA
        # 32x32x16 holds the SIMD's issue for 8 of its 32 cycles and a gap runs max(32, 8 + sum of issue costs): v_exp 8, v_cvt_pk 5,
        # v_mul / v_add 4 (measured; the guide's constants) — round 4's additive 594-cycle floor is gone.  Per 32x32 score block:
        #   forward  : 4 MFMAs x 8 + 16 exp x 8 + 8 cvt x 5 + 16 row-sum adds x 4 + LDS reads ~ 309 issue cycles (pipe: 128)
        #   backward : 10 MFMAs x 8 + 16 exp x 8 + 16 cvt x 5 + 16 mul x 4 = 352 issue cycles: the floor of ANY schedule of the single-pass
        #              algorithm (pipe: 320); with ONE wave per SIMD the dS round trip through LDS cannot hide (4 ds_write_b64 x 13 +
        #              4 transposed reads x 5, measured first-in-gap prices): 424.
        # Priced at the clock the attention kernels HOLD under load (2.08 GHz: s_memtime against s_memrealtime, round-4 lab) and,
        # as separate fields, at the 2.4 GHz maximum.
        CLK_GHZ, CLK_MAX_GHZ, SIMDS = 2.08, 2.4, 256 * 4
        FWD_ISSUE, BWD_ISSUE, BWD_ISSUE_1WAVE = 309, 352, 424
        blocks = B * args.nheads * (L / 32.0) ** 2
        floor_ms = lambda cyc, ghz: blocks * cyc / (SIMDS * ghz * 1e9) * 1e3
        floor_fwd_ms, floor_bwd_ms = floor_ms(FWD_ISSUE, CLK_GHZ), floor_ms(BWD_ISSUE, CLK_GHZ)
        issue = {'model': 'measured (profiles/round5_mfma_fillers.md): a gap = max(32, 8 + sum of issue costs), exp 8, cvt_pk 5, mul / add 4; per '
                          '32x32 block: fwd %d issue cycles per SIMD, bwd (single pass) %d for any schedule, %d with one wave per SIMD (dS round '
                          'trip through LDS not hidden); %d SIMDs at the %.2f GHz the kernels hold' % (FWD_ISSUE, BWD_ISSUE, BWD_ISSUE_1WAVE, SIMDS, CLK_GHZ),
                 'model_version': 'round5-measured',
                 'fwd_floor_ms': floor_fwd_ms, 'bwd_floor_ms': floor_bwd_ms, 'bwd_floor_ms_one_wave_per_simd': floor_ms(BWD_ISSUE_1WAVE, CLK_GHZ),
                 'fwd_floor_ms_at_2p4GHz': floor_ms(FWD_ISSUE, CLK_MAX_GHZ), 'bwd_floor_ms_at_2p4GHz': floor_ms(BWD_ISSUE, CLK_MAX_GHZ),
                 'fwd_frac_of_issue_floor': floor_fwd_ms / fwd[1], 'bwd_frac_of_issue_floor': floor_bwd_ms / bwd[1],
                 'bwd_frac_of_one_wave_floor': floor_ms(BWD_ISSUE_1WAVE, CLK_GHZ) / bwd[1]}
        # algorithmic bytes of one launch: backward reads q, k, v, o, dO and writes dq, dk, dv once; forward reads q, k, v, writes o
        alg_bytes = (8 if bwd[1] >= fwd[1] else 4) * B * L * args.hidden_dim * 2.0
        roof = {'bound': 'mfma', 'kernel': which, 'achieved': ach, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / PEAK_BF16_MFMA_TFLOPS, 'traffic': traffic, 'traffic_unit': 'bytes per launch (HBM side)',
                'traffic_ratio': (traffic / alg_bytes) if traffic else None, 'algorithmic_bytes': alg_bytes,
                'traffic_source': traffic_src,
                'launch_ms': ms, 'launches_timed': bwd[0] if bwd[1] >= fwd[1] else fwd[0],
                'attn_fwd_ms': fwd[1], 'attn_bwd_ms': bwd[1],
                'attn_fwd_tflops': attn_fwd_flop / (fwd[1] * 1e-3) / 1e12,
                'whole_step_tflops': fps * gf_frame / 1e3 / world,
                'whole_step_frac_of_peak': fps * gf_frame / 1e3 / world / PEAK_BF16_MFMA_TFLOPS,
                'issue_bound_frac': (floor_bwd_ms / bwd[1]) if bwd[1] >= fwd[1] else (floor_fwd_ms / fwd[1]), 'issue_bound': issue}
        if alone and 'attn_bwd_ms' in alone:
            # (the in-step figures above are the contract's; these say what the co-scheduled weight-gradient GEMMs cost the launch)
            roof['same_launches_alone'] = {
                'attn_fwd_ms': alone['attn_fwd_ms'], 'attn_bwd_ms': alone['attn_bwd_ms'],
                'frac': (2.0 * attn_fwd_flop / (alone['attn_bwd_ms'] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS) if bwd[1] >= fwd[1]
                else (attn_fwd_flop / (alone['attn_fwd_ms'] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS),
                'issue_bound_frac': (floor_bwd_ms / alone['attn_bwd_ms']) if bwd[1] >= fwd[1] else (floor_fwd_ms / alone['attn_fwd_ms']),
                'note': 'measured right after the timed region with nothing else on the device; in the step the query stream and the '
                        'weight-gradient stream run beside these launches (the weight-gradient gate of round 3 is off since round 4)'}
        elif alone:
            roof['same_launches_alone'] = alone

    if rank == 0:
        res = {
            'metric': 'frames/sec (fwd+matcher+bwd), T=%d·P=%d·d=256' % (T, P),
            'value': fps, 'unit': 'frames/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms_per_step, 'ms_per_step_blocks': [round(b_[0] / a.steps * 1e3, 4) for b_ in timed], 'sclk_ghz': sclk,
            'timed_blocks': ('%d blocks of exactly --steps steps, each between barrier + synchronize, max over ranks; value / ms_per_step = the '
                             'median block' % len(timed)),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': ('enc/dec Transformer head (svanet_variants append_to_seq, 6 + 6 layers, post-norm, F=1024) on the '
                                    'BASELINE configs[1] shapes' if a.workload == 'encdec' else 'BASELINE configs[%d]: SVANet head' %
                                    {'cfg2': 1, 'cfg4': 3, 'cfg5': 4, 'resnet': 1}[a.workload]) + (
                                   ' + Hungarian/GIoU criterion, B=%d/GPU, T=%d, P=%d, '
                                   'd=256, h=8, 6 layers, N=100, video_matcher, Din=%d, train mode; step = fwd + '
                                   'criterion + bwd (+RCCL grad all-reduce) + AdamW' % (B, T, P, args.input_vid_dim)) + (
                                       '; ViT-B/16 extractor (random init, frozen) on all %d frames + %d sketches inside the step' % (B * T, B)
                                       if a.workload == 'cfg4' else (('; ResNet-34 (frames, 7x7 tokens) + ResNet-18 (sketch) extractors inside the step, ' + ('TRAINED (batch-statistics BatchNorm, in the optimiser)' if a.train_backbone else 'frozen')) if a.workload == 'resnet' else '')),
                       'global_batch': B * world, 'parallelism': f'dp{world}'},
            'n_ranks_seen': n_ranks_seen, 'dist_backend': backend if world > 1 else None,
            'final_loss': final_loss,
            # host time to ISSUE a step (Python + launches, no sync): when it approaches ms_per_step the run is host-bound
            'host_issue_ms_per_step': sorted(free_ms)[1],            # median of 3 steps issued into an empty queue (see above)
            'host_wall_ms_per_step_in_timed_region': t_issued / a.steps * 1e3,   # includes queue back-pressure
            'host_wall_ms_first_last': [round(x, 2) for x in host_ms[:3] + host_ms[-3:]],
            'launch_mode': 'hipGraph replay (whole step captured)' if use_graph else 'eager',
        }
        if ar_report is not None:
            # rank 0's last step: what finish() had to wait for behind backward, and every bucket's span on the communication stream
            res['allreduce_exposed_ms'] = ar_report['exposed_ms']
            res['allreduce_buckets'] = ar_report['buckets']
            res['allreduce_backend'] = 'nccl (RCCL), forced at world size 1' if force_ar else backend
            res['allreduce_launch_order'] = [b_['bucket'] for b_ in ar_report['buckets']]   # = 0, 1, 2, ...: buckets fire in arrival order
            if 'spin_model' in ar_report:
                # SVOL_ALLREDUCE_SPIN: every bucket's collective is followed by a spin kernel of its modelled 8-rank xGMI duration
                res['allreduce_spin_model'] = ar_report['spin_model']
                res['allreduce_modelled_ms'] = ar_report['modelled_ms']
        if issue_per_rank is not None:
            res['host_issue_ms_per_rank'] = issue_per_rank
            res['host_cores_per_rank'] = cores_rank
        if roof:
            res['roofline'] = roof
        if world == 1 and not a.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
        print(json.dumps(res), flush=True)
    if world > 1 or force_ar:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
