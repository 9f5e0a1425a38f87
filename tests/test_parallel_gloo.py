"""N>1 path on CPU: world_size-2 ``gloo`` run of the bucketed gradient all-reduce
(svol_amd/parallel.py) driving the oracle's train step on each rank's shard of the batch.

Expected semantics (reference apex DDP): the reduced gradient equals the AVERAGE of the
per-shard single-process gradients (mean of per-rank means — NOT the gradient of one global-batch loss).
"""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import svol_oracle as O
from svol_amd import parallel
from svol_amd import synthetic as syn


def _args():
    return syn.head_args(hidden_dim=32, nheads=4, num_layers=2, num_queries=8, num_queries_per_frame=2, num_frames=4,
                         input_vid_dim=32, input_skch_dim=32, matcher='video_matcher')


def _shard(x, rank, world):
    n = x.shape[0] // world
    return x[rank * n:(rank + 1) * n]


def _local_grads(rank, world, reducer_factory=None):
    args = _args()
    B, T, P = 2 * world, 4, 6
    sd = syn.synth_state_dict(args, seed=1)
    inp = syn.synth_inputs(args, B, T, P, seed=1, pad_frames=1)
    tg = syn.synth_targets(B, T, seed=1)
    params = torch.nn.ParameterDict({k.replace('.', '/'): torch.nn.Parameter(v.clone()) for k, v in sd.items()})
    psd = {k.replace('/', '.'): p for k, p in params.items()}
    n = B // world
    my_inp = {k: _shard(v, rank, world) for k, v in inp.items()}
    my_tg = tg[rank * n:(rank + 1) * n]
    red = reducer_factory(psd) if reducer_factory else None
    if red:
        red.zero_grad()
    tot, _ = O.train_step(psd, args, my_inp, my_tg)
    if red:
        red.finish()
    return {k: (p.grad.clone() if p.grad is not None else None) for k, p in psd.items()}, float(tot), red


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        def factory(named):
            # skip parameters that never get a gradient; 64 KiB buckets -> many buckets, exercising the overlap path
            names_skip = ('sketch_video_cross_attn.out_proj', 'class_head')
            skip = [p for k, p in named.items() if any(s in k for s in names_skip)]
            return parallel.BucketedGradAllReduce(list(named.values()), bucket_bytes=64 << 10, skip=skip)
        grads, tot, red = _local_grads(rank, world, factory)
        assert len(red.buckets) > 3
        mean_loss = parallel.reduce_scalar_mean(torch.tensor(tot))
        np.savez(os.path.join(outdir, f'rank{rank}.npz'), loss=float(mean_loss),
                 **{k: g.numpy() for k, g in grads.items() if g is not None})
        # second step on the same buckets (zero_grad + re-fill must keep the views bound)
        grads2, _, _ = None, None, None
        red.zero_grad()
        assert all(float(b['flat'].abs().max()) == 0.0 for b in red.buckets)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_bucketed_allreduce_world2_matches_mean_of_shard_grads():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        got = [np.load(os.path.join(d, f'rank{r}.npz')) for r in range(world)]
    # single-process reference: per-shard gradients, averaged
    ref = [_local_grads(r, world)[0:2] for r in range(world)]
    mean_loss = sum(t for _, t in ref) / world
    for r in range(world):
        assert abs(float(got[r]['loss']) - mean_loss) < 1e-6
    for k in ref[0][0]:
        gs = [ref[r][0][k] for r in range(world)]
        if gs[0] is None:
            assert k not in got[0].files or float(np.abs(got[0][k]).max()) == 0.0
            continue
        avg = sum(gs) / world
        for r in range(world):
            np.testing.assert_allclose(got[r][k], avg.numpy(), rtol=1e-5, atol=1e-7, err_msg=k)
    # both ranks hold identical reduced gradients
    for k in got[0].files:
        np.testing.assert_array_equal(got[0][k], got[1][k])


def test_single_process_reducer_is_identity():
    grads_plain, tot_a, _ = _local_grads(0, 1)
    grads_red, tot_b, red = _local_grads(0, 1, lambda named: parallel.BucketedGradAllReduce(list(named.values()), bucket_bytes=32 << 10))
    assert tot_a == tot_b
    for k, g in grads_plain.items():
        if g is None:
            assert float(grads_red[k].abs().max()) == 0.0  # never touched: stays at the zeroed bucket value
        else:
            torch.testing.assert_close(grads_red[k], g, rtol=1e-6, atol=1e-8)


def test_arrival_order_puts_heads_first_and_input_projections_last():
    """buckets must follow the order in which backward produces gradients, or the first (largest) bucket's
    all-reduce only starts after backward has finished."""
    from svol_amd.modeling.svanet import build_svanet
    args = syn.head_args(hidden_dim=32, nheads=4, num_layers=3, num_queries=8, num_frames=4, input_vid_dim=32, input_skch_dim=32)
    model = build_svanet(args)
    names = {id(p): n for n, p in model.named_parameters()}
    order = [names[id(p)] for p in parallel.arrival_order(model)]
    assert sorted(order) == sorted(n for n, p in model.named_parameters() if p.requires_grad)
    k_heads = max(i for i, n in enumerate(order) if 'bbox_embed' in n or 'class_embed' in n)
    k_first_layer = min(i for i, n in enumerate(order) if 'transformer.layers.' in n)
    k_last_layer = max(i for i, n in enumerate(order) if 'transformer.layers.' in n)
    k_late = min(i for i, n in enumerate(order) if 'query_embed' in n or 'input_' in n)
    assert k_heads < k_first_layer and k_last_layer < k_late
    gate = [i for i, n in enumerate(order) if 'sketch_video_cross_attn' in n]
    body = [n for n in order if 'transformer.layers.' in n and 'sketch_video_cross_attn' not in n]
    layer_of = [int(n.split('transformer.layers.')[1].split('.')[0]) for n in body]
    assert layer_of == sorted(layer_of, reverse=True)            # last layer first
    l2 = [n for n in body if 'transformer.layers.2.' in n]
    assert 'norm6' in l2[0] and 'norm1' in l2[-1]                # inside a layer: last sub-block first
    # the gate-vector algebra of all layers runs ahead of the layer loop, so its gradients come after every layer's (last layer
    # first) and before the input projections'
    assert min(gate) > max(i for i, n in enumerate(order) if n in body) and max(gate) < k_late
    gl = [int(order[i].split('transformer.layers.')[1].split('.')[0]) for i in gate]
    assert gl == sorted(gl, reverse=True)
    # the reducer keeps that order when told so
    red = parallel.BucketedGradAllReduce(parallel.arrival_order(model), bucket_bytes=16 << 10,
                                         skip=parallel.unused_parameters(model), ordered=True)
    flat_names = [names[id(p)] for b in red.buckets for p in b['params']]
    assert flat_names == [n for n in order if 'class_head' not in n and 'sketch_video_cross_attn.out_proj' not in n]
    red.remove()


def test_flat_adamw_speaks_the_torch_adamw_schema_on_load_and_save():
    """CPU half of the optimizer-checkpoint interop (the update kernel itself is GPU-only: tests/test_gpu_parallel.py)."""
    torch.manual_seed(0)
    shapes = [(6, 5), (5,), (3,), (4, 4)]
    dead = 2
    pt = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    opt = torch.optim.AdamW(pt, lr=2e-3, betas=(0.8, 0.95), eps=1e-7, weight_decay=0.03)
    for step in range(3):
        opt.zero_grad()
        for i, p in enumerate(pt):
            if i != dead:
                p.grad = torch.randn(p.shape)
        opt.step()
    sd = opt.state_dict()
    pf = [torch.nn.Parameter(p.detach().clone()) for p in pt]
    red = parallel.BucketedGradAllReduce(pf, bucket_bytes=64, skip=[pf[dead]])
    assert len(red.buckets) > 1
    fo = parallel.FlatAdamW(red, params=pf, lr=1.0)
    assert isinstance(fo, torch.optim.Optimizer) and fo.state_dict()['state'] == {}
    for a, b in zip(pf, pt):                       # re-homed into the flat buffers, values unchanged
        assert torch.equal(a.detach(), b.detach())
    fo.load_state_dict(sd)
    assert fo.t == 3 and fo.lr == 2e-3 and fo.betas == (0.8, 0.95) and fo.eps == 1e-7 and fo.weight_decay == 0.03
    out = fo.state_dict()
    assert sorted(out['state']) == sorted(sd['state']) == [0, 1, 3]
    for i in out['state']:
        assert float(out['state'][i]['step']) == float(sd['state'][i]['step']) == 3.0
        assert torch.equal(out['state'][i]['exp_avg'], sd['state'][i]['exp_avg'])
        assert torch.equal(out['state'][i]['exp_avg_sq'], sd['state'][i]['exp_avg_sq'])
        assert out['state'][i]['exp_avg'].shape == pt[i].shape
    assert out['param_groups'][0]['params'] == [0, 1, 2, 3]
    fresh = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in pt], lr=1.0)
    fresh.load_state_dict(out)                     # torch accepts what FlatAdamW writes
    assert fresh.param_groups[0]['lr'] == 2e-3 and fresh.param_groups[0]['betas'] == (0.8, 0.95)
    # torch's schedulers wrap it and drive the learning rate the step kernel reads
    sched = torch.optim.lr_scheduler.StepLR(fo, step_size=1, gamma=0.1)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')            # "lr_scheduler.step() before optimizer.step()"
        sched.step()
    assert abs(fo.lr - 2e-4) < 1e-12
    # a mismatching file is refused
    bad = opt.state_dict()
    bad['param_groups'][0]['params'] = [0, 1, 2]
    with pytest.raises(ValueError):
        fo.load_state_dict(bad)
    bad = opt.state_dict()
    bad['state'][3]['exp_avg_sq'] = torch.zeros(3)
    with pytest.raises(ValueError):
        fo.load_state_dict(bad)
    red.remove()
