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
