"""Data-parallel gradient exchange on the real device: two ranks share the one MI355X of the test box (gloo backend,
the functional stand-in for RCCL, which needs one GPU per rank) and run the product model with
BucketedGradAllReduce; see tests/dp_gpu_worker.py for what is asserted."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29531', os.path.join(HERE, 'dp_gpu_worker.py')]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert p.stdout.count('reducer == mean of per-rank gradients') == 2, p.stdout[-2000:]


@pytest.mark.gpu
def test_flat_adamw_matches_torch_adamw():
    """svol_amd.parallel.FlatAdamW (one kernel per gradient bucket) against torch.optim.AdamW on the same gradients, 4 steps;
    a parameter the reducer skips stays untouched, a bucket boundary falls inside the list."""
    import torch
    from svol_amd import parallel
    torch.manual_seed(0)
    shapes = [(64, 33), (33,), (7,), (128, 128), (5, 3, 2), (1,)]
    pa = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    skip = [pa[2]]
    red = parallel.BucketedGradAllReduce(pa, bucket_bytes=40000, skip=skip)
    opt_a = parallel.FlatAdamW(red, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    opt_b = torch.optim.AdamW([p for i, p in enumerate(pb) if i != 2], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    for step in range(4):
        red.zero_grad()
        opt_b.zero_grad()
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 2:
                continue
            g = torch.randn(a.shape, device='cuda') * (1.0 + step)
            a.grad.copy_(g)
            b.grad = g.clone()
        opt_a.step()
        opt_b.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-6 * max(1.0, float(b.detach().abs().max())), i
    assert torch.equal(pa[2].detach(), pb[2].detach())
