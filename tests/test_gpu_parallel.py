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
