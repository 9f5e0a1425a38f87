"""GPU parity of the whole hot path (head forward + matcher + criterion + backward) through the
product modules against the reference's golden vectors: 1e-3 fp32 / 1e-2 bf16 on outputs and
losses (north_star), Hungarian assignment bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = ['tiny_video', 'tiny_frame', 'cfg1_video', 'cfg1_frame', 'mid_video', 'mid32_video']


@pytest.mark.parametrize('name', CASES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_head_golden(name, dtype):
    from tests import gpu_checks as G
    res = G.check_head_case(name, dtype)
    bad = {k: v for k, v in res.items() if not (v[0] <= v[1])}
    assert not bad, '\n'.join(f'{k}: err={e:.3e} tol={t:.1e}' for k, (e, t) in bad.items())


@pytest.mark.parametrize('name', ['cfg1_video', 'cfg1_frame'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_head_golden_gradient_sinks(name, dtype):
    """Same parity bar with the gradients owned by BucketedGradAllReduce: weight / bias / LayerNorm gradient
    kernels accumulate straight into the flat buckets and report completion to the reducer."""
    from tests import gpu_checks as G
    res = G.check_head_case(name, dtype, sinks=True)
    bad = {k: v for k, v in res.items() if not (v[0] <= v[1])}
    assert not bad, '\n'.join(f'{k}: err={e:.3e} tol={t:.1e}' for k, (e, t) in bad.items())
