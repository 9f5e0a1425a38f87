import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import svol_amd.ops as ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
D=256; M=50176; N=2048
x = torch.randn(M, D, device='cuda').bfloat16(); W = (torch.randn(N, D, device='cuda')*0.05).bfloat16(); b = torch.zeros(N, device='cuda')
hid = torch.empty(M, N, dtype=torch.bfloat16, device='cuda'); pre = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
both = torch.empty(M, 2*N, dtype=torch.bfloat16, device='cuda')
one = torch.empty(1, N, dtype=torch.bfloat16, device='cuda').expand(M, N)
few = torch.empty(16, N, dtype=torch.bfloat16, device='cuda')
for name, fn in [
  ('gelu (no pre)', lambda: ops.gemm_nt(x, W, b, ops.ACT_GELU, out=hid)),
  ('gelu+pre separate', lambda: ops.gemm_nt(x, W, b, ops.ACT_GELU, want_pre=True, out=hid, pre_out=pre)),
  ('gelu+pre same rows [M,2N]', lambda: ops.gemm_nt(x, W, b, ops.ACT_GELU, want_pre=True, out=both[:, :N], pre_out=both[:, N:])),
  ('gelu+pre pre->stride0', lambda: ops.gemm_nt(x, W, b, ops.ACT_GELU, want_pre=True, out=hid, pre_out=one)),
  ('gelu+pre both->stride0', lambda: ops.gemm_nt(x, W, b, ops.ACT_GELU, want_pre=True, out=one, pre_out=one)),
  ('gelu out->stride0', lambda: ops.gemm_nt(x, W, b, ops.ACT_GELU, out=one)),
  ('none', lambda: ops.gemm_nt(x, W, b, out=hid)),
  ('none out->stride0', lambda: ops.gemm_nt(x, W, b, out=one)),
]:
    t = timeit(fn); print(f'{name:32s} {t*1e3:.1f} us', flush=True)
