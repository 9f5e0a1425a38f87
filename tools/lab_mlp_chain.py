import os, sys, math, torch
sys.path.insert(0, '/root/repo')
from svol_amd import ops
dev = 'cuda'
M, F = 50176, 2048
g = torch.Generator(device=dev).manual_seed(0)
def r(*s, sc=1.0, dt=torch.bfloat16): return (torch.randn(*s, device=dev, generator=g) * sc).to(dt)
X, Wa, Wb = r(M, 256), r(F, 256, sc=1 / 16), r(256, F, sc=1 / 45)
ba, bb, r32 = r(F, dt=torch.float32), r(256, dt=torch.float32), r(M, 256, dt=torch.float32)
dY = r(M, 256)
Wbt, Wat = Wb.t().contiguous(), Wa.t().contiguous()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
hid, dpre, Y = ops.mlp_chain_fwd(X, Wa, ba, Wb, bb, r32)
print('chain fwd us', t(lambda: ops.mlp_chain_fwd(X, Wa, ba, Wb, bb, r32)))
print('chain bwd us', t(lambda: ops.mlp_chain_bwd(dY, Wbt, dpre, Wat)))
L = ops._lib.lib()
hid2, pre2, S3 = torch.empty_like(hid), torch.empty_like(hid), torch.empty_like(Y)
def two_fwd():
    L.svol_gemm_nt(ops._ptr(X), 256, None, 0, ops._ptr(Wa), 256, ops._ptr(hid2), F, ops._ptr(ba), None, ops.ACT_GELU_D if hasattr(ops, 'ACT_GELU_D') else 5, ops._ptr(pre2), F, None, 0, 0, M, F, 256, ops._dt(X), ops._stream())
    L.svol_gemm_nt(ops._ptr(hid2), F, None, 0, ops._ptr(Wb), F, ops._ptr(S3), 256, ops._ptr(bb), None, 0, None, 0, ops._ptr(r32), 256, 1, M, 256, F, ops._dt(X), ops._stream())
try:
    print('two-launch fwd us', t(two_fwd))
    print('fwd agree: hid', float((hid.float() - hid2.float()).abs().max()), 'dpre', float((dpre.float() - pre2.float()).abs().max()), 'Y', float((Y - S3).abs().max()))
except Exception as e:
    print('two-launch fwd failed', e)
def two_bwd():
    dT, cs = ops.gemm_nt_dact(dY, Wbt, dpre, 5)
    return dT, ops.gemm_nt(dT, Wat) if hasattr(ops, 'gemm_nt') else None
try:
    print('two-launch bwd us', t(two_bwd))
except Exception as e:
    print('two-launch bwd failed', e)
