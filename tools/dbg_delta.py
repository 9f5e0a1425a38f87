import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svol_amd import ops, _lib
B, H, L, dh = 1, 2, 128, 32
d = H * dh
pm = 1.4426950408889634 / math.sqrt(dh)
g = torch.Generator().manual_seed(0)
q, k, v, do = [(torch.randn(B * L, d, generator=g)).to(torch.bfloat16).cuda() for _ in range(4)]
qp = (q.float() * pm).to(torch.bfloat16)
o, lse = ops.attn_fwd(qp, k, v, B, H, L, L, dh, None, pm)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
delta = torch.full((B, H, L), 7.0, dtype=torch.float32, device='cuda')
rc = _lib.lib().svol_attn_bwd(qp.data_ptr(), d, k.data_ptr(), d, v.data_ptr(), d, o.data_ptr(), d, do.data_ptr(), d, lse.data_ptr(),
                              delta.data_ptr(), 0, dq.data_ptr(), d, dk.data_ptr(), d, dv.data_ptr(), d, B, H, L, L, dh,
                              1.0 / math.sqrt(dh), pm, 0, 0, 1, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ref = (o.float() * do.float()).view(B, L, H, dh).sum(-1).permute(0, 2, 1)
print('rc', rc, 'delta err', float((delta - ref).abs().max()), 'ref max', float(ref.abs().max()))
print(delta[0, 0, :8].tolist()); print(ref[0, 0, :8].tolist())
prod = (o.float() * do.float()).view(B, L, H, dh)
lo = prod[..., [0,1,2,3,4,5,6,7,16,17,18,19,20,21,22,23]].sum(-1).permute(0, 2, 1)
hi = prod[..., [8,9,10,11,12,13,14,15,24,25,26,27,28,29,30,31]].sum(-1).permute(0, 2, 1)
print('lo', lo[0,0,:4].tolist()); print('hi', hi[0,0,:4].tolist()); print('2lo', (2*lo)[0,0,:4].tolist(), '2hi', (2*hi)[0,0,:4].tolist())
# maybe rows are mixed: compare against delta of other rows
print('match rows?', [int((ref[0,0] - delta[0,0,i]).abs().argmin()) for i in range(8)])
