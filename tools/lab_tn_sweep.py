import os, sys, torch
sys.path.insert(0, '/root/repo')
from svol_amd import ops
dev = 'cuda'
M = 50176
g = torch.Generator(device=dev).manual_seed(0)
def r(*s): return (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
x2, dpre, h, dy = r(M, 256), r(M, 2048), r(M, 2048), r(M, 256)
dq = r(M, 256)
def probs():
    return [(dpre, x2, torch.zeros(2048, 256, device=dev), torch.zeros(2048, device=dev)),
            (dy, h, torch.zeros(256, 2048, device=dev), torch.zeros(256, device=dev))]
pp = probs()
def run(): ops.gemm_tn_grouped(pp)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print('tile', os.environ.get('SVOL_TN_TILE', 'auto'), 'ffn pair us', e0.elapsed_time(e1) / 20 * 1e3)
pq = [(dq, x2, torch.zeros(256, 256, device=dev), None)] * 4
for _ in range(3): ops.gemm_tn_grouped(pq)
torch.cuda.synchronize(); e0.record()
for _ in range(20): ops.gemm_tn_grouped(pq)
e1.record(); torch.cuda.synchronize()
print('tile', os.environ.get('SVOL_TN_TILE', 'auto'), '4 x 256x256 us', e0.elapsed_time(e1) / 20 * 1e3)

dqk = r(M, 512)
p5 = probs() + [(dq, x2, torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)),
                (dqk, x2, torch.zeros(512, 256, device=dev), torch.zeros(512, device=dev)),
                (dq, x2, torch.zeros(256, 256, device=dev), None)]
for _ in range(3): ops.gemm_tn_grouped(p5)
torch.cuda.synchronize(); e0.record()
for _ in range(20): ops.gemm_tn_grouped(p5)
e1.record(); torch.cuda.synchronize()
print('splits', os.environ.get('SVOL_TN_SPLITS', 'auto'), 'video half group of 5 us', e0.elapsed_time(e1) / 20 * 1e3)
