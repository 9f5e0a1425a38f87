import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svol_amd import ops
from tests import gpu_checks as G
B, L, D, H = 2, 50, 32, 4
dt = torch.float32
x, pos = G._rnd((B, L, D), dt, 40), G._rnd((B, L, D), dt, 41)
u = G._rnd((B, H, D), torch.float32, 42, 0.2)
g = 1 + 0.1 * G._rnd((D,), torch.float32, 43); b = 0.1 * G._rnd((D,), torch.float32, 44)
dy, dyp = G._rnd((B, L, D), dt, 45), G._rnd((B, L, D), dt, 46)
xd = x.cuda().requires_grad_(True); ud = u.cuda().requires_grad_(True)
y, ypos = ops.gate(xd, pos.cuda(), ud, g.cuda(), b.cuda(), H)
torch.autograd.backward([y, ypos], [dy.cuda(), dyp.cuda()])
x64, u64 = x.double().requires_grad_(True), u.double().requires_grad_(True)
s = torch.einsum('bld,bhd->bhl', x64 + pos.double(), u64); s.retain_grad()
p = torch.softmax(s, -1); a = p.mean(1); a.retain_grad()
from oracle import svol_oracle as O
yr = O.layer_norm(x64 * (1 + a[..., None]), g.double(), b.double())
(yr * dy.double() + (yr + pos.double()) * dyp.double()).sum().backward()
print('du got', ud.grad[0, 0, :6].cpu().numpy()); print('du ref', u64.grad[0, 0, :6].numpy())
print('ratio', (ud.grad.cpu().double() / u64.grad)[0, :, :4])
# c reference
c_ref = (p * a.grad[:, None, :] / H).sum(-1)
print('c_ref', c_ref.detach().numpy())
print('ds ref sum', s.grad.sum(-1).detach().numpy())
