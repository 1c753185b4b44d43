"""dev: element-wise gradient error vs the RAW reference goldens, per tensor (what FLIP_TOL has to absorb)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_sndcgan_gpu as T
from contrad_amd.optim import FusedAdam
from contrad_amd.training.gan import contrad as hip_contrad

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'sndcgan.npz'))
DEV = 'cuda'
_, D = T.build()
aug = torch.from_numpy(g['aug']).to(DEV); x = torch.from_numpy(g['x']).to(DEV); fake = torch.from_numpy(g['fake']).to(DEV)
P = T._P(lambda t: aug)
d_loss, a = hip_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, x, fake)
opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)); opt.zero_grad()
(d_loss + a['penalty']).backward()
grads = {k: p.grad for k, p in D.named_parameters()}
print('N =', int(g['N']))
for k in g.files:
    if k.startswith('grad/'):
        ok, l2 = T.grad_close(grads[k[5:]], g[k], 1.0)
        mx = (grads[k[5:]].cpu() - torch.from_numpy(g[k])).abs().max().item() / torch.from_numpy(g[k]).abs().max().item()
        print('%-40s L2 %.2e  max %.2e' % (k, l2, mx))
