"""Dev tool: stage-by-stage comparison of the D backward against PyTorch-CPU autograd."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import contrad_oracle as O
from contrad_amd import ops
from contrad_amd.models.gan import get_architecture
from contrad_amd.models.gan import sndcgan as S

DEV = 'cuda'
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'sndcgan.npz'))
aug = torch.from_numpy(g['aug'])
B = aug.shape[0]
sd = O.det_fill(O.sndcgan_d_param_shapes(), seed=1234)
for k in sd:
    if k.endswith('weight_orig') or k.endswith('bias'):
        sd[k].requires_grad_()
# oracle forward with retained intermediates
h = aug * 2 - 1
pre, post = [], []
for i, (ci, co, k, s, p) in enumerate(O.SNDCGAN_D_CONVS):
    pr = 'main.%d' % (2 * i)
    y = F.conv2d(h, O.spectral_norm_weight(sd, pr, True), sd[pr + '.bias'], stride=s, padding=p)
    y.retain_grad(); pre.append(y)
    h = F.leaky_relu(y, 0.1); post.append(h)
feats = h.reshape(B, -1)
out, proj, proj2 = O.d_heads(sd, feats, True, True)
N = B // 3
v = F.normalize(proj); r = F.normalize(proj2)
loss = O.nt_xent(v[:N], v[N:2*N], 0.1) + O.supcon_fake(r[:N], r[N:2*N], r[2*N:], 0.1) + O.gan_d_loss(out[:N], out[2*N:], 'nonsat')
loss.backward()

G, D = get_architecture('sndcgan', (32, 32, 3))
D.load_state_dict(O.det_fill(O.sndcgan_d_param_shapes(), seed=1234))
D = D.to(DEV).train()

# monkeypatch conv2d_dgrad / colstats to record
rec = []
orig_dgrad = ops.conv2d_dgrad
def dgrad(*a, **kw):
    o = orig_dgrad(*a, **kw); rec.append(('dgrad', o)); return o
ops.conv2d_dgrad = dgrad
from contrad_amd.training.gan import contrad as C
class P: pass
P.augment_fn = staticmethod(lambda t: aug.to(DEV)); P.temp = 0.1; P.lbd_a = 1.0; P.distributed = False
x = torch.from_numpy(g['x']).to(DEV); fake = torch.from_numpy(g['fake']).to(DEV)
dl, a = C.loss_D_fn(P, D, {'loss': 'nonsat'}, x, fake)
(dl + a['penalty']).backward()
print('loss', dl.item() + a['penalty'].item(), loss.item())
dg = [o for t, o in rec if o.dim() == 4 and o.shape[1] > 1]
# dg order: g for a6(pre6), then pre5 ... pre0
for j, o in enumerate(dg):
    i = 6 - j
    ref = pre[i].grad.permute(0, 2, 3, 1)
    print('g_pre[%d] shape %s rel %.3e   colsum rel %.3e' % (i, tuple(o.shape), rel(o, ref),
          rel(o.sum((0, 1, 2)), ref.double().sum((0, 1, 2)).float())))
    d = (o.cpu() - ref).abs()
    idx = d.flatten().argmax().item()
    print('     worst elem idx', np.unravel_index(idx, tuple(o.shape)), 'got', o.flatten()[idx].item(), 'ref', ref.flatten()[idx].item(),
          ' act', post[i].permute(0, 2, 3, 1).flatten()[idx].item())
for k, prm in D.named_parameters():
    if k.endswith('bias'):
        print(k, rel(prm.grad, sd[k].grad))
