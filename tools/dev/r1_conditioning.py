"""dev: how well-conditioned is d r1 / d bias in fp32?  The oracle (= the reference arithmetic) in fp32 vs float64 on the
512^2 R1-only fixture: weights agree to 1e-4, conv biases only to 0.7e-3 ... 1.5e-3 (tests/test_r1_gradient_gpu.py)."""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from oracle import stylegan2_oracle as S
from sg2_inputs import seeded_images
torch.set_num_threads(8)
size=512
shapes=S.d_param_shapes(512, False, 1.0)
sd=S.det_fill_d(shapes, seed=513, head_std=0.3)
NIMG=int(os.environ.get("R1_N","2")); aug=seeded_images(NIMG,512,9004)
res={}
for dt in (torch.float32, torch.float64):
    osd={k:v.clone().to(dt) for k,v in sd.items()}
    names=[k for k in osd if not k.endswith('kernel')]
    for k in names: osd[k].requires_grad_()
    r1=S.r1_penalty(lambda t: S.d_forward(osd,t,size)[0], aug.to(dt))
    gs=torch.autograd.grad(r1,[osd[k] for k in names],allow_unused=True)
    res[dt]={k:g for k,g in zip(names,gs)}
    print(dt, r1.item())
for k in names:
    a,b=res[torch.float32][k],res[torch.float64][k]
    if a is None or b.abs().max()==0: continue
    print('%-30s l2 %.2e  maxrel %.2e'%(k, ((a.double()-b).norm()/b.norm()).item(), ((a.double()-b).abs().max()/b.abs().max()).item()))
