"""Dev tool: host enqueue time vs GPU time of the StyleGAN2 D-steps (is the step host-bound?)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd.augment import SimCLRAugment
from contrad_amd.engine import d_step_stylegan2, d_step_stylegan2_contrad, set_grad
from contrad_amd.hostio import THROTTLE
from contrad_amd.models.gan import get_architecture
from contrad_amd.optim import FusedAdam
from contrad_amd.training.gan import setup

dev = torch.device('cuda')
for arch, size, N, fn, r1, every, kw in (('stylegan2', 32, 64, d_step_stylegan2, 0.1, 1, dict(scale=(0.2, 1.0))),
                                         ('stylegan2_512', 512, 16, d_step_stylegan2_contrad, 0.5, 16,
                                          dict(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2,
                                               p_blur=0.5, sigma_range=(0.1, 2.0)))):
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture(arch, (size, size, 3)); G, D = G.to(dev).train(), D.to(dev).train()
    P = setup(argparse.Namespace(mode='contrad', aug='x', temp=0.1, lbd_a=1.0, distributed=False, lbd_r1=r1, d_reg_every=every))
    P.augment_fn = SimCLRAugment(**kw)
    opt = FusedAdam(D.parameters(), lr=2e-3, betas=(0.0, 0.99)); set_grad(G, False)
    x = torch.rand(N, 3, size, size, device=dev)
    for s in range(1, 4): fn(P, G, D, opt, {'loss': 'nonsat'}, x, s)
    torch.cuda.synchronize()
    THROTTLE.events.clear()
    THROTTLE.begin = lambda: None
    THROTTLE.end = lambda: None
    import contrad_amd.hostio as H
    # host-only time: no throttle, sync only at the end
    K = 8
    t0 = time.perf_counter()
    for s in range(1, K + 1):
        fn(P, G, D, opt, {'loss': 'nonsat'}, x, s if every == 1 else 1)
    tc = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print('%s N=%d: wall %.2f ms/step, host enqueue %.2f ms/step' % (arch, N, t1 / K * 1e3, tc / K * 1e3), flush=True)
