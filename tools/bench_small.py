"""Dev tool: D-step time vs per-rank batch on one GPU (what each rank of an N-GPU strong-scaling run executes)."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd.augment import SimCLRAugment
from contrad_amd.engine import d_step, set_grad
from contrad_amd.models.gan import get_architecture
from contrad_amd.optim import FusedAdam
from contrad_amd.training.gan import contrad
dev = torch.device('cuda')
for n in (512, 256, 128, 64):
    G, D = get_architecture('sndcgan', (32, 32, 3)); G, D = G.to(dev).train(), D.to(dev).train()
    P = argparse.Namespace(temp=0.1, lbd_a=1.0, distributed=False, augment_fn=SimCLRAugment(scale=(0.2, 1.0)),
                           train_fn={'D': contrad.loss_D_fn})
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)); set_grad(G, False)
    x = torch.rand(n, 3, 32, 32, device=dev)
    for _ in range(5): d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
    tc = time.perf_counter() - t0                      # CPU time to enqueue 20 steps
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print('n_local=%4d  %.2f ms/step (cpu enqueue %.2f ms/step)  -> %.0f img/s per GPU' % (n, t1 / 20 * 1e3, tc / 20 * 1e3, n * 20 / t1), flush=True)
