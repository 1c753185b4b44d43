"""Dev tool: time the StyleGAN2 discriminator steps of BASELINE configs 4 and 5 on one GPU."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd import config
from contrad_amd.augment import SimCLRAugment
from contrad_amd.engine import d_step_stylegan2, d_step_stylegan2_contrad, set_grad
from contrad_amd.models.gan import get_architecture
from contrad_amd.optim import FusedAdam
from contrad_amd.training.gan import setup


def run(arch, size, N, aug, fn, lbd_r1, d_reg_every, steps, lr, betas):
    dev = torch.device('cuda')
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture(arch, (size, size, 3))
    G, D = G.to(dev).train(), D.to(dev).train()
    P = argparse.Namespace(mode='contrad', aug='x', temp=0.1, lbd_a=1.0, distributed=False, lbd_r1=lbd_r1,
                           d_reg_every=d_reg_every)
    P = setup(P)
    P.augment_fn = aug
    opt = FusedAdam(D.parameters(), lr=lr, betas=betas)
    set_grad(G, False)
    x = torch.rand(N, 3, size, size, device=dev)
    times = []
    for step in range(1, steps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        d_loss, aux = fn(P, G, D, opt, {'loss': 'nonsat'}, x, step)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        print('  step %2d  %8.1f ms  loss %.4f gan %.4f%s' % (step, times[-1] * 1e3, d_loss.item(), aux['penalty'].item(),
              ('  r1 %.5f' % aux['r1'].item()) if 'r1' in aux else ''), flush=True)
    print('%s N=%d: median %.1f ms/step -> %.1f img/s; peak mem %.1f GB' % (arch, N, np.median(times[2:]) * 1e3,
          N / np.median(times[2:]), torch.cuda.max_memory_allocated() / 2**30), flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'both'
    if which in ('32', 'both'):
        run('stylegan2', 32, 64, SimCLRAugment(scale=(0.2, 1.0)), d_step_stylegan2, 0.1, 1, 8, 2e-3, (0.0, 0.99))
    if which in ('512', 'both'):
        aug = SimCLRAugment(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
                            sigma_range=(0.1, 2.0))
        run('stylegan2_512', 512, 16, aug, d_step_stylegan2_contrad, 0.5, 16, 18, 2.5e-3, (0.0, 0.99))
