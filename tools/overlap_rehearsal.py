"""Rehearsal of the data-parallel gradient exchange's HBM side on ONE GPU (SURVEY.md 8e claims the exchange
hides behind the backward).  No second GPU is available here, so what can be measured is the interference of the
exchange's MEMORY traffic with the step: while the captured D-step replays, a second stream streams the bytes an
all-reduce of the packed gradients moves through this GPU's HBM -- it reads the local gradient and writes the reduced one,
twice over for a ring (reduce-scatter + all-gather): 2 x (read + write) of the gradient size per step, spread over the
step in chunks with idle gaps between them (torch.cuda._sleep).  The copies are ordinary wide elementwise kernels (an
RCCL ring runs on a handful of CUs: this is the harsher case -- the copy's waves share every CU with the Winograd blocks).

usage: python tools/overlap_rehearsal.py [c10_b64 | sg2_512 | both]  ->  ms per step alone / with the side traffic"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd.augment import SimCLRAugment                                    # noqa: E402
from contrad_amd.engine import GraphedDStep, GraphedSG2DStep, d_step, d_step_stylegan2_contrad, set_grad   # noqa: E402
from contrad_amd.models.gan import get_architecture                             # noqa: E402
from contrad_amd.optim import FusedAdam                                         # noqa: E402
from contrad_amd.training.gan import contrad, setup                             # noqa: E402


def timed(step, side, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        step(i)
        if side is not None:
            side()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def make_side(nbytes, step_ms, chunks=8):
    """2 x (read + write) of nbytes per call on a side stream, in `chunks` pieces per pass with gaps that spread them over
    ~80 % of the step."""
    dev = torch.device('cuda')
    n = nbytes // 4
    src, dst = torch.empty(n, device=dev), torch.empty(n, device=dev)
    st = torch.cuda.Stream()
    piece = (n + chunks - 1) // chunks
    gap_cycles = int(0.8 * step_ms * 1e-3 * 2.1e9 / (2 * chunks))      # (torch.cuda._sleep counts shader clocks)

    def side():
        with torch.cuda.stream(st):
            for _pass in range(2):
                for c in range(chunks):
                    a, b = c * piece, min(n, (c + 1) * piece)
                    dst[a:b].copy_(src[a:b])
                    torch.cuda._sleep(gap_cycles)
    return side, st


def run_c10(n_local):
    dev = torch.device('cuda')
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture('sndcgan', (32, 32, 3))
    G, D = G.to(dev).train(), D.to(dev).train()
    P = argparse.Namespace(temp=0.1, lbd_a=1.0, distributed=False, augment_fn=SimCLRAugment(scale=(0.2, 1.0)).to(dev),
                           train_fn={'D': contrad.loss_D_fn})
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)); set_grad(G, False)
    x = torch.rand(n_local, 3, 32, 32, device=dev)
    for _ in range(3):
        d_step(P, G, D, opt, {'loss': 'nonsat'}, x)
    g = GraphedDStep(P, G, D, opt, {'loss': 'nonsat'}, x, warmup=1)
    grad_bytes = sum(p.numel() for p in D.parameters()) * 4
    return (lambda i: g()), grad_bytes


def run_sg2_512():
    dev = torch.device('cuda')
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture('stylegan2_512', (512, 512, 3))
    G, D = G.to(dev).train(), D.to(dev).train()
    P = argparse.Namespace(mode='contrad', aug='x', temp=0.1, lbd_a=1.0, distributed=False, lbd_r1=0.5, d_reg_every=16)
    P = setup(P)
    P.augment_fn = SimCLRAugment(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
                                 sigma_range=(0.1, 2.0)).to(dev)
    opt = FusedAdam(D.parameters(), lr=2.5e-3, betas=(0.0, 0.99)); set_grad(G, False)
    x = torch.rand(16, 3, 512, 512, device=dev)
    for s in (16, 1, 2):
        d_step_stylegan2_contrad(P, G, D, opt, {'loss': 'nonsat'}, x, s)
    g = GraphedSG2DStep(P, G, D, opt, {'loss': 'nonsat'}, x, contrad_script=True, warmup=1)
    grad_bytes = sum(p.numel() for p in D.parameters()) * 4
    return (lambda i: g(i % 15 + 1)), grad_bytes          # plain steps only (the lazy-R1 step is 1 in 16)


def report(name, step, grad_bytes, reps):
    for _ in range(3):
        step(0)
    alone = timed(step, None, reps)
    side, st = make_side(grad_bytes, alone)
    for _ in range(2):
        step(0); side()
    torch.cuda.synchronize()
    both = timed(step, side, reps)
    alone2 = timed(step, None, reps)
    # the side traffic by itself (its own duration bounds what "hidden" can mean)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        side()
    torch.cuda.synchronize()
    side_ms = (time.perf_counter() - t0) / reps * 1e3
    print('%-10s gradient %6.1f MB: step alone %7.3f ms (again %7.3f) | with 2 x (read + write) of it on a second stream %7.3f ms '
          '(%+.1f %%) | the side traffic alone, with its gaps, %6.3f ms per step' %
          (name, grad_bytes / 1e6, alone, alone2, both, (both / min(alone, alone2) - 1) * 100, side_ms), flush=True)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'both'
    if which in ('c10_b64', 'both'):
        step, gb = run_c10(64)
        report('c10 b64', step, gb, 200)
        step, gb = run_c10(512)
        report('c10 b512', step, gb, 60)
    if which in ('sg2_512', 'both'):
        step, gb = run_sg2_512()
        report('sg2_512', step, gb, 30)
