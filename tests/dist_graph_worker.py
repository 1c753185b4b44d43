"""Worker of tests/test_graph_gpu.py::test_graph_with_collectives_on_one_rank_rccl (own process: it owns a process
group and sets the 1-rank test hook).  D- and G-step of the SNDCGAN loop and the StyleGAN2 D-step, each (a) eager and
(b) replayed from a hipGraph that was captured WITH its RCCL collectives (SyncBN statistics, packed embedding
all-gather, overlapped / flat gradient all-reduce, Adam's 1/W in the device scalars); prints OK when (a) and (b) agree
bitwise in weights, optimizer steps and losses."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = 'cuda'


def sndcgan_setup(N, overlap):
    from contrad_amd import config
    from contrad_amd.augment import get_augment
    from contrad_amd.engine import OverlappedGradReducer, set_grad
    from contrad_amd.models.gan import get_architecture
    from contrad_amd.optim import FusedAdam
    from contrad_amd.training.gan import setup
    config.clear_config()
    config.parse_config_files_and_bindings([os.path.join(config.CONFIG_ROOT, 'defaults', 'gan.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'defaults', 'augment.gin'),
                                            os.path.join(config.CONFIG_ROOT, 'gan', 'cifar10', 'c10_b64.gin')])
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture('sndcgan', (32, 32, 3))
    G, D = G.to(DEV).train(), D.to(DEV).train()
    P = setup(argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=True))
    P.augment_fn = get_augment(mode='simclr').to(DEV)
    opt_D = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt_G = FusedAdam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    if overlap:
        D.enable_grad_overlap(OverlappedGradReducer())
    set_grad(G, False); set_grad(D, True)
    x = torch.rand(N, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
    return P, G, D, opt_D, opt_G, x


def sndcgan_d(overlap):
    from contrad_amd.engine import GradAllReducer, GraphedDStep, d_step
    N, K, W = 32, 4, 2
    P, G, D, opt, _og, x = sndcgan_setup(N, overlap)
    red = None if overlap else GradAllReducer(D.parameters())
    torch.manual_seed(7); np.random.seed(7)
    le = []
    for _ in range(W + K):
        dl, aux = d_step(P, G, D, opt, {'loss': 'nonsat'}, x, red)
        le.append((dl.item(), aux['penalty'].item()))
    want = [p.detach().clone() for p in D.parameters()]
    P, G, D, opt, _og, x = sndcgan_setup(N, overlap)
    torch.manual_seed(7); np.random.seed(7)
    g = GraphedDStep(P, G, D, opt, {'loss': 'nonsat'}, x, warmup=W)
    assert g.dist and (g.reducer is None) == overlap
    lg = []
    for _ in range(K):
        dl, aux = g()
        lg.append((dl.item(), aux['penalty'].item()))
    assert lg == le[W:], (lg, le[W:])
    for a, p in zip(want, D.parameters()):
        assert torch.equal(a, p.detach())


def sndcgan_g():
    from contrad_amd.engine import GradAllReducer, GraphedGStep, set_grad
    N, K, W = 32, 3, 2
    outs = []
    for graph in (False, True):
        P, G, D, _od, opt, x = sndcgan_setup(N, False)
        set_grad(G, True); set_grad(D, False)
        red = GradAllReducer(G.parameters())
        torch.manual_seed(9); np.random.seed(9)
        losses = []

        def eager():
            gen = G(G.sample_latent(N))
            gl = P.train_fn['G'](P, D, {'loss': 'nonsat'}, None, gen)
            opt.zero_grad()
            gl.backward()
            red()
            opt.step()
            return gl.detach()
        for _ in range(W):
            losses.append(eager().item())
        gs = GraphedGStep(P, G, D, opt, {'loss': 'nonsat'}, N, 32, 32) if graph else None
        for _ in range(K):
            losses.append((gs() if graph else eager()).item())
        outs.append((losses, [p.detach().clone() for p in G.parameters()], G.norm_init.running_mean.clone()))
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(outs[0][2], outs[1][2])


def stylegan2_d(overlap=False):
    from contrad_amd.augment import SimCLRAugment
    from contrad_amd.engine import (GradAllReducer, GraphedSG2DStep, d_step_stylegan2, set_grad, setup_grad_exchange)
    from contrad_amd.models.gan import get_architecture
    from contrad_amd.optim import FusedAdam
    from contrad_amd.training.gan import setup
    N, K, W = 8, 3, 2

    def build():
        torch.manual_seed(0); np.random.seed(0)
        G, D = get_architecture('stylegan2', (32, 32, 3))
        G, D = G.to(DEV).train(), D.to(DEV).train()
        P = setup(argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=True, lbd_r1=0.1,
                                     d_reg_every=1))
        P.augment_fn = SimCLRAugment(scale=(0.2, 1.0))
        opt = FusedAdam(D.parameters(), lr=2e-3, betas=(0.0, 0.99))
        set_grad(G, False)
        x = torch.rand(N, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
        return P, G, D, opt, x

    def seed():
        torch.manual_seed(7); np.random.seed(7); torch.cuda.manual_seed(7)
    P, G, D, opt, x = build()
    # overlap: packed weight gradients all-reduced at their production sites inside the backward, biases afterwards
    red = setup_grad_exchange(D) if overlap else GradAllReducer(D.parameters())
    assert (D._pack_comm is not None) == overlap
    seed()
    le = []
    for s in range(1, W + K + 1):
        dl, aux = d_step_stylegan2(P, G, D, opt, {'loss': 'nonsat'}, x, s, red)
        le.append((dl.item(), aux['penalty'].item(), aux['r1'].item()))
    want = [p.detach().clone() for p in D.parameters()]
    P, G, D, opt, x = build()
    if overlap:
        setup_grad_exchange(D)
    seed()
    g = GraphedSG2DStep(P, G, D, opt, {'loss': 'nonsat'}, x, contrad_script=False, warmup=W)
    assert g.dist and g.reducer is not None
    assert len(g.reducer.params) == (len(D.overlap_rest()) if overlap else len(list(D.parameters())))
    lg = []
    for s in range(1, K + 1):
        dl, aux = g(s)
        lg.append((dl.item(), aux['penalty'].item(), aux['r1'].item()))
    assert lg == le[W:], (lg, le[W:])
    for a, p in zip(want, D.parameters()):
        assert torch.equal(a, p.detach())


if __name__ == '__main__':
    import contrad_amd.engine as eng
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % sys.argv[1], rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    eng.FORCE_DIST = True
    try:
        for name, fn in (('sndcgan D-step, overlapped exchange', lambda: sndcgan_d(True)),
                         ('sndcgan D-step, flat exchange', lambda: sndcgan_d(False)),
                         ('sndcgan G-step', sndcgan_g), ('stylegan2 D-step + R1', stylegan2_d),
                         ('stylegan2 D-step + R1, weight gradients exchanged inside the backward', lambda: stylegan2_d(True))):
            fn()
            print('OK', name, flush=True)
    finally:
        dist.destroy_process_group()
    print('ALL OK', flush=True)
