"""Data-parallel equivalence of the StyleGAN2 + ContraD D-step (BASELINE config 5: "16 x 8", one process per GPU
replacing the reference's nn.DataParallel(G_D), train_stylegan2_contraD.py:117-164,218-226) with a REAL second rank:
two processes share cuda:0 over a gloo group and each runs ``engine.d_step_stylegan2_contrad`` on its half of a global
batch -- separate N / 2N discriminator calls, packed embedding all-gather + regrouping, local-slice backward, R1 on the
local reals, flat gradient all-reduce, Adam's 1/W.

The minibatch-stddev statistics are rank-local by construction, so the single-process reference is not
"one rank on the global batch" but the same per-rank discriminator calls made in ONE process, with the global loss
assembled by hand:

    sum_r grad_r / W  ==  grad( L_con(global) / W  +  mean_r L_gan,r  +  mean_r (0.5 lbd_r1 d_reg_every) R1_r )
"""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NL, WORLD = 4, 2
LBD_R1, EVERY = 0.1, 1


def _models(dev):
    from contrad_amd.augment import SimCLRAugment
    from contrad_amd.models.gan import get_architecture
    torch.manual_seed(0); np.random.seed(0)
    _G, D = get_architecture('stylegan2', (32, 32, 3))
    D = D.to(dev).train()
    return D, SimCLRAugment(scale=(0.2, 1.0))


def _global_inputs(aug):
    """Reals, fakes and the three augmentation parameter blocks of a step in GLOBAL row order: fakes (N), the two real
    views (2N: view 1 of all reals, view 2 of all reals), the R1 call (N)."""
    N = NL * WORLD
    g = torch.Generator().manual_seed(321)
    images = torch.rand(N, 3, 32, 32, generator=g)
    fakes = torch.rand(N, 3, 32, 32, generator=g)
    torch.manual_seed(11); np.random.seed(11)
    Pf, cf, _ = aug.sample(N, 32, 32)
    Pr, _, _ = aug.sample(2 * N, 32, 32)
    P1, _, _ = aug.sample(N, 32, 32)
    for blk in (Pf, Pr, P1):
        blk[:, 15] = float(cf)
    return images, fakes, (Pf, Pr, P1), cf


class _FixedG(object):
    """Stands in for the generator: the step's fakes are an input of this test."""

    def __init__(self, fakes):
        self.fakes = fakes

    def sample_latent(self, n):
        return None

    def __call__(self, z, style_mix=0.9):
        return self.fakes


def _inject(aug, blocks, cf):
    calls = {'k': 0}

    def sample(B, a, b):
        blk = blocks[calls['k'] % len(blocks)]
        calls['k'] += 1
        assert blk.shape[0] == B
        return blk, cf, None
    aug.sample = sample


def _namespace(setup, aug, distributed):
    P = setup(argparse.Namespace(mode='contrad', aug='simclr', temp=0.1, lbd_a=1.0, distributed=distributed,
                                 lbd_r1=LBD_R1, d_reg_every=EVERY))
    P.augment_fn = aug
    return P


def _worker(rank, world, port, path, overlap=False):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from contrad_amd.engine import GradAllReducer, d_step_stylegan2_contrad, setup_grad_exchange
        from contrad_amd.optim import FusedAdam
        from contrad_amd.training.gan import setup
        import contrad_amd.third_party.gather_layer as gl
        import contrad_amd.training.gan.contrad as cd

        def gather_rows(x):          # gloo has no all_gather_into_tensor for device tensors: list form, same result
            outs = [torch.empty_like(x) for _ in range(world)]
            dist.all_gather(outs, x.contiguous())
            return torch.stack(outs, 0)
        gl.all_gather_rows = gather_rows
        cd.all_gather_rows = gather_rows

        dev = torch.device('cuda', 0)
        D, aug = _models(dev)
        images, fakes, (Pf, Pr, P1), cf = _global_inputs(aug)
        N = NL * world
        sl = slice(rank * NL, (rank + 1) * NL)
        rows2 = torch.cat([torch.arange(N)[sl], N + torch.arange(N)[sl]])
        _inject(aug, [Pf[sl], Pr[rows2], P1[sl]], cf)
        P = _namespace(setup, aug, True)
        opt = FusedAdam(D.parameters(), lr=2e-3, betas=(0.0, 0.99))
        # overlap: the packed weight gradients are all-reduced at their production sites inside the backward (fused call)
        # or in PackWeightsFn.backward (the R1 call's any-order graph), the biases in one packed collective afterwards
        reducer = setup_grad_exchange(D) if overlap else GradAllReducer(D.parameters())
        d_loss, aux = d_step_stylegan2_contrad(P, _FixedG(fakes[sl].to(dev)), D, opt, {'loss': 'nonsat'},
                                               images[sl].to(dev), 1, reducer)
        torch.cuda.synchronize()
        torch.save({'d_loss': d_loss.detach().cpu(), 'gan': aux['penalty'].detach().cpu(), 'r1': aux['r1'].detach().cpu(),
                    'grads': [p.grad.detach().cpu().clone() for p in D.parameters()],
                    'params': [p.detach().cpu().clone() for p in D.parameters()]}, '%s.rank%d' % (path, rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_two_rank_stylegan2_contrad_step_equals_the_hand_assembled_global_step(tmp_path, overlap):
    import torch.multiprocessing as mp
    from contrad_amd.engine import r1_loss
    from contrad_amd.training.gan import setup
    from contrad_amd.training.gan.contrad import _ContraDContrastive, _GanDLoss
    path = str(tmp_path / 'dp2')
    mp.spawn(_worker, args=(WORLD, 29551 + int(overlap), path, overlap), nprocs=WORLD, join=True)
    res = [torch.load('%s.rank%d' % (path, r)) for r in range(WORLD)]

    dev = torch.device('cuda', 0)
    D, aug = _models(dev)
    images, fakes, (Pf, Pr, P1), cf = _global_inputs(aug)
    N = NL * WORLD
    P = _namespace(setup, aug, False)
    flags = dict(sg_linear=True, projection=True, projection2=True)
    per_rank, gan, r1 = [], [], []
    for r in range(WORLD):
        sl = slice(r * NL, (r + 1) * NL)
        rows2 = torch.cat([torch.arange(N)[sl], N + torch.arange(N)[sl]])
        aug_f = aug.apply(fakes[sl].to(dev), Pf[sl], cf)
        aug_r = aug.apply(torch.cat([images[sl], images[sl]]).to(dev), Pr[rows2], cf)
        (d_gen, aux_g), (d_real2, aux_r) = D.call_batches([aug_f, aug_r], **flags)
        per_rank.append((aux_r, aux_g))
        g, _dr, _dg = _GanDLoss.apply(torch.cat([d_real2, d_gen], dim=0), NL, 'nonsat')
        gan.append(g)
        _inject(aug, [P1[sl]], cf)
        r1.append(r1_loss(D, images[sl].to(dev), aug))

    def assemble(key):            # [view 1 of all ranks; view 2 of all ranks; fakes of all ranks]
        return torch.cat([a_r[key][:NL] for a_r, _ in per_rank] + [a_r[key][NL:] for a_r, _ in per_rank] +
                         [a_g[key] for _, a_g in per_rank], dim=0)
    simclr, sup = _ContraDContrastive.apply(assemble('projection'), assemble('projection2'), N, P.temp, False)
    con = simclr + P.lbd_a * sup
    total = con / WORLD + sum(gan) / WORLD + sum((0.5 * LBD_R1) * v * EVERY for v in r1) / WORLD
    total.backward()

    for r in range(WORLD):       # the contrastive loss is global on every rank; GAN loss and R1 are rank-local
        assert abs(res[r]['d_loss'].item() - con.item()) < 1e-4 * abs(con.item())
        assert abs(res[r]['gan'].item() - gan[r].item()) < 1e-4 * abs(gan[r].item())
        assert abs(res[r]['r1'].item() - r1[r].item()) < 1e-3 * abs(r1[r].item())
    names = [k for k, _ in D.named_parameters()]
    for i, p in enumerate(D.parameters()):
        assert torch.equal(res[0]['grads'][i], res[1]['grads'][i]), names[i]        # after the exchange: identical
        want = p.grad.detach().cpu()
        got = res[0]['grads'][i] / WORLD
        scale = want.abs().max().item()
        if scale < 1e-7:
            assert got.abs().max().item() < 1e-6, names[i]
        else:
            assert ((got - want).norm() / want.norm()).item() < 1e-3, names[i]
    for a, b in zip(res[0]['params'], res[1]['params']):                              # Adam(1/W) on identical gradients
        assert torch.equal(a, b)
