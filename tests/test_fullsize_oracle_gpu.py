"""Whole D-steps at the BASELINE sizes against the CPU oracle, on the same host-sampled random numbers.

The per-layer tests cover the launch plans that only large batches select (padding-skipping tile plans from 1 024
images, split-K and tile choices that depend on the row count) against ``F.conv2d``; the goldens pin the step at N = 2 ... 8.
These tests close the gap in between: the COMPLETE step -- augmentation of 3N images, D forward, the three losses, backward
-- at the sizes BASELINE.json names, compared with the oracle's restatement of ``training/gan/contrad.py:35-70`` /
``train_stylegan2.py:106-113,199-212`` / ``train_stylegan2_contraD.py:95-164``: losses, d_real / d_gen, r1 and every
parameter-gradient norm at the north-star tolerance 1e-3.  The fakes are whatever the HIP generator produced (its own
parity is tested elsewhere; StyleGAN2's latents come from the device RNG) and are handed to the oracle as data.
"""
import argparse
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from contrad_amd.augment import SimCLRAugment
from contrad_amd.engine import loss_D_fn_separate, r1_loss, set_grad
from contrad_amd.models.gan import get_architecture
from contrad_amd.training.gan import contrad as hip_contrad
from oracle import contrad_oracle as O
from oracle import stylegan2_oracle as S
from sg2_inputs import seeded_images

pytestmark = pytest.mark.gpu
TOL = 1e-3
DEV = 'cuda'


def _threads():
    torch.set_num_threads(min(32, os.cpu_count() or 8))


def _relerr(got, ref):
    return abs(float(got) - float(ref)) / max(abs(float(ref)), 1e-30)


def _sndcgan_step_against_oracle(N, tag, margin):
    """One SNDCGAN + ContraD discriminator step at batch N (3N images through D, simclr augmentation) against the oracle:
    fakes, both losses, d_real / d_gen, all 26 gradient norms and the power-iteration vectors."""
    _threads()
    G, D = get_architecture('sndcgan', (32, 32, 3))
    dsd = O.det_fill(O.sndcgan_d_param_shapes(), seed=21)
    gsd_fill = O.det_fill(O.sndcgan_g_param_shapes(), seed=22)
    D.load_state_dict(dsd)
    gsd = dict(G.state_dict()); gsd.update(gsd_fill); G.load_state_dict(gsd)
    G, D = G.to(DEV).train(), D.to(DEV).train()
    set_grad(G, False)
    P = argparse.Namespace(temp=0.1, lbd_a=1.0, distributed=False, augment_fn=SimCLRAugment(scale=(0.2, 1.0)))
    x = torch.rand(N, 3, 32, 32, generator=torch.Generator().manual_seed(9))

    torch.manual_seed(31); np.random.seed(31)
    with torch.no_grad():
        fake = G(G.sample_latent(N))
    d_loss, aux = hip_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, x.to(DEV), fake)
    D.zero_grad()
    (d_loss + aux['penalty']).backward()
    torch.cuda.synchronize()

    osd = O.det_fill(O.sndcgan_d_param_shapes(), seed=21)
    for k in osd:
        if k.endswith('weight_orig') or k.endswith('bias'):
            osd[k].requires_grad_()
    torch.manual_seed(31); np.random.seed(31)
    with torch.no_grad():
        ofake = O.sndcgan_g_forward(gsd_fill, O.sample_latent_sndcgan(N))
    margin(tag + '/fakes (G forward, max-abs)', (fake.cpu() - ofake).abs().max().item(), TOL)
    p = O.sample_simclr_params(3 * N, 32, 32, O.SIMCLR_CIFAR)
    aug = O.simclr_apply(torch.cat([x, x, fake.cpu()]), p)
    closs, gloss, d_real, d_gen = O.contrad_loss_d(lambda t: O.sndcgan_d_forward(osd, t, sg_linear=True)[:3], aug, N)
    (closs + gloss).backward()

    margin(tag + '/contrad_loss', _relerr(d_loss.item(), closs.item()), TOL)
    margin(tag + '/gan_loss', _relerr(aux['penalty'].item(), gloss.item()), TOL)
    margin(tag + '/d_real', abs(aux['d_real'].item() - d_real.item()), TOL * max(1.0, abs(d_real.item())))
    margin(tag + '/d_gen', abs(aux['d_gen'].item() - d_gen.item()), TOL * max(1.0, abs(d_gen.item())))
    for k, prm in D.named_parameters():
        margin(tag + '/gradnorm/' + k, _relerr(prm.grad.norm().item(), osd[k].grad.norm().item()), TOL)
    # the power-iteration vectors after the step
    for k, b in D.named_buffers():
        if k.endswith('weight_u') or k.endswith('weight_v'):
            margin(tag + '/' + k, (b.cpu() - osd[k]).abs().max().item(), TOL)


def test_config2_sndcgan_step_at_batch_512_against_oracle(margin):
    """BASELINE configs[1]: SNDCGAN + ContraD, 32x32, N = 512 (1 536 images through D), simclr augmentation."""
    _sndcgan_step_against_oracle(512, 'config2', margin)


def test_config3_sndcgan_step_at_the_per_rank_batch_64_against_oracle(margin):
    """BASELINE configs[2] as ONE of its eight ranks sees it (train_gan.py:245-247: 512 // 8 = 64 reals, 192 images through
    D): the launch plans of 192-image GEMMs -- 64x64 tiles, split-K forward / data-gradient workspaces, no pixel-major
    tiles -- differ from those at 1 536 images, so the whole step is compared at this size too.  (The 8-rank exchange
    itself is covered by tests/test_dp_two_ranks_gpu.py at the same per-rank batch, against the oracle on the global
    batch.)"""
    _sndcgan_step_against_oracle(64, 'config3_rank', margin)


def test_config4_stylegan2_32_step_with_r1_at_batch_64_against_oracle(margin):
    """BASELINE configs[3]: StyleGAN2 (small32) + ContraD, N = 64, R1 every step with lbd_r1 = 0.1 (train_stylegan2.py
    semantics: ONE 3N-image discriminator call, r1 on a fresh augmentation of the reals)."""
    _threads()
    N, lbd_r1 = 64, 0.1
    G, D = get_architecture('stylegan2', (32, 32, 3))
    shapes = S.d_param_shapes(32, True)
    sd = S.det_fill_d(shapes, seed=2031)
    D.load_state_dict(sd)
    G, D = G.to(DEV).train(), D.to(DEV).train()
    set_grad(G, False)
    P = argparse.Namespace(temp=0.1, lbd_a=1.0, distributed=False, augment_fn=SimCLRAugment(scale=(0.2, 1.0)))
    x = torch.rand(N, 3, 32, 32, generator=torch.Generator().manual_seed(10))
    torch.manual_seed(4); np.random.seed(4); torch.cuda.manual_seed(4)
    with torch.no_grad():
        fake = G(G.sample_latent(N), style_mix=0.9)

    torch.manual_seed(41); np.random.seed(41)
    d_loss, aux = hip_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, x.to(DEV), fake)
    r1 = r1_loss(D, x.to(DEV), P.augment_fn)
    D.zero_grad()
    (d_loss + aux['penalty'] + (0.5 * lbd_r1) * r1 * 1).backward()
    torch.cuda.synchronize()

    osd = {k: v.clone() for k, v in sd.items()}
    for k in osd:
        if not k.endswith('kernel'):
            osd[k].requires_grad_()
    torch.manual_seed(41); np.random.seed(41)
    p = O.sample_simclr_params(3 * N, 32, 32, O.SIMCLR_CIFAR)
    aug = O.simclr_apply(torch.cat([x, x, fake.cpu()]), p)
    p1 = O.sample_simclr_params(N, 32, 32, O.SIMCLR_CIFAR)
    aug_r1 = O.simclr_apply(x, p1)
    closs, gloss, d_real, d_gen = O.contrad_loss_d(lambda t: S.d_forward(osd, t, 32, sg_linear=True)[:3], aug, N)
    or1 = S.r1_penalty(lambda t: S.d_forward(osd, t, 32)[0], aug_r1)
    (closs + gloss + (0.5 * lbd_r1) * or1).backward()

    margin('config4/contrad_loss', _relerr(d_loss.item(), closs.item()), TOL)
    margin('config4/gan_loss', _relerr(aux['penalty'].item(), gloss.item()), TOL)
    margin('config4/d_real', abs(aux['d_real'].item() - d_real.item()), TOL * max(1.0, abs(d_real.item())))
    margin('config4/d_gen', abs(aux['d_gen'].item() - d_gen.item()), TOL * max(1.0, abs(d_gen.item())))
    margin('config4/r1', _relerr(r1.item(), or1.item()), TOL)
    for k, prm in D.named_parameters():
        margin('config4/gradnorm/' + k, _relerr(prm.grad.norm().item(), osd[k].grad.norm().item()), TOL)


def _config5_batch():
    """16 (BASELINE configs[4]'s per-GPU batch) when the host has the memory and cores for the oracle at that size, else 4."""
    if os.environ.get('CONTRAD_CONFIG5_N'):
        return int(os.environ['CONTRAD_CONFIG5_N'])
    try:
        with open('/proc/meminfo') as f:
            avail_gb = int(next(l for l in f if l.startswith('MemAvailable')).split()[1]) / 2 ** 20
    except (OSError, StopIteration, ValueError):
        avail_gb = 0.0
    return 16 if avail_gb >= 128 and (os.cpu_count() or 1) >= 32 else 4


def test_config5_stylegan2_512_step_at_512_against_oracle(margin):
    """BASELINE configs[4] at the AFHQ resolution, simclr_hq, the call structure of train_stylegan2_contraD.py (fakes N and
    real views 2N through D separately, r1 on its own call, weight (0.5 * lbd_r1) * d_reg_every = 80 as in the lazy-R1
    step): the three losses, d_real / d_gen, r1 AND every parameter-gradient norm of the whole step, at BASELINE's per-GPU
    batch N = 16 where the host can hold the oracle (its 512^2 forward + R1 double backward + backward peak at 43 GB of
    host memory; two minutes on the GPU box's 256 cores; round 6: worst margin 0.34 of the tolerance,
    profiles/r06_config5_n16_margins.txt), at N = 4 (11 GB, 80 s on 8 cores) on smaller hosts; CONTRAD_CONFIG5_N overrides.
    The gradient ENTRIES are checked against the reference golden at N = 2 in tests/test_stylegan2_512_gpu.py."""
    _threads()
    N = _config5_batch()
    w_r1 = (0.5 * 10.0) * 16
    hq = dict(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
              sigma_range=(0.1, 2.0))
    G, D = get_architecture('stylegan2_512', (512, 512, 3))
    shapes = S.d_param_shapes(512, False, 1.0)
    sd = S.det_fill_d(shapes, seed=2032)
    D.load_state_dict(sd)
    D = D.to(DEV).train()
    P = argparse.Namespace(temp=0.1, lbd_a=1.0, distributed=False, augment_fn=SimCLRAugment(**hq))
    x = seeded_images(N, 512, 12)
    fake = seeded_images(N, 512, 13)          # (G(512)'s forward has its own golden; any image batch serves as fakes)

    torch.manual_seed(51); np.random.seed(51)
    d_loss, aux = loss_D_fn_separate(P, D, {'loss': 'nonsat'}, x.to(DEV), fake.to(DEV))
    r1 = r1_loss(D, x.to(DEV), P.augment_fn)
    D.zero_grad()
    torch.add(d_loss + aux['penalty'], r1, alpha=w_r1).backward()
    torch.cuda.synchronize()

    osd = {k: v.clone() for k, v in sd.items()}
    for k in osd:
        if not k.endswith('kernel'):
            osd[k].requires_grad_()
    torch.manual_seed(51); np.random.seed(51)
    aug_f = O.simclr_apply(fake, O.sample_simclr_params(N, 512, 512, O.SIMCLR_HQ_AFHQ))
    aug_r = O.simclr_apply(torch.cat([x, x]), O.sample_simclr_params(2 * N, 512, 512, O.SIMCLR_HQ_AFHQ))
    aug_r1 = O.simclr_apply(x, O.sample_simclr_params(N, 512, 512, O.SIMCLR_HQ_AFHQ))
    d_gen, pf, p2f, _ = S.d_forward(osd, aug_f, 512, sg_linear=True)
    d_rs, pr, p2r, _ = S.d_forward(osd, aug_r, 512, sg_linear=True)
    views_r, reals = F.normalize(pr), F.normalize(p2r)
    others, fakes = F.normalize(pf), F.normalize(p2f)
    simclr = O.nt_xent(views_r[:N], views_r[N:], 0.1)
    sup = O.supcon_fake(reals[:N], reals[N:], fakes, 0.1)
    gan = F.softplus(d_gen).mean() + F.softplus(-d_rs[:N]).mean()
    or1 = S.r1_penalty(lambda t: S.d_forward(osd, t, 512)[0], aug_r1)
    (simclr + sup + gan + w_r1 * or1).backward()

    margin('config5/contrad_loss', _relerr(d_loss.item(), (simclr + sup).item()), TOL)
    margin('config5/gan_loss', _relerr(aux['penalty'].item(), gan.item()), TOL)
    margin('config5/d_real', abs(aux['d_real'].item() - d_rs[:N].mean().item()), TOL * max(1.0, abs(d_rs[:N].mean().item())))
    margin('config5/d_gen', abs(aux['d_gen'].item() - d_gen.mean().item()), TOL * max(1.0, abs(d_gen.mean().item())))
    margin('config5/r1', _relerr(r1.item(), or1.item()), TOL)
    for k, prm in D.named_parameters():
        margin('config5/gradnorm/' + k, _relerr(prm.grad.norm().item(), osd[k].grad.norm().item()), TOL)
