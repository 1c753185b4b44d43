"""BASELINE config 5 (StyleGAN2_512 + ContraD, 512x512) and full-size config 4 on the device: reference-generated
goldens at N = 2 (tests/golden/make_golden.py::gen_stylegan2_512 -- ResidualDiscriminatorP(512, channel_multiplier=1)
driven with train_stylegan2_contraD.py's call sequence, and Generator(512) with explicit noise), plus size-independent
properties at the BASELINE sizes (N = 16 at 512^2, N = 64 + R1 every step at 32^2): finite, bitwise deterministic,
adjoint identities of the 512^2 layer shapes."""
import argparse
import copy

import numpy as np
import pytest
import torch

from contrad_amd import autograd_ops as A
from contrad_amd import ops
from contrad_amd.engine import (d_step_stylegan2, d_step_stylegan2_contrad, loss_D_fn_separate, r1_loss, set_grad)
from contrad_amd.models.gan import get_architecture
from contrad_amd.optim import FusedAdam
from oracle import stylegan2_oracle as S
from sg2_inputs import seeded_images

pytestmark = pytest.mark.gpu
TOL = 1e-3
FLIP_TOL = float(__import__('os').environ.get('CONTRAD_FLIP_TOL', '1e-3'))      # see tests/test_sndcgan_gpu.py
VERBOSE = bool(__import__('os').environ.get('CONTRAD_TEST_VERBOSE'))
DEV = 'cuda'


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def l2(a, b):
    a, b = torch.as_tensor(a).double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


class _P(object):
    temp, lbd_a, distributed = 0.1, 1.0, False


def _config5_step(golden, record=False):
    """train_stylegan2_contraD.py's D-step (separate N / 2N calls, lazy-R1 step) with the augmentation outputs of the
    fixture injected."""
    g = golden('stylegan2_512_d')
    N = int(g['N'])
    G, D = get_architecture('stylegan2_512', (512, 512, 3))
    shapes = S.d_param_shapes(512, False, 1.0)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == shapes and list(D.state_dict()) == list(shapes)
    sd = S.det_fill_d(shapes, seed=int(g['wseed']))
    D.load_state_dict(sd)
    D = D.to(DEV).train()
    aug_f = seeded_images(N, 512, int(g['seed_f']))
    aug_r = seeded_images(2 * N, 512, int(g['seed_r']))
    aug_r1 = seeded_images(N, 512, int(g['seed_r1']))
    for t, key in ((aug_f, 'sum_f'), (aug_r, 'sum_r'), (aug_r1, 'sum_r1')):      # same inputs as the generator run
        assert abs(t.double().sum().item() - float(g[key])) < 1e-6 * float(g[key])
    cpu_inputs = (aug_f, aug_r, aug_r1)
    aug_f, aug_r, aug_r1 = aug_f.to(DEV), aug_r.to(DEV), aug_r1.to(DEV)

    # forward values of the two separate calls
    with torch.no_grad():
        d_gen, af = D(aug_f, sg_linear=True, projection=True, projection2=True)
        d_rs, ar = D(aug_r, sg_linear=True, projection=True, projection2=True)
    assert rel(d_gen, g['d_gen']) < TOL and rel(d_rs, g['d_rs']) < TOL
    assert rel(af['projection'], g['proj_f']) < TOL and rel(af['projection2'], g['proj2_f']) < TOL
    assert rel(ar['projection'], g['proj_r']) < TOL and rel(ar['projection2'], g['proj2_r']) < TOL

    # the step: train_stylegan2_contraD.py semantics with the augmentation outputs injected
    P = _P()
    calls = []

    def augment_fn(t):
        calls.append(t.shape[0])
        if t.shape[0] == 2 * N:
            return aug_r
        return aug_f if len([c for c in calls if c == N]) == 1 else aug_r1
    P.augment_fn = augment_fn
    P.lbd_r1, P.d_reg_every = 0.5, 16
    if record:
        D._record_activations = True
    x = torch.rand(N, 3, 512, 512, device=DEV)
    d_loss, aux = loss_D_fn_separate(P, D, {'loss': 'nonsat'}, x, torch.rand(N, 3, 512, 512, device=DEV))
    r1 = r1_loss(D, x, P.augment_fn)
    assert calls == [N, 2 * N, N]
    loss = d_loss + aux['penalty'] + (0.5 * P.lbd_r1) * r1 * P.d_reg_every
    D.zero_grad()
    loss.backward()
    return g, N, D, sd, cpu_inputs, d_loss, aux, r1


def test_discriminator_512_contrad_step_against_reference(golden):
    """Config 5 against the RAW reference golden at the north_star tolerance: losses, r1 and every gradient norm at
    1e-3; the stored gradient entries (first 256 of each tensor) at 1e-3 relative L2 (leaky-relu slope flips of single
    units vs the reference's BLAS are the one thing the raw comparison cannot exclude -- the element-wise check without
    them is test_discriminator_512_step_on_the_same_linear_region below)."""
    g, N, D, _sd, _inp, d_loss, aux, r1 = _config5_step(golden)
    want = float(g['simclr']) + float(g['sup'])
    assert abs(d_loss.item() - want) < TOL * abs(want)
    assert abs(aux['penalty'].item() - float(g['gan'])) < TOL * float(g['gan'])
    assert abs(r1.item() - float(g['r1'])) < TOL * float(g['r1'])
    grads = {k: p.grad for k, p in D.named_parameters()}
    report, bad = [], []
    for k in g.files:
        if k.startswith('gradnorm/'):
            name = k[len('gradnorm/'):]
            ref = float(g[k])
            e = abs(grads[name].norm().item() - ref) / max(ref, 1e-30)
            report.append((e, 'norm', name))
            bad += [(name, e)] if not e < TOL else []
        elif k.startswith('gradhead/'):
            name = k[len('gradhead/'):]
            ref = torch.from_numpy(g[k])
            got = grads[name].reshape(-1)[:ref.numel()].cpu()
            e = l2(got, ref)
            report.append((e, 'head', name))
            bad += [(name, e)] if not e < FLIP_TOL else []
    if VERBOSE:
        for r in sorted(report)[-10:]:
            print('config5 raw golden: %.2e %s %s' % r)
    assert not bad, bad


def test_discriminator_512_step_on_the_same_linear_region(golden):
    """Config 5, strict element-wise check of EVERY gradient entry (first and second order): the oracle evaluated on
    the leaky-relu linear regions recorded from the HIP forwards (the merged N + 2N pass and the R1 batch); max-abs
    error relative to the tensor's max, 1e-3.  (N = 2 at 512^2: ~1 min of oracle time on the host cores.)"""
    import os
    import torch.nn.functional as F
    from oracle import contrad_oracle as O
    g, N, D, sd, (aug_f, aug_r, aug_r1), d_loss, aux, r1 = _config5_step(golden, record=True)
    (rec_a, hl_a, hpq_a), (rec_b, hl_b, hpq_b) = D._recorded[-2], D._recorded[-1]

    def masks_of(rec, sl):
        return [(t[sl] > 0).permute(0, 3, 1, 2).cpu() for t in rec]

    def head_masks(hl, hpq, sl):
        hl, hpq = hl.reshape(hl.shape[0], -1)[sl].cpu(), hpq.reshape(hpq.shape[0], -1)[sl].cpu()
        return (hl > 0, hpq[:, :512] > 0, hpq[:, 512:] > 0)

    torch.set_num_threads(min(32, os.cpu_count() or 8))
    osd = {k: v.clone() for k, v in sd.items()}
    for k in osd:
        if not k.endswith('kernel'):
            osd[k].requires_grad_()
    sf, sr, sall = slice(0, N), slice(N, 3 * N), slice(None)          # call_batches([fakes, real views])
    d_gen, pf, p2f, _ = S.d_forward(osd, aug_f, 512, sg_linear=True, masks=masks_of(rec_a, sf),
                                    head_masks=head_masks(hl_a, hpq_a, sf))
    d_rs, pr, p2r, _ = S.d_forward(osd, aug_r, 512, sg_linear=True, masks=masks_of(rec_a, sr),
                                   head_masks=head_masks(hl_a, hpq_a, sr))
    views_r, reals = F.normalize(pr), F.normalize(p2r)
    others, fakes = F.normalize(pf), F.normalize(p2f)
    simclr = O.nt_xent(views_r[:N], views_r[N:], 0.1)
    sup = O.supcon_fake(reals[:N], reals[N:], fakes, 0.1)
    gan = F.softplus(d_gen).mean() + F.softplus(-d_rs[:N]).mean()
    or1 = S.r1_penalty(lambda t: S.d_forward(osd, t, 512, masks=masks_of(rec_b, sall),
                                             head_masks=head_masks(hl_b, hpq_b, sall))[0], aug_r1)
    (simclr + sup + gan + (0.5 * 0.5) * or1 * 16).backward()
    assert abs(d_loss.item() - (simclr + sup).item()) < TOL * abs((simclr + sup).item())
    assert abs(aux['penalty'].item() - gan.item()) < TOL * gan.item()
    assert abs(r1.item() - or1.item()) < TOL * or1.item()
    worst, bad = [], []
    for k, prm in D.named_parameters():
        ref = osd[k].grad
        e = rel(prm.grad, ref)
        worst.append((e, k))
        bad += [(k, e)] if not e < TOL else []
    if VERBOSE:
        for e, k in sorted(worst)[-8:]:
            print('config5 same-region: %-28s %.2e' % (k, e))
    assert not bad, bad


def test_generator_512_forward_against_reference(golden):
    g = golden('stylegan2_512_g')
    G, _ = get_architecture('stylegan2_512', (512, 512, 3))
    shapes = S.g_param_shapes(512, False, 1.0)
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == shapes
    G.load_state_dict(S.fill_kernels(S.det_fill_g(shapes, seed=int(g['wseed'])), shapes))
    G = G.to(DEV).train()
    z = torch.from_numpy(g['z']).to(DEV)
    B = z.shape[0]
    nseed = int(g['nseed'])
    noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2),
                         generator=torch.Generator().manual_seed(nseed + i)).to(DEV) for i in range(G.num_layers)]
    with torch.no_grad():
        img0 = G(z, style_mix=0.0, noise=noise)
        img1 = G(z, style_mix=0.9, noise=noise, _mix=(torch.from_numpy(g['z_mix']).to(DEV),
                                                      torch.from_numpy(g['mix_layer'])))
    assert img0.shape == (B, 3, 512, 512)
    assert rel(img0[:, :, ::32, ::32], g['img0_sub']) < TOL and rel(img1[:, :, ::32, ::32], g['img1_sub']) < TOL
    assert rel(img0[:, :, 200:232, 300:332], g['img0_patch']) < TOL
    assert rel(img0.double().sum(3), g['img0_rowsum']) < TOL and rel(img1.double().sum(3), g['img1_rowsum']) < TOL


def _setup(arch, size, N, aug_kwargs, lbd_r1, every, lr):
    from contrad_amd.augment import SimCLRAugment
    from contrad_amd.training.gan import setup
    torch.manual_seed(0); np.random.seed(0)
    G, D = get_architecture(arch, (size, size, 3))
    G, D = G.to(DEV).train(), D.to(DEV).train()
    P = setup(argparse.Namespace(mode='contrad', aug='x', temp=0.1, lbd_a=1.0, distributed=False, lbd_r1=lbd_r1,
                                 d_reg_every=every))
    P.augment_fn = SimCLRAugment(**aug_kwargs)
    set_grad(G, False)
    x = seeded_images(N, size, 5).to(DEV)
    return G, D, P, x, lr


def _two_identical_runs(fn, G, D, P, x, lr, step):
    outs = []
    for _ in range(2):
        Dc = copy.deepcopy(D)
        opt = FusedAdam(Dc.parameters(), lr=lr, betas=(0.0, 0.99))
        torch.manual_seed(123); np.random.seed(123)
        torch.cuda.manual_seed(123)
        d_loss, aux = fn(P, G, Dc, opt, {'loss': 'nonsat'}, x, step)
        torch.cuda.synchronize()
        outs.append((d_loss.detach().clone(), aux, [p.grad.clone() for p in Dc.parameters()],
                     [p.detach().clone() for p in Dc.parameters()]))
    return outs


def test_config5_full_size_step_is_finite_and_deterministic():
    """StyleGAN2_512, N = 16, simclr_hq, one plain step and one lazy-R1 step (step 16)."""
    hq = dict(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
              sigma_range=(0.1, 2.0))
    G, D, P, x, lr = _setup('stylegan2_512', 512, 16, hq, 0.5, 16, 2.5e-3)
    for step in (1, 16):
        a, b = _two_identical_runs(d_step_stylegan2_contrad, G, D, P, x, lr, step)
        assert torch.isfinite(a[0]) and torch.isfinite(a[1]['penalty'])
        assert ('r1' in a[1]) == (step == 16)
        if step == 16:
            assert torch.isfinite(a[1]['r1']) and a[1]['r1'].item() > 0
        assert torch.equal(a[0], b[0])
        for ga, gb in zip(a[2], b[2]):
            assert torch.isfinite(ga).all() and torch.equal(ga, gb)
        for pa, pb, p0 in zip(a[3], b[3], D.parameters()):
            assert torch.equal(pa, pb)
        assert any(not torch.equal(pa, p0) for pa, p0 in zip(a[3], D.parameters()))
    assert torch.cuda.max_memory_allocated() < 80 * 2 ** 30


def test_config4_full_size_step_is_finite_and_deterministic():
    """StyleGAN2 small32, N = 64, simclr, R1 every step (--no_lazy)."""
    G, D, P, x, lr = _setup('stylegan2', 32, 64, dict(scale=(0.2, 1.0)), 0.1, 1, 2e-3)
    a, b = _two_identical_runs(d_step_stylegan2, G, D, P, x, lr, 1)
    assert torch.isfinite(a[0]) and torch.isfinite(a[1]['penalty']) and torch.isfinite(a[1]['r1'])
    assert torch.equal(a[0], b[0])
    for ga, gb in zip(a[2], b[2]):
        assert torch.isfinite(ga).all() and torch.equal(ga, gb)
    for pa, pb in zip(a[3], b[3]):
        assert torch.equal(pa, pb)


@pytest.mark.parametrize('shape', [
    # (N, H, W, Cin, Cout, k, stride, pad): layer shapes of ResidualDiscriminatorP(512, channel_multiplier=1)
    (4, 512, 512, 32, 32, 3, 1, 1),       # conv1 at 512^2 (128x32 tile)
    (4, 515, 515, 32, 64, 3, 2, 0),       # conv2 on the blurred (pad 2,2) map: odd size, stride 2
    (4, 256, 256, 32, 64, 1, 1, 0),       # skip 1x1 on the blurred + decimated map
    (4, 259, 259, 64, 128, 3, 2, 0),
    (8, 67, 67, 256, 512, 3, 2, 0),
    (16, 4, 4, 516, 512, 3, 1, 1),        # 513 channels padded to 516: the general float4 kernel
    (16, 4, 4, 528, 512, 3, 1, 1),        # last_conv as the model lays it out (513 -> 528: the lean loop)
])
def test_adjoint_identities_of_the_512_layer_shapes(shape):
    """<conv(x), g> == <x, dgrad(g)> == <W, wgrad(x, g)> on the kernels the 512^2 model selects."""
    N, H, W, C, K, k, s, p = shape
    gen = torch.Generator(device=DEV).manual_seed(C * 7 + H)
    x = torch.randn(N, H, W, C, device=DEV, generator=gen)
    w = torch.randn(K, C, k, k, device=DEV, generator=gen) / np.sqrt(C * k * k)
    wp = ops.pack_weight(w.cpu()).to(DEV)
    y = ops.conv2d_fwd(x, wp, None, K, k, k, s, p)
    gy = torch.randn(y.shape, device=DEV, generator=gen)
    dx = ops.conv2d_dgrad(gy, wp, tuple(x.shape), k, k, s, p)
    dw = ops.conv2d_wgrad(x, gy, k, k, s, p, ldw=wp.shape[1])
    a = (y.double() * gy.double()).sum().item()
    b = (x.double() * dx.double()).sum().item()
    c = (wp.double() * dw.double()).sum().item()
    # |a| ~ sqrt(#terms) * rms(y) * rms(gy); fp32 accumulation noise is ~1e-6 of that, a wrong tap / stride is O(1) of it
    tol = 1e-4 * abs(a) + 2e-5 * np.sqrt(y.numel()) * y.double().pow(2).mean().sqrt().item()
    assert abs(a - b) < tol and abs(a - c) < tol, (a, b, c, tol)


def test_config5_generator_step_at_512():
    """train_stylegan2_contraD.py:138-146,207 at the AFHQ size: G(512) with grad -> simclr_hq (blur backward) -> D with
    sg_linear=False -> softplus(-d).mean(); every generator parameter receives a finite gradient, D none."""
    from contrad_amd.train_stylegan2 import loss_G_nonsat, sample_generator
    hq = dict(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
              sigma_range=(0.1, 2.0))
    G, D, P, x, lr = _setup('stylegan2_512', 512, 4, hq, 0.5, 16, 2.5e-3)
    set_grad(G, True); set_grad(D, False)
    torch.manual_seed(3); np.random.seed(3); torch.cuda.manual_seed(3)
    gen = sample_generator(G, 4, style_mix=0.9, enable_grad=True)
    assert gen.shape == (4, 3, 512, 512) and gen.requires_grad
    d_gen, _ = D(P.augment_fn(gen), sg_linear=False, projection=True, projection2=True)
    g_loss = loss_G_nonsat(d_gen)
    g_loss.backward()
    assert torch.isfinite(g_loss)
    for k, p in G.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    assert all(p.grad is None for p in D.parameters())
    nz = sum(1 for p in G.parameters() if p.grad.abs().max().item() > 0)
    assert nz >= len(list(G.parameters())) - 2
