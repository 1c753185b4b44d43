"""StyleGAN2 generator step on the HIP path (scope rows N1 / N2): differentiable mapping network, modulated convs
(input-modulate / output-demodulate form), noise injection, ToRGB + upsampled skip, the gradient through the
augmentation (incl. simclr_hq's blur at the AFHQ size) and through D's backward-to-input; EMA ``accumulate``; the
two training-script loops."""
import argparse
import os

import numpy as np
import pytest
import torch

from contrad_amd import autograd_ops as A
from contrad_amd import ops
from contrad_amd.augment import SimCLRAugment
from contrad_amd.engine import set_grad
from contrad_amd.models.gan import get_architecture
from contrad_amd.training.gan import contrad as hip_contrad
from oracle import contrad_oracle as O
from oracle import stylegan2_oracle as S

pytestmark = pytest.mark.gpu
TOL = 1e-3
FLIP_TOL = float(__import__('os').environ.get('CONTRAD_FLIP_TOL', '1e-3'))
DEV = 'cuda'
RAW_NORM_TOL = 1e-3       # gradient norms vs the RAW golden (observed 1.8e-4)
AUG_BWD_TOL = 1e-3        # observed 5.8e-6 (profiles/r04_test_margins.txt)


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def l2(a, b):
    a, b = torch.as_tensor(a).double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_nhwc_dot_and_modconv_nodes_against_torch():
    g = torch.Generator().manual_seed(2)
    N, H, W, C = 3, 9, 7, 32
    x = torch.randn(N, H, W, C, generator=g)
    s = torch.rand(N, C, generator=g) + 0.5
    gy = torch.randn(N, H, W, C, generator=g)
    want = (x * gy).sum((1, 2))
    assert rel(ops.nhwc_dot(gy.to(DEV), x.to(DEV)), want) < 1e-5
    nz = torch.randn(N, 1, H, W, generator=g)
    assert rel(ops.nhwc_dot(gy.to(DEV), nz.to(DEV), per_channel=False), (gy * nz.view(N, H, W, 1)).sum((1, 2))) < 1e-5
    # big spatial extent: several row segments per image
    xb, gb = torch.randn(2, 130, 130, 64, generator=g), torch.randn(2, 130, 130, 64, generator=g)
    assert rel(ops.nhwc_dot(gb.to(DEV), xb.to(DEV)), (xb.double() * gb.double()).sum((1, 2))) < 1e-5

    # NhwcScaleFn
    xd, sd = x.to(DEV).requires_grad_(), s.to(DEV).requires_grad_()
    (A.NhwcScaleFn.apply(xd, sd) * gy.to(DEV)).sum().backward()
    xr, sr = x.clone().requires_grad_(), s.clone().requires_grad_()
    ((xr * sr.view(N, 1, 1, C)) * gy).sum().backward()
    assert rel(xd.grad, xr.grad) < 1e-5 and rel(sd.grad, sr.grad) < 1e-5

    # ModconvEpilogueFn
    demod, nw, bias = torch.rand(N, C, generator=g) + 0.5, torch.tensor([0.3]), torch.randn(C, generator=g) * 0.1
    leaves = [t.clone().requires_grad_() for t in (x, demod, nw, bias)]
    pre = leaves[0] * leaves[1].view(N, 1, 1, C) + leaves[2] * nz.view(N, H, W, 1) + leaves[3]
    (torch.nn.functional.leaky_relu(pre, 0.2) * 2 ** 0.5 * gy).sum().backward()
    dl = [t.to(DEV).requires_grad_() for t in (x, demod, nw, bias)]
    out = A.ModconvEpilogueFn.apply(dl[0], dl[1], nz.to(DEV), dl[2], dl[3])
    assert rel(out, torch.nn.functional.leaky_relu(pre, 0.2).detach() * 2 ** 0.5) < 1e-5
    (out * gy.to(DEV)).sum().backward()
    for a, b in zip(dl, leaves):
        assert rel(a.grad, b.grad) < 1e-4


def _build(gseed, dseed):
    G, D = get_architecture('stylegan2', (32, 32, 3))
    gshapes = S.g_param_shapes(32, True)
    gsd = S.fill_kernels(S.det_fill_g(gshapes, seed=gseed), gshapes)
    for k in gsd:
        if k.endswith('noise.weight'):
            gsd[k] = torch.full_like(gsd[k], 0.1)
    G.load_state_dict(gsd)
    D.load_state_dict(S.det_fill_d(S.d_param_shapes(32, True), seed=dseed))
    return G.to(DEV).train(), D.to(DEV).train(), gsd


class _P(object):
    temp, lbd_a, distributed = 0.1, 1.0, False


def test_stylegan2_generator_step_matches_reference(golden, margin):
    g = golden('stylegan2_gstep')
    G, D, _ = _build(int(g['gseed']), int(g['dseed']))
    set_grad(G, True); set_grad(D, False)
    z = torch.from_numpy(g['z']).to(DEV)
    noise = [torch.from_numpy(g['noise%d' % i]).to(DEV) for i in range(G.num_layers)]
    gen = G(z, style_mix=0.9, noise=noise, _mix=(torch.from_numpy(g['z_mix']).to(DEV), torch.from_numpy(g['mix_layer'])))
    assert gen.requires_grad and rel(gen, g['gen']) < TOL
    # the differentiable composition and the fused forward-only launches agree
    with torch.no_grad():
        gen_ng = G(z, style_mix=0.9, noise=noise,
                   _mix=(torch.from_numpy(g['z_mix']).to(DEV), torch.from_numpy(g['mix_layer'])))
    assert rel(gen_ng, gen.detach()) < 1e-5
    P = _P()
    P.augment_fn = SimCLRAugment(scale=(0.2, 1.0))
    seed = int(g['seed'])
    torch.manual_seed(seed); np.random.seed(seed)
    g_loss = hip_contrad.loss_G_fn(P, D, {'loss': 'nonsat'}, None, gen)
    assert abs(g_loss.item() - float(g['g_loss'])) < TOL * abs(float(g['g_loss']))
    g_loss.backward()
    grads = {k: p.grad for k, p in G.named_parameters()}
    assert all(p.grad is None for p in D.parameters())
    assert all(v is not None for v in grads.values())
    for k in g.files:
        if k.startswith('gradnorm/'):
            name = k[len('gradnorm/'):]
            ref = float(g[k])
            if ref < 1e-9:
                assert grads[name].norm().item() < 1e-6, name
            else:
                e = abs(grads[name].norm().item() - ref) / ref
                margin('stylegan2 gstep raw golden/gradnorm/' + name, e, RAW_NORM_TOL)
        elif k.startswith('grad/'):
            name = k[5:]
            if float(g['gradnorm/' + name]) >= 1e-9:
                margin('stylegan2 gstep raw golden/grad-l2/' + name, l2(grads[name], g[k]), FLIP_TOL)


def test_stylegan2_generator_step_on_the_same_linear_region(margin):
    """Strict element-wise check against the oracle with D's leaky-relu regions recorded from the HIP run."""
    B = 4
    G, D, gsd = _build(780, 2026)
    set_grad(G, True); set_grad(D, False)
    D._record_activations = True
    gg = torch.Generator().manual_seed(61)
    z = torch.randn(B, 512, generator=gg)
    z_mix = torch.randn(B, 512, generator=gg)
    mix_layer = torch.tensor([3, G.n_latent, 1, 5])
    noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=gg) for i in range(G.num_layers)]
    P = _P()
    P.augment_fn = SimCLRAugment(scale=(0.2, 1.0))
    gen = G(z.to(DEV), style_mix=0.9, noise=[n.to(DEV) for n in noise], _mix=(z_mix.to(DEV), mix_layer))
    torch.manual_seed(17); np.random.seed(17)
    g_loss = hip_contrad.loss_G_fn(P, D, {'loss': 'nonsat'}, None, gen)
    g_loss.backward()
    rec, hl, hpq = D._recorded[0]
    masks = [(t > 0).permute(0, 3, 1, 2).cpu() for t in rec]
    hl, hpq = hl.reshape(B, -1).cpu(), hpq.reshape(B, -1).cpu()
    osd = {k: v.clone() for k, v in gsd.items()}
    names = [k for k, _ in G.named_parameters()]
    for k in names:
        osd[k].requires_grad_()
    ogen = S.g_forward(osd, z, 32, noise, mix=(z_mix, mix_layer))
    torch.manual_seed(17); np.random.seed(17)
    p = O.sample_simclr_params(B, 32, 32, O.SIMCLR_CIFAR)
    dsd = S.det_fill_d(S.d_param_shapes(32, True), seed=2026)
    od = S.d_forward(dsd, O.simclr_apply(ogen, p), 32, sg_linear=False, masks=masks,
                     head_masks=(hl > 0, hpq[:, :512] > 0, hpq[:, 512:] > 0))[0]
    ol = O.gan_g_loss(od, 'nonsat')
    ol.backward()
    assert rel(gen, ogen.detach()) < TOL
    assert abs(g_loss.item() - ol.item()) < TOL * abs(ol.item())
    for k, prm in G.named_parameters():
        ref = osd[k].grad
        if ref.norm().item() < 1e-9:
            continue
        # G's own leaky-relus / the augmentation's clamps can still flip single units: L2 criterion
        margin('stylegan2 gstep same-region/grad-l2/' + k, l2(prm.grad, ref), TOL)


@pytest.mark.parametrize('size,B', [(96, 5), (512, 3)])
def test_large_image_augment_backward_matches_oracle(size, B, margin):
    """d(sum(out * w)) / d(images) through simclr_hq at sizes beyond the LDS-resident path: gather transpose, contrast,
    straight-through HSV, gray, and the masked Gaussian blur's reflect-padding transpose (ksize 9 / 51)."""
    from sg2_inputs import seeded_images
    torch.manual_seed(size); np.random.seed(size)
    x = seeded_images(B, size, 91)
    w = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(1))
    p = O.sample_simclr_params(B, size, size, O.SIMCLR_HQ_AFHQ)
    p['jitter_mask'][:] = torch.tensor([1., 1., 0., 1., 1.][:B])
    p['gray_mask'][:] = torch.tensor([0., 1., 0., 0., 1.][:B])
    p['blur_mask'][:] = torch.tensor([1., 0., 1., 1., 0.][:B])
    aug = SimCLRAugment(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
                        sigma_range=(0.1, 2.0))
    P = torch.zeros(B, ops.AUG_NPARAM)
    th = p['theta']
    P[:, 0], P[:, 1], P[:, 2], P[:, 3] = th[:, 0, 0], th[:, 1, 1], th[:, 0, 2], th[:, 1, 2]
    for i, k in enumerate(['flip_sign', 'jitter_mask', 'f_contrast', 'f_h', 'f_s', 'f_v', 'gray_mask', 'blur_mask']):
        P[:, 4 + i] = p[k]
    for cf in (True, False):
        p['contrast_first'] = cf
        xr = x.clone().requires_grad_()
        (O.simclr_apply(xr, p) * w).sum().backward()
        xd = x.to(DEV).requires_grad_()
        out = aug.apply(xd, P, cf, p['sigma'])
        (out * w.to(DEV)).sum().backward()
        margin('augment backward %d^2 contrast_first=%s (l2)' % (size, cf), l2(xd.grad, xr.grad), AUG_BWD_TOL)


def test_blur_adjoint_identity():
    """<blur(x), g> == <x, blur^T(g)> for the masked separable blur with reflect padding."""
    B, H, R = 2, 70, 9
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, 3, H, H, generator=g).to(DEV)
    gy = torch.randn(B, 3, H, H, generator=g).to(DEV)
    _, k1 = SimCLRAugment.blur_kernel(512, 1.3)
    k1 = k1[25 - R:25 + R + 1].contiguous()
    k1 = (k1 / k1.sum()).to(DEV)
    P = torch.zeros(B, ops.AUG_NPARAM); P[:, 11] = torch.tensor([1., 0.])
    P = P.to(DEV)
    y = ops.gaussian_blur_masked(x, P, k1, R)
    gx = ops.gaussian_blur_masked_bwd(gy, P, k1, R)
    a, b = (y.double() * gy.double()).sum().item(), (x.double() * gx.double()).sum().item()
    assert abs(a - b) < 1e-5 * abs(a) + 1e-4
    assert torch.equal(gx[1], gy[1])


def test_ema_accumulate_matches_reference_formula():
    """utils.accumulate (utils.py:130-143): dst = decay*dst + (1-decay)*src on parameters, buffers copied; the packed
    weight cache of the EMA generator must see the update."""
    from contrad_amd.train_stylegan2 import accumulate
    torch.manual_seed(0)
    G, _ = get_architecture('stylegan2', (32, 32, 3))
    Ge, _ = get_architecture('stylegan2', (32, 32, 3))
    G, Ge = G.to(DEV), Ge.to(DEV).eval()
    z = torch.randn(2, 512, device=DEV)
    with torch.no_grad():
        before = Ge(z).clone()
    want = {k: 0.75 * p.detach().clone() + 0.25 * dict(G.named_parameters())[k].detach() for k, p in Ge.named_parameters()}
    accumulate(Ge, G, 0.75)
    for k, p in Ge.named_parameters():
        assert rel(p, want[k]) < 1e-6, k
    with torch.no_grad():
        after = Ge(z)
    assert (after - before).abs().max().item() > 1e-4
    accumulate(Ge, G, 0)                  # decay 0 (before ema_start): plain copy
    for k, p in Ge.named_parameters():
        assert torch.equal(p, dict(G.named_parameters())[k])


@pytest.mark.parametrize('script', ['train_stylegan2', 'train_stylegan2_contraD'])
def test_stylegan2_training_scripts_run_checkpoint_and_resume(tmp_path, script):
    """Both loops through their CLI: G-step first, lazy R1 (d_reg_every 2 so that it fires), EMA, checkpoints
    (gen.pt / dis.pt / gen_ema.pt / optim.pt), resume."""
    import importlib
    main = importlib.import_module('contrad_amd.' + script).main
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gin = os.path.join(root, 'configs', 'gan', 'stylegan2', 'c10_style64.gin')
    logdir = str(tmp_path / 'run')
    common = [gin, 'stylegan2', '--mode=contrad', '--aug=simclr', '--lbd_r1', '0.1', '--d_reg_every', '2',
              '--synthetic', '--batch_size', '8', '--halflife_k', '1', '--ema_start_k', '0', '--print_every', '1']
    main(common + ['--max_steps', '4', '--evaluate_every', '4', '--logdir', logdir])
    for f in ('gen.pt', 'dis.pt', 'gen_ema.pt', 'optim.pt', 'log.txt'):
        assert os.path.exists(os.path.join(logdir, f)), f
    sd = torch.load(os.path.join(logdir, 'gen_ema.pt'))
    assert all(torch.isfinite(v).all() for v in sd.values())
    assert torch.load(os.path.join(logdir, 'optim.pt'))['epoch'] == 4
    main(common + ['--max_steps', '6', '--resume', logdir])
    log = open(os.path.join(logdir, 'log.txt')).read()
    assert '[Steps       6]' in log and 'nan' not in log.lower() and '[r1 ' in log
