"""GPU parity of the remaining kernels (weight prep / spectral norm, RGB convs, column statistics, BatchNorm,
GAN losses, Adam, fused augmentation) through the C ABI, against the oracle / PyTorch-CPU fp32 and the goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from contrad_amd import ops
from oracle import contrad_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3
DEV = 'cuda'


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize('shape', [(64, 3, 3, 3), (128, 64, 4, 4), (512, 512, 3, 3), (512, 8192), (1, 512), (128, 512)])
@pytest.mark.parametrize('training', [True, False])
def test_spectral_norm_weight_prep_and_grad(shape, training):
    g = torch.Generator().manual_seed(len(shape) * 100 + shape[0])
    w = torch.randn(*shape, generator=g) * 0.02
    K = shape[0]
    IN = w.numel() // K
    u = F.normalize(torch.randn(K, generator=g), dim=0)
    v = F.normalize(torch.randn(IN, generator=g), dim=0)
    sd = {'l.weight_orig': w.clone().requires_grad_(), 'l.weight_u': u.clone(), 'l.weight_v': v.clone()}
    w_eff = O.spectral_norm_weight(sd, 'l', training=training)
    gw_eff = torch.randn(w_eff.shape, generator=g)
    (w_eff * gw_eff).sum().backward()

    wd, ud, vd = w.to(DEV), u.to(DEV), v.to(DEV)
    spec = ops.SnSpec(wd, ud, vd)
    ldw = ops.round_up(K, 4)
    wp = torch.zeros(spec.T * spec.C, ldw, device=DEV)
    offs, n = ops.sn_scratch_floats([spec])
    scratch = torch.empty(n, device=DEV)
    sigma = torch.empty(1, device=DEV)
    us, vs = torch.empty(K, device=DEV), torch.empty(IN, device=DEV)
    ops.sn_weight_prep([spec], [wp], [ldw], training, scratch, offs, sigma, [us], [vs])
    w4 = shape if len(shape) == 4 else (K, IN, 1, 1)
    got = ops.unpack_weight(wp.cpu(), K, w4[1], w4[2], w4[3]).reshape(shape)
    assert rel(got, w_eff.detach()) < TOL
    assert rel(ud, sd['l.weight_u']) < TOL and rel(vd, sd['l.weight_v']) < TOL
    assert torch.equal(us, ud) and torch.equal(vs, vd)
    # backward
    gwp = ops.pack_weight(gw_eff.reshape(w4)).to(DEV)
    gw = torch.empty_like(wd)
    ops.sn_weight_grad([spec], [wp], [ldw], [gwp], [gw], scratch, offs, sigma, [us], [vs])
    assert rel(gw, sd['l.weight_orig'].grad) < TOL


def test_fixed_scale_weight_prep():
    w = torch.randn(32, 16, 3, 3)
    spec = ops.SnSpec(w.to(DEV), fixed_scale=0.25)
    wp = torch.zeros(9 * 16, 32, device=DEV)
    offs, n = ops.sn_scratch_floats([spec])
    scratch, sigma = torch.empty(n, device=DEV), torch.empty(1, device=DEV)
    ops.sn_weight_prep([spec], [wp], [32], True, scratch, offs, sigma)
    assert rel(ops.unpack_weight(wp.cpu(), 32, 16, 3, 3), w * 0.25) < 1e-6
    gwp = torch.randn(9 * 16, 32, device=DEV)
    gw = torch.empty(32, 16, 3, 3, device=DEV)
    ops.sn_weight_grad([spec], [wp], [32], [gwp], [gw], scratch, offs, sigma)
    assert rel(gw, ops.unpack_weight(gwp.cpu(), 32, 16, 3, 3) * 0.25) < 1e-6


@pytest.mark.parametrize('K,k,H', [(64, 3, 32), (128, 1, 32), (32, 1, 40), (64, 3, 12)])
def test_rgb_conv_fwd_wgrad_dgrad(K, k, H):
    g = torch.Generator().manual_seed(K + k)
    N = 5
    img = torch.rand(N, 3, H, H, generator=g)
    w = torch.randn(K, 3, k, k, generator=g) * 0.2
    b = torch.randn(K, generator=g) * 0.1
    ir, wr, br = img.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y_lin = F.conv2d(ir * 2 - 1, wr, br, padding=k // 2)
    y_ref = F.leaky_relu(y_lin, 0.1)
    gy = torch.randn(y_lin.shape, generator=g)
    gi, gw, gb = torch.autograd.grad(y_lin, (ir, wr, br), gy)
    wp = ops.pack_weight(w).to(DEV)
    y = ops.rgb_conv_fwd(img.to(DEV), wp, b.to(DEV), K, k, 2.0, -1.0, 0.1, 1.0)
    assert rel(y.permute(0, 3, 1, 2), y_ref.detach()) < TOL
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dwp = torch.zeros_like(wp)
    db = torch.zeros(K, device=DEV)
    ops.rgb_conv_wgrad(img.to(DEV), gyd, k, 2.0, -1.0, dwp, db)
    assert rel(ops.unpack_weight(dwp.cpu(), K, 3, k, k), gw) < TOL
    assert rel(db, gb) < TOL
    di = ops.rgb_conv_dgrad(gyd, wp, None, 3, k, act=0, out_scale=2.0)
    assert rel(di, gi) < TOL


def test_generator_last_layer_tanh():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 64, 16, 16, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.1
    b = torch.randn(3, generator=g) * 0.1
    ref = 0.5 * torch.tanh(F.conv_transpose2d(x, w, b, stride=1, padding=1)) + 0.5
    out = ops.rgb_conv_dgrad(x.permute(0, 2, 3, 1).contiguous().to(DEV), ops.pack_weight(w).to(DEV), b.to(DEV), 3, 3,
                             act=1, out_scale=0.5, out_shift=0.5)
    assert rel(out, ref) < TOL


@pytest.mark.parametrize('M,K', [(1000, 128), (37, 1), (5000, 1536), (4096, 8192), (20000, 64), (1003, 68), (513, 4), (31, 516)])
def test_colstats(M, K):
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g)
    s = ops.colstats(x.to(DEV), with_sq=True)
    assert rel(s[0], x.double().sum(0)) < TOL and rel(s[1], (x.double() ** 2).sum(0)) < TOL
    s1 = ops.colstats(x.to(DEV), with_sq=False)
    assert torch.equal(s1[0], s[0])


def test_bn_relu_and_running_stats():
    g = torch.Generator().manual_seed(1)
    N, C, H = 6, 64, 8
    x = torch.randn(N, C, H, H, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    cb = torch.randn(C, generator=g) * 0.1
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.relu(F.batch_norm(x + cb.view(1, C, 1, 1), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    x2 = ops.as_rows(xd)
    stats = ops.colstats(x2, with_sq=True)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    ops.bn_running_update(stats, float(N * H * H), cb.to(DEV), 0.1, rmd, rvd)
    ops.bn_relu_apply(x2, x2, stats, float(N * H * H), gamma.to(DEV), beta.to(DEV), 1e-5)
    assert rel(xd.permute(0, 3, 1, 2), ref) < TOL
    assert rel(rmd, rm_ref) < TOL and rel(rvd, rv_ref) < TOL


@pytest.mark.parametrize('kind', ['nonsat', 'wgan', 'hinge', 'lsgan'])
def test_gan_losses(kind):
    N = 37
    g = torch.Generator().manual_seed(2)
    d = (torch.randn(3 * N, 1, generator=g) * 3).requires_grad_()
    ref = O.gan_d_loss(d[:N], d[2 * N:], kind)
    ref.backward()
    out, grad = ops.gan_d_loss(d.detach().to(DEV), N, kind)
    assert abs(out[0].item() - ref.item()) < TOL * max(1, abs(ref.item()))
    assert rel(grad, d.grad) < TOL
    assert abs(out[1].item() - d[:N].mean().item()) < 1e-5 and abs(out[2].item() - d[2 * N:].mean().item()) < 1e-5
    dg = d.detach()[:N].clone().requires_grad_()
    refg = (F.softplus(-dg).mean() if kind == 'nonsat' else
            (0.5 * ((dg - 1.0) ** 2).mean() if kind == 'lsgan' else -dg.mean()))
    refg.backward()
    og, gg = ops.gan_g_loss(dg.detach().to(DEV), kind)
    assert abs(og.item() - refg.item()) < TOL * max(1, abs(refg.item())) and rel(gg, dg.grad) < TOL


def test_adam_trajectory(golden):
    for tag in ('c10', 'sg2'):
        g = golden('adam_' + tag)
        p = torch.from_numpy(g['p0']).to(DEV)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for t, gr in enumerate(torch.from_numpy(g['grads']), 1):
            ops.adam_step([p], [gr.to(DEV)], [m], [v], t, float(g['lr']), float(g['b1']), float(g['b2']))
            assert rel(p, torch.from_numpy(g['traj'][t - 1])) < 1e-5


def _params_from_golden(g, tag, B):
    P = torch.zeros(B, ops.AUG_NPARAM)
    th = torch.from_numpy(g[tag + '_p_theta'])
    P[:, 0], P[:, 1], P[:, 2], P[:, 3] = th[:, 0, 0], th[:, 1, 1], th[:, 0, 2], th[:, 1, 2]
    P[:, 4] = torch.from_numpy(g[tag + '_p_flip_sign'])
    P[:, 5] = torch.from_numpy(g[tag + '_p_jitter_mask'])
    P[:, 6] = torch.from_numpy(g[tag + '_p_f_contrast'])
    P[:, 7] = torch.from_numpy(g[tag + '_p_f_h'])
    P[:, 8] = torch.from_numpy(g[tag + '_p_f_s'])
    P[:, 9] = torch.from_numpy(g[tag + '_p_f_v'])
    P[:, 10] = torch.from_numpy(g[tag + '_p_gray_mask'])
    if (tag + '_p_blur_mask') in g.files:
        P[:, 11] = torch.from_numpy(g[tag + '_p_blur_mask'])
    return P, bool(g[tag + '_p_contrast_first'])


@pytest.mark.parametrize('tag', ['c10a', 'c10b'])
def test_simclr_augment_against_reference_golden(golden, tag):
    g = golden('augment')
    x = torch.from_numpy(g[tag + '_x'])
    P, cf = _params_from_golden(g, tag, x.shape[0])
    out = ops.simclr_augment(x.to(DEV), P.to(DEV), cf, True)
    ref = torch.from_numpy(g[tag + '_out'])
    err = (out.cpu() - ref).abs().max().item()
    assert err < 1e-4, err          # values in [0,1]: absolute == relative to max
    # host sampler of the product reproduces the reference's RNG draw order bit for bit
    from contrad_amd.augment import SimCLRAugment
    aug = SimCLRAugment(scale=(0.2, 1.0))
    seed = int(g[tag + '_seed'])
    torch.manual_seed(seed); np.random.seed(seed)
    P2, cf2, _ = aug.sample(x.shape[0], 32, 32)
    assert cf2 == cf and torch.equal(P2[:, :11], P[:, :11])


def test_simclr_hq_large_path_and_blur(golden):
    g = golden('augment')
    x = torch.from_numpy(g['hq_x'])
    P, cf = _params_from_golden(g, 'hq', x.shape[0])
    from contrad_amd.augment import SimCLRAugment
    aug = SimCLRAugment(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
                        sigma_range=(0.1, 2.0))
    out = aug.apply(x.to(DEV), P, cf, float(g['hq_p_sigma']))
    ref = torch.from_numpy(g['hq_out'])
    assert (out.cpu() - ref).abs().max().item() < 1e-4
    seed = int(g['hq_seed'])
    torch.manual_seed(seed); np.random.seed(seed)
    P2, cf2, sigma2 = aug.sample(x.shape[0], 64, 64)
    assert cf2 == cf and torch.equal(P2[:, :12], P[:, :12]) and abs(sigma2 - float(g['hq_p_sigma'])) < 1e-12


def test_simclr_large_image_two_pass_matches_oracle():
    """> 64 KiB per image takes the statistics + apply path (AFHQ-like); oracle on explicit parameters."""
    B, H = 4, 96
    torch.manual_seed(3); np.random.seed(3)
    x = torch.rand(B, 3, H, H)
    p = O.sample_simclr_params(B, H, H, O.SIMCLR_HQ_AFHQ)
    p['jitter_mask'][:] = torch.tensor([1., 1., 0., 1.])
    p['blur_mask'][:] = torch.tensor([1., 0., 1., 1.])
    ref = O.simclr_apply(x, p)
    P = torch.zeros(B, ops.AUG_NPARAM)
    th = p['theta']
    P[:, 0], P[:, 1], P[:, 2], P[:, 3] = th[:, 0, 0], th[:, 1, 1], th[:, 0, 2], th[:, 1, 2]
    for i, k in enumerate(['flip_sign', 'jitter_mask', 'f_contrast', 'f_h', 'f_s', 'f_v', 'gray_mask', 'blur_mask']):
        P[:, 4 + i] = p[k]
    from contrad_amd.augment import SimCLRAugment
    aug = SimCLRAugment(scale=(0.08, 1.0), brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2, p_blur=0.5,
                        sigma_range=(0.1, 2.0))
    out = aug.apply(x.to(DEV), P, p['contrast_first'], p['sigma'])
    assert (out.cpu() - ref).abs().max().item() < 1e-4
