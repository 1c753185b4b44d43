#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/*.npz.

Runs ONLY in the build container: imports the reference (read-only, /root/reference) through the
shims in _refshim.py, executes its CPU PyTorch path on small seeded inputs, ASSERTS that the
oracle (oracle/contrad_oracle.py) reproduces every output, and stores inputs + explicit random
parameters + reference outputs as small .npz fixtures.  The reference never travels; the fixtures
are data only.

    python tests/golden/make_golden.py            # regenerate everything
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _refshim  # noqa: E402

_refshim.install()
from oracle import contrad_oracle as O  # noqa: E402

torch.set_num_threads(8)


def _np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        elif isinstance(v, (bool, int, float)):
            out[k] = np.asarray(v)
        elif v is None:
            continue
        else:
            out[k] = np.asarray(v)
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **_np(arrays))
    print('wrote %-28s %7.1f KiB' % (name + '.npz', os.path.getsize(path) / 1024))


def check(a, b, tol, what):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1e-30)
    assert err <= tol * max(ref, 1.0), '%s: max err %.3e (ref max %.3e)' % (what, err, ref)
    return err


# ------------------------------------------------------------------------------------------------
def gen_losses():
    from training.criterion import nt_xent
    from training.gan.contrad import supcon_fake
    # the SURVEY known-answer smoke values
    torch.manual_seed(0)
    z = F.normalize(torch.randn(12, 16))
    a = nt_xent(z[:4], z[4:8], temperature=0.1).item()
    b = supcon_fake(z[:4], z[4:8], z[8:], temperature=0.1).item()
    assert abs(a - 4.954558372497559) < 1e-6 and abs(b - 3.962817430496216) < 1e-6, (a, b)

    out = {}
    for tag, (N, d, temp) in {'small': (4, 16, 0.1), 'mid': (24, 128, 0.1), 'hot': (16, 128, 0.5)}.items():
        g = torch.Generator().manual_seed(7 + N)
        u1 = torch.randn(3 * N, d, generator=g, requires_grad=True)     # un-normalised projection
        u2 = torch.randn(3 * N, d, generator=g, requires_grad=True)     # un-normalised projection2
        v = F.normalize(u1)
        r = F.normalize(u2)
        l1 = nt_xent(v[:N], v[N:2 * N], temperature=temp)
        l2 = supcon_fake(r[:N], r[N:2 * N], r[2 * N:], temperature=temp)
        (l1 + l2).backward()
        # oracle
        o1 = u1.detach().clone().requires_grad_()
        o2 = u2.detach().clone().requires_grad_()
        ov, orr = F.normalize(o1), F.normalize(o2)
        ol1 = O.nt_xent(ov[:N], ov[N:2 * N], temp)
        ol2 = O.supcon_fake(orr[:N], orr[N:2 * N], orr[2 * N:], temp)
        (ol1 + ol2).backward()
        check(ol1, l1, 1e-6, 'nt_xent ' + tag)
        check(ol2, l2, 1e-6, 'supcon ' + tag)
        check(o1.grad, u1.grad, 1e-6, 'nt_xent grad ' + tag)
        check(o2.grad, u2.grad, 1e-6, 'supcon grad ' + tag)
        out.update({tag + '_u1': u1, tag + '_u2': u2, tag + '_temp': temp, tag + '_N': N,
                    tag + '_nt_xent': l1, tag + '_supcon': l2,
                    tag + '_g1': u1.grad, tag + '_g2': u2.grad})
    out['known_nt_xent'] = a
    out['known_supcon'] = b
    save('losses', **out)


# ------------------------------------------------------------------------------------------------
def _param_dict(p):
    return {('p_' + k): (v if v is not None else None) for k, v in p.items()}


def gen_augment():
    import augment as A
    out = {}
    # --- CIFAR simclr, B=12 (enough samples that every branch of every mask occurs) -------------
    _refshim.bind_cifar_defaults()
    for tag, seed in (('c10a', 0), ('c10b', 3)):
        B = 12
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.rand(B, 3, 32, 32, generator=g)
        torch.manual_seed(seed); np.random.seed(seed)
        ref = A.simclr()(x)
        torch.manual_seed(seed); np.random.seed(seed)
        p = O.sample_simclr_params(B, 32, 32, O.SIMCLR_CIFAR)
        mine = O.simclr_apply(x, p)
        check(mine, ref, 1e-6, 'simclr ' + tag)
        out.update({tag + '_x': x, tag + '_out': ref, tag + '_seed': seed})
        out.update({tag + '_' + k: v for k, v in _param_dict(p).items()})
        # per-stage outputs with the same explicit parameters (reference layers driven directly)
        st1 = O.resized_crop(x, p['theta'])
        st2 = O.hflip(st1, p['flip_sign'])
        out.update({tag + '_stage_crop': st1, tag + '_stage_flip': st2})

    # --- stage-level checks against the reference layers with injected randomness --------------
    g = torch.Generator().manual_seed(55)
    x = torch.rand(6, 3, 32, 32, generator=g)
    from augment.color_jitter import RandomHSVFunction
    from augment.utils import rgb2hsv, hsv2rgb
    f_h = torch.tensor([-0.1, 0.05, 0.0, 0.1, -0.03, 0.07]).view(6, 1, 1)
    f_s = torch.tensor([0.6, 1.4, 1.0, 0.9, 1.2, 0.7]).view(6, 1, 1)
    f_v = torch.tensor([1.4, 0.6, 1.0, 1.1, 0.8, 1.3]).view(6, 1, 1)
    xe = x.clone()
    xe[0, :, 0, 0] = 0.0                       # black pixel  -> Cmax = 0 path
    xe[0, :, 0, 1] = 0.5                       # gray pixel   -> atan2(0,0)
    xe[0, :, 0, 2] = torch.tensor([1.0, 0.0, 0.0])
    xe[0, :, 0, 3] = torch.tensor([0.0, 0.0, 1.0])
    ref_hsv = rgb2hsv(xe)
    ref_rt = hsv2rgb(ref_hsv)
    ref_adj = RandomHSVFunction.apply(xe, f_h, f_s, f_v)
    check(O.rgb2hsv(xe), ref_hsv, 1e-7, 'rgb2hsv')
    check(O.hsv2rgb(ref_hsv), ref_rt, 1e-7, 'hsv2rgb')
    check(O.adjust_hsv(xe, f_h.view(6), f_s.view(6), f_v.view(6)), ref_adj, 1e-7, 'adjust_hsv')
    out.update({'hsv_x': xe, 'hsv_fh': f_h.view(6), 'hsv_fs': f_s.view(6), 'hsv_fv': f_v.view(6),
                'hsv_hsv': ref_hsv, 'hsv_roundtrip': ref_rt, 'hsv_adjusted': ref_adj})

    cj = A.ColorJitterLayer()
    fc = torch.tensor([0.6, 1.4, 1.0, 0.8, 1.25, 1.1])

    class _Fixed(object):
        def __init__(self, vals): self.vals = list(vals)
    # contrast via the reference method with the factor injected through the RNG hook
    orig_uniform = torch.Tensor.uniform_
    try:
        torch.Tensor.uniform_ = lambda self, *a, **k: self.copy_(fc.view_as(self))
        ref_con = cj.adjust_contrast(x)
    finally:
        torch.Tensor.uniform_ = orig_uniform
    check(O.adjust_contrast(x, fc), ref_con, 1e-7, 'contrast')
    ref_gray = A.RandomColorGrayLayer()(x)
    check(O.color_gray(x), ref_gray, 1e-7, 'gray')
    out.update({'con_x': x, 'con_f': fc, 'con_out': ref_con, 'gray_out': ref_gray})

    # --- flip is an exact permutation ---------------------------------------
    sign = torch.tensor([1., -1., -1., 1., -1., 1.])
    fl = O.hflip(x, sign)
    for i in range(6):
        exp = x[i] if sign[i] > 0 else x[i].flip(-1)
        assert torch.equal(fl[i], exp), 'flip not exact'

    # --- simclr_hq on a small "hq-like" image (64x64 -> ksize 7) with AFHQ strengths ----------
    _refshim.bind_afhq()
    B = 8
    g = torch.Generator().manual_seed(321)
    x = torch.rand(B, 3, 64, 64, generator=g)
    torch.manual_seed(5); np.random.seed(5)
    ref = A.simclr_hq()(x)
    torch.manual_seed(5); np.random.seed(5)
    p = O.sample_simclr_params(B, 64, 64, O.SIMCLR_HQ_AFHQ)
    mine = O.simclr_apply(x, p)
    check(mine, ref, 1e-6, 'simclr_hq')
    out.update({'hq_x': x, 'hq_out': ref, 'hq_seed': 5})
    out.update({'hq_' + k: v for k, v in _param_dict(p).items()})

    # --- simclr_hq_cutout (augment/__init__.py:124-133), CutOut.length = 15 (augment.gin) ---
    _refshim.bind('CutOut', length=15)
    B = 8
    g = torch.Generator().manual_seed(322)
    x = torch.rand(B, 3, 64, 64, generator=g)
    torch.manual_seed(6); np.random.seed(6)
    ref = A.simclr_hq_cutout()(x)
    torch.manual_seed(6); np.random.seed(6)
    p = O.sample_simclr_params(B, 64, 64, O.SIMCLR_HQ_CUTOUT_AFHQ)
    check(O.simclr_apply(x, p), ref, 1e-6, 'simclr_hq_cutout')
    out.update({'cut_x': x, 'cut_out': ref, 'cut_seed': 6})
    out.update({'cut_' + k: v for k, v in _param_dict(p).items()})
    _refshim.bind_cifar_defaults()
    save('augment', **out)


# ------------------------------------------------------------------------------------------------
def _load_sd(module, sd):
    missing = module.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return missing


def gen_sndcgan():
    from models.gan import get_architecture
    from training.gan import contrad as ref_contrad
    from argparse import Namespace
    G, D = get_architecture('sndcgan', (32, 32, 3))
    D.train(); G.train()

    # state-dict key contract: names + shapes must match the oracle's table
    shapes = O.sndcgan_d_param_shapes()
    ref_shapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    assert ref_shapes == shapes, 'D state-dict mismatch'
    assert list(ref_shapes) == list(shapes), 'D state-dict ORDER'
    gshapes = O.sndcgan_g_param_shapes()
    ref_gshapes = {k: tuple(v.shape) for k, v in G.state_dict().items() if 'num_batches' not in k}
    assert ref_gshapes == gshapes, (set(ref_gshapes) ^ set(gshapes))
    assert list(ref_gshapes) == list(gshapes), 'G state-dict ORDER' 

    sd = O.det_fill(shapes, seed=1234)
    _load_sd(D, sd)
    gsd = O.det_fill(gshapes, seed=4321)
    gfull = dict(G.state_dict())
    gfull.update({k: v.clone() for k, v in gsd.items()})
    G.load_state_dict(gfull)

    N = 4
    g = torch.Generator().manual_seed(99)
    x = torch.rand(N, 3, 32, 32, generator=g)
    z = torch.rand(N, 128, generator=g) * 2 - 1

    # ---- G forward (train-mode BN) ----
    with torch.no_grad():
        fake = G(z)
    osd = {k: v.clone() for k, v in gsd.items()}
    with torch.no_grad():
        ofake = O.sndcgan_g_forward(osd, z)
    check(ofake, fake, 1e-6, 'G forward')
    gafter = G.state_dict()
    for k in osd:
        check(osd[k], gafter[k], 1e-6, 'G buffer ' + k)

    # ---- contrad D loss with explicit augmentation = identity-free: feed pre-augmented images ----
    torch.manual_seed(11); np.random.seed(11)
    p = O.sample_simclr_params(3 * N, 32, 32, O.SIMCLR_CIFAR)
    cat = torch.cat([x, x, fake], 0)
    aug = O.simclr_apply(cat, p)

    P = Namespace(augment_fn=lambda t: aug, temp=0.1, lbd_a=1.0, distributed=False)
    D.zero_grad()
    d_loss, aux = ref_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, x, fake)
    (d_loss + aux['penalty']).backward()
    ref_grads = {k: v.grad.clone() for k, v in D.named_parameters()}
    ref_after = {k: v.clone() for k, v in D.state_dict().items()}

    # oracle
    osd = {k: v.clone() for k, v in sd.items()}
    for k in osd:
        if k.endswith('weight_orig') or k.endswith('bias'):
            osd[k].requires_grad_()
    fwd = lambda t: O.sndcgan_d_forward(osd, t, sg_linear=True)[:3]
    closs, gloss, dr, dg = O.contrad_loss_d(fwd, aug, N)
    (closs + gloss).backward()
    check(closs, d_loss, 1e-6, 'contrad loss')
    check(gloss, aux['penalty'], 1e-6, 'gan loss')
    gerr = 0.0
    for k, gref in ref_grads.items():
        gerr = max(gerr, check(osd[k].grad, gref, 2e-5, 'grad ' + k))
    for k in osd:
        if k.endswith('weight_u') or k.endswith('weight_v'):
            check(osd[k], ref_after[k], 1e-6, 'sn buffer ' + k)
    print('  sndcgan D: max grad err vs reference %.2e' % gerr)

    # forward outputs at the three heads
    with torch.no_grad():
        D2 = get_architecture('sndcgan', (32, 32, 3))[1]
        D2.train(); _load_sd(D2, sd)
        logit, auxo = D2(aug, sg_linear=True, projection=True, projection2=True, penultimate=True)

    # ---- one Adam step on D (lr 2e-4, betas (.5,.999)) ----
    opt = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt.step()
    ref_post = {k: v.detach().clone() for k, v in D.named_parameters()}
    omax = 0.0
    for k, pref in ref_post.items():
        pp = osd[k].detach().clone()
        O.adam_step(pp, osd[k].grad, torch.zeros_like(pp), torch.zeros_like(pp), 1, 2e-4, 0.5, 0.999)
        omax = max(omax, check(pp, pref, 1e-6, 'adam ' + k))

    out = {'x': x, 'z': z, 'fake': fake, 'aug': aug, 'N': N,
           'logit': logit, 'projection': auxo['projection'], 'projection2': auxo['projection2'],
           'penultimate_head': auxo['penultimate'][:, :64], 'penultimate_sum': auxo['penultimate'].sum(1),
           'contrad_loss': d_loss, 'gan_loss': aux['penalty'], 'd_real': aux['d_real'], 'd_gen': aux['d_gen']}
    for k, gref in ref_grads.items():
        out['gradnorm/' + k] = gref.norm()
        if gref.numel() <= 4096:
            out['grad/' + k] = gref
        else:
            out['gradhead/' + k] = gref.reshape(-1)[:512]
    for k, v in ref_after.items():
        if k.endswith('weight_u'):
            out['after/' + k] = v
        elif k.endswith('weight_v'):
            out['afterhead/' + k] = v[:512]
    for k, pref in ref_post.items():
        out['adamsum/' + k] = pref.double().sum()
        out['adamhead/' + k] = pref.reshape(-1)[:256]
    for k in ('norm_init.running_mean', 'main.1.running_mean', 'main.1.running_var', 'main.7.running_var'):
        out['gbuf/' + k] = gafter[k][:256]
    out.update({'p_' + k: v for k, v in p.items() if v is not None})
    save('sndcgan', **out)


# ------------------------------------------------------------------------------------------------
def gen_sndcgan_eval():
    """G_SNDCGAN in eval mode (running statistics; what ``gen.pt`` is sampled with, train_gan.py:181, sndcgan.py:41-52)."""
    from models.gan import get_architecture
    G, _D = get_architecture('sndcgan', (32, 32, 3))
    gshapes = O.sndcgan_g_param_shapes()
    gsd = O.det_fill(gshapes, seed=4321)
    gfull = dict(G.state_dict())
    gfull.update({k: v.clone() for k, v in gsd.items()})
    G.load_state_dict(gfull)
    G.eval()
    z = torch.rand(5, 128, generator=torch.Generator().manual_seed(98)) * 2 - 1
    with torch.no_grad():
        img = G(z)
        check(O.sndcgan_g_forward({k: v.clone() for k, v in gsd.items()}, z, training=False), img, 1e-6, 'G eval forward')
    for k, v in G.state_dict().items():                      # eval mode leaves every buffer alone
        if k in gsd:
            assert torch.equal(v, gsd[k]), k
    save('sndcgan_eval', z=z, img=img, wseed=4321)


# ------------------------------------------------------------------------------------------------
def gen_sndcgan_gstep():
    """Generator step (train_gan.py:170-177): G(z) with grad -> loss_G_fn = softplus(-D(augment(G(z)))) -> G grads."""
    import augment as A
    from models.gan import get_architecture
    from training.gan import contrad as ref_contrad
    from argparse import Namespace
    _refshim.bind_cifar_defaults()
    G, D = get_architecture('sndcgan', (32, 32, 3))
    G.train(); D.train()
    sd = O.det_fill(O.sndcgan_d_param_shapes(), seed=1234)
    _load_sd(D, sd)
    gsd = O.det_fill(O.sndcgan_g_param_shapes(), seed=4321)
    gfull = dict(G.state_dict()); gfull.update({k: v.clone() for k, v in gsd.items()}); G.load_state_dict(gfull)
    for p in D.parameters():
        p.requires_grad = False
    N = 6
    g = torch.Generator().manual_seed(123)
    z = torch.rand(N, 128, generator=g) * 2 - 1
    aug = A.simclr()
    seed = 31
    P = Namespace(augment_fn=aug, temp=0.1, lbd_a=1.0, distributed=False)
    gen = G(z)
    torch.manual_seed(seed); np.random.seed(seed)
    g_loss = ref_contrad.loss_G_fn(P, D, {'loss': 'nonsat'}, None, gen)
    G.zero_grad()
    g_loss.backward()
    ref_grads = {k: v.grad.clone() for k, v in G.named_parameters()}

    # oracle
    osd = {k: v.clone() for k, v in sd.items()}
    ogsd = {k: v.clone() for k, v in gsd.items()}
    gparams = [k for k in ogsd if not ('running' in k)]
    for k in gparams:
        ogsd[k].requires_grad_()
    ogen = O.sndcgan_g_forward(ogsd, z)
    torch.manual_seed(seed); np.random.seed(seed)
    p = O.sample_simclr_params(N, 32, 32, O.SIMCLR_CIFAR)
    oaug = O.simclr_apply(ogen, p)
    od = O.sndcgan_d_forward(osd, oaug, sg_linear=False)[0]
    ol = O.gan_g_loss(od, 'nonsat')
    ol.backward()
    check(ogen, gen, 1e-6, 'gstep gen')
    check(ol, g_loss, 1e-6, 'gstep loss')
    gerr = 0.0
    for k, gref in ref_grads.items():
        gerr = max(gerr, check(ogsd[k].grad, gref, 2e-5, 'gstep grad ' + k))
    print('  sndcgan G-step: max grad err vs reference %.2e, loss %.5f' % (gerr, g_loss.item()))
    out = {'z': z, 'seed': seed, 'N': N, 'gen': gen, 'g_loss': g_loss}
    for k, gref in ref_grads.items():
        out['gradnorm/' + k] = gref.norm()
        if gref.numel() <= 8192:
            out['grad/' + k] = gref
        else:
            out['gradhead/' + k] = gref.reshape(-1)[:512]
    save('sndcgan_gstep', **out)


# ------------------------------------------------------------------------------------------------
def gen_adam():
    g = torch.Generator().manual_seed(5)
    p = torch.randn(1000, generator=g)
    grads = [torch.randn(1000, generator=g) * (10.0 ** (-i)) for i in range(4)]
    for tag, (lr, b1, b2) in {'c10': (2e-4, 0.5, 0.999), 'sg2': (2e-3, 0.0, 0.99)}.items():
        pr = torch.nn.Parameter(p.clone())
        opt = torch.optim.Adam([pr], lr=lr, betas=(b1, b2))
        po = p.clone(); m = torch.zeros_like(po); v = torch.zeros_like(po)
        traj = []
        for t, gr in enumerate(grads, 1):
            pr.grad = gr.clone()
            opt.step()
            O.adam_step(po, gr, m, v, t, lr, b1, b2)
            check(po, pr.detach(), 1e-6, 'adam traj')
            traj.append(pr.detach().clone())
        save('adam_' + tag, p0=p, grads=torch.stack(grads), traj=torch.stack(traj), lr=lr, b1=b1, b2=b2)


# ------------------------------------------------------------------------------------------------
def gen_stylegan2():
    from oracle import stylegan2_oracle as S
    from models.gan import get_architecture
    from models.gan.stylegan2.op import upfirdn2d as ref_upfirdn2d
    from models.gan.stylegan2.op import fused_leaky_relu as ref_flr
    from training.gan import contrad as ref_contrad
    from argparse import Namespace
    out = {}

    # ---- upfirdn2d in the modes the hot path uses + a negative pad ----
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 9, 11, generator=g)
    k = S.make_kernel()
    for tag, (up, down, pad, kk) in {'blur22': (1, 1, (2, 2), k), 'blur11': (1, 1, (1, 1), k),
                                      'up2': (2, 1, (2, 1), k * 4), 'down2': (1, 2, (1, 1), k),
                                      'neg': (1, 1, (-1, 2), k), 'k2': (2, 1, (1, 0), k[:2, :2].contiguous() * 4)}.items():
        ref = ref_upfirdn2d(x, kk, up=up, down=down, pad=pad)
        check(S.upfirdn2d(x, kk, up, down, pad), ref, 1e-6, 'upfirdn2d ' + tag)
        out['ufd_%s_out' % tag] = ref
        out['ufd_%s_cfg' % tag] = torch.tensor([up, down, pad[0], pad[1]])
        out['ufd_%s_k' % tag] = kk
    out['ufd_x'] = x
    b = torch.randn(5, generator=g)
    check(S.fused_leaky_relu(x, b), ref_flr(x, b), 1e-7, 'fused_leaky_relu')
    out['flr_b'] = b
    out['flr_out'] = ref_flr(x, b)

    # ---- ResidualDiscriminatorP small32: state dict, ContraD D loss, R1 with double backward ----
    G, D = None, None
    from models.gan.stylegan2.discriminator import ResidualDiscriminatorP
    D = ResidualDiscriminatorP(size=32, small32=True, mlp_linear=True, d_hidden=512)
    D.train()
    shapes = S.d_param_shapes(32, True)
    ref_shapes = {kk: tuple(v.shape) for kk, v in D.state_dict().items()}
    assert ref_shapes == shapes, (set(ref_shapes) ^ set(shapes))
    assert list(ref_shapes) == list(shapes), 'state-dict ORDER'
    sd = S.det_fill_d(shapes, seed=2024)
    D.load_state_dict({kk: v.clone() for kk, v in sd.items()})
    N = 4
    g = torch.Generator().manual_seed(17)
    x = torch.rand(N, 3, 32, 32, generator=g)
    fake = torch.rand(N, 3, 32, 32, generator=g)
    aug = torch.rand(3 * N, 3, 32, 32, generator=g)       # stands for augment(cat[x,x,fake]) (explicit input)
    aug_r1 = torch.rand(N, 3, 32, 32, generator=g)        # stands for augment(x) inside r1_loss

    P = Namespace(augment_fn=lambda t: aug if t.size(0) == 3 * N else aug_r1, temp=0.1, lbd_a=1.0, distributed=False)
    D.zero_grad()
    d_loss, aux = ref_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, x, fake)
    # r1_loss of train_stylegan2.py:106-113 (the training scripts are not importable: imageio/torchvision)
    xa = aug_r1.detach().clone().requires_grad_()
    d_real = D(xa)
    grad_real, = torch.autograd.grad(outputs=d_real.sum(), inputs=xa, create_graph=True, retain_graph=True)
    r1 = grad_real.pow(2).reshape(N, -1).sum(1).mean()
    lbd_r1, d_reg_every = 0.1, 1                           # c10_style64 + --no_lazy (README.md:112-118)
    loss = d_loss + aux['penalty'] + (0.5 * lbd_r1) * r1 * d_reg_every
    loss.backward()
    ref_grads = {kk: v.grad.clone() for kk, v in D.named_parameters()}

    osd = {kk: v.clone() for kk, v in sd.items()}
    for kk in osd:
        if not kk.endswith('kernel'):
            osd[kk].requires_grad_()
    o_all, o_p, o_p2, o_f = S.d_forward(osd, aug, 32, sg_linear=True)
    closs, gloss, dr, dg = O.contrad_loss_d(lambda t: (o_all, o_p, o_p2), aug, N)
    or1 = S.r1_penalty(lambda t: S.d_forward(osd, t, 32)[0], aug_r1)
    (closs + gloss + 0.05 * or1).backward()
    check(closs, d_loss, 1e-6, 'sg2 contrad loss')
    check(gloss, aux['penalty'], 1e-6, 'sg2 gan loss')
    check(or1, r1, 1e-6, 'sg2 r1')
    gerr = 0.0
    for kk, gref in ref_grads.items():
        gerr = max(gerr, check(osd[kk].grad, gref, 2e-5, 'sg2 grad ' + kk))
    print('  stylegan2 D: max grad err vs reference %.2e, r1 = %.4f' % (gerr, r1.item()))
    with torch.no_grad():
        logit, auxo = D(aug, sg_linear=True, projection=True, projection2=True, penultimate=True)
        d_r1 = D(aug_r1)
    out.update({'x': x, 'fake': fake, 'aug': aug, 'aug_r1': aug_r1, 'N': N, 'logit': logit,
                'projection': auxo['projection'], 'projection2': auxo['projection2'],
                'penultimate_head': auxo['penultimate'][:, :64], 'penultimate_sum': auxo['penultimate'].sum(1),
                'contrad_loss': d_loss, 'gan_loss': aux['penalty'], 'r1': r1, 'd_r1_logits': d_r1,
                'grad_real_head': grad_real.detach().reshape(N, -1)[:, :256], 'grad_real_norm': grad_real.detach().norm()})
    for kk, gref in ref_grads.items():
        out['gradnorm/' + kk] = gref.norm()
        if gref.numel() <= 4096:
            out['grad/' + kk] = gref
        else:
            out['gradhead/' + kk] = gref.reshape(-1)[:512]
    save('stylegan2_d', **out)


# ------------------------------------------------------------------------------------------------
def gen_stylegan2_g():
    from oracle import stylegan2_oracle as S
    from models.gan.stylegan2.generator import Generator
    G = Generator(size=32, n_mlp=8, small32=True)
    G.train()
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    assert shapes == S.g_param_shapes(32, True)
    assert list(shapes) == list(S.g_param_shapes(32, True)), 'state-dict ORDER (the deterministic fill is keyed on it)' 
    sd = S.fill_kernels(S.det_fill_g(shapes, seed=777), shapes)
    G.load_state_dict({k: v.clone() for k, v in sd.items()})
    B = 3
    g = torch.Generator().manual_seed(41)
    z = torch.randn(B, 512, generator=g)
    noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=g) for i in range(G.num_layers)]
    with torch.no_grad():
        ref0 = G(z, style_mix=0.0, noise=noise)
        check(S.g_forward(sd, z, 32, noise), ref0, 1e-5, 'G forward (no mixing)')
        # style mixing: reproduce the reference's draws (CPU generator here: randn, rand, randint in that order)
        torch.manual_seed(9)
        ref1 = G(z, style_mix=0.9, noise=noise)
        torch.manual_seed(9)
        z_mix = torch.randn(B, 512)
        nomix = torch.rand(B) >= 0.9
        mix_layer = torch.randint(G.n_latent, (B,)).masked_fill(nomix, G.n_latent)
        check(S.g_forward(sd, z, 32, noise, mix=(z_mix, mix_layer)), ref1, 1e-5, 'G forward (mixing)')
        lat = G.get_latent(z)
        check(S.mapping(sd, z), lat, 1e-5, 'mapping')
    out = {'z': z, 'img_nomix': ref0, 'img_mix': ref1, 'z_mix': z_mix, 'mix_layer': mix_layer, 'latent': lat}
    for i, n in enumerate(noise):
        out['noise%d' % i] = n
    save('stylegan2_g', **out)


# ------------------------------------------------------------------------------------------------
def seeded_rand(shape, seed):
    """Inputs too large to commit are regenerated from a CPU-generator seed on both sides (torch's CPU mt19937
    stream is machine-independent); a float64 checksum travels with the fixture."""
    return torch.rand(*shape, generator=torch.Generator().manual_seed(seed))


def seeded_images(n, size, seed):
    """Distinct smooth synthetic 'photos' in [0,1]: a per-sample random 6x6 colour field, bilinearly upsampled, plus
    10 % pixel noise (i.i.d. uniform noise alone gives every image the same statistics -> collapsed embeddings, a
    parity test that could not see most bugs).  Same rule in tests/ (sg2_inputs.py)."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(n, 3, 6, 6, generator=g)
    img = F.interpolate(base, size=(size, size), mode='bilinear', align_corners=False)
    return (0.9 * img + 0.1 * torch.rand(n, 3, size, size, generator=g)).clamp_(0, 1)


def gen_stylegan2_512():
    """BASELINE config 5 at full resolution: ResidualDiscriminatorP(512, channel_multiplier=1) driven with the call
    sequence of train_stylegan2_contraD.py:129-164 (G_D.forward: fakes (N) and the two real views (2N) in SEPARATE
    D calls, _loss_D_fn :95-109, lazy R1 :129-136,222-226), N = 2; and Generator(512) forward with explicit noise."""
    from oracle import stylegan2_oracle as S
    from models.gan import get_architecture
    from training.criterion import nt_xent
    from training.gan.contrad import supcon_fake
    torch.manual_seed(0)
    G, D = get_architecture('stylegan2_512', (512, 512, 3))
    D.train(); G.train()
    shapes = S.d_param_shapes(512, False, 1.0)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == shapes
    assert list(D.state_dict()) == list(shapes)
    sd = S.det_fill_d(shapes, seed=512)
    D.load_state_dict({k: v.clone() for k, v in sd.items()})
    N, T, lbd_a, lbd_r1, every = 2, 0.1, 1.0, 0.5, 16
    aug_f = seeded_images(N, 512, 9001)         # stands for augment(G(z))
    aug_r = seeded_images(2 * N, 512, 9002)     # stands for augment(cat[x, x])
    aug_r1 = seeded_images(N, 512, 9003)        # stands for augment(x) inside the R1 branch

    def contrad_512(dfwd, r1fn):
        d_gen, pf, p2f = dfwd(aug_f)
        d_rs, pr, p2r = dfwd(aug_r)
        views_r, reals = F.normalize(pr), F.normalize(p2r)
        others, fakes = F.normalize(pf), F.normalize(p2f)
        simclr = nt_xent(views_r[:N], views_r[N:], temperature=T)
        sup = supcon_fake(reals[:N], reals[N:], fakes, temperature=T)
        d_real = d_rs[:N]
        gan = F.softplus(d_gen).mean() + F.softplus(-d_real).mean()
        r1 = r1fn()
        return simclr, sup, gan, r1, (d_gen, d_rs, pf, p2f, pr, p2r)

    def ref_fwd(t):
        o, a = D(t, sg_linear=True, projection=True, projection2=True)
        return o, a['projection'], a['projection2']

    keep = {}

    def ref_r1():
        xa = aug_r1.detach().clone().requires_grad_()
        d_real = D(xa)
        grad_real, = torch.autograd.grad(outputs=d_real.sum(), inputs=xa, create_graph=True, retain_graph=True)
        keep['d_r1'], keep['grad_real'] = d_real.detach(), grad_real.detach()
        return grad_real.pow(2).reshape(N, -1).sum(1).mean()

    D.zero_grad()
    simclr, sup, gan, r1, outs = contrad_512(ref_fwd, ref_r1)
    loss = simclr + lbd_a * sup + gan + (0.5 * lbd_r1) * r1 * every
    loss.backward()
    ref_grads = {k: v.grad.clone() for k, v in D.named_parameters()}
    print('  sg2-512 reference: simclr %.5f sup %.5f gan %.5f r1 %.5e' % (simclr.item(), sup.item(), gan.item(), r1.item()))

    osd = {k: v.clone() for k, v in sd.items()}
    for k in osd:
        if not k.endswith('kernel'):
            osd[k].requires_grad_()
    o_simclr, o_sup, o_gan, o_r1, _ = contrad_512(lambda t: S.d_forward(osd, t, 512, sg_linear=True)[:3],
                                                  lambda: S.r1_penalty(lambda t: S.d_forward(osd, t, 512)[0], aug_r1))
    (o_simclr + lbd_a * o_sup + o_gan + (0.5 * lbd_r1) * o_r1 * every).backward()
    check(o_simclr, simclr, 1e-6, '512 simclr'); check(o_sup, sup, 1e-6, '512 sup')
    check(o_gan, gan, 1e-6, '512 gan'); check(o_r1, r1, 1e-5, '512 r1')
    gerr = 0.0
    for k, gref in ref_grads.items():
        gerr = max(gerr, check(osd[k].grad, gref, 5e-5, '512 grad ' + k))
    print('  stylegan2-512 D: max grad err oracle vs reference %.2e' % gerr)
    d_gen, d_rs, pf, p2f, pr, p2r = [t.detach() for t in outs]
    out = {'N': N, 'seed_f': 9001, 'seed_r': 9002, 'seed_r1': 9003, 'wseed': 512,
           'sum_f': aug_f.double().sum(), 'sum_r': aug_r.double().sum(), 'sum_r1': aug_r1.double().sum(),
           'd_gen': d_gen, 'd_rs': d_rs, 'proj_f': pf, 'proj2_f': p2f, 'proj_r': pr, 'proj2_r': p2r,
           'simclr': simclr, 'sup': sup, 'gan': gan, 'r1': r1, 'd_r1_logits': keep['d_r1'],
           'grad_real_norm': keep['grad_real'].norm(), 'grad_real_head': keep['grad_real'].reshape(N, -1)[:, :256]}
    for k, gref in ref_grads.items():
        out['gradnorm/' + k] = gref.norm()
        out['gradhead/' + k] = gref.reshape(-1)[:256]
    save('stylegan2_512_d', **out)

    # ---- Generator(512, channel_multiplier=1) forward, explicit noise, with and without style mixing ----
    gshapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    assert gshapes == S.g_param_shapes(512, False, 1.0) and list(gshapes) == list(S.g_param_shapes(512, False, 1.0))
    gsd = S.fill_kernels(S.det_fill_g(gshapes, seed=778), gshapes)
    G.load_state_dict({k: v.clone() for k, v in gsd.items()})
    B = 2
    g = torch.Generator().manual_seed(43)
    z = torch.randn(B, 512, generator=g)
    nseed = 9100
    noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=torch.Generator().manual_seed(nseed + i))
             for i in range(G.num_layers)]
    with torch.no_grad():
        img0 = G(z, style_mix=0.0, noise=noise)
        check(S.g_forward(gsd, z, 512, noise), img0, 1e-5, 'G512 forward')
        # style mixing: reproduce the reference's CPU draws (randn, rand, randint in that order, generator.py:252-259)
        for mseed in range(100):           # a seed whose draws mix one sample and leave the other alone
            torch.manual_seed(mseed)
            z_mix = torch.randn(B, 512)
            nomix = torch.rand(B) >= 0.9
            mix_layer = torch.randint(G.n_latent, (B,)).masked_fill(nomix, G.n_latent)
            if nomix.sum().item() == 1:
                break
        torch.manual_seed(mseed)
        img1 = G(z, style_mix=0.9, noise=noise)
        check(S.g_forward(gsd, z, 512, noise, mix=(z_mix, mix_layer)), img1, 1e-5, 'G512 forward (mixing)')
    save('stylegan2_512_g', z=z, z_mix=z_mix, mix_layer=mix_layer, nseed=nseed, wseed=778,
         img0_sub=img0[:, :, ::32, ::32], img1_sub=img1[:, :, ::32, ::32],
         img0_rowsum=img0.double().sum(3), img1_rowsum=img1.double().sum(3), img0_patch=img0[:, :, 200:232, 300:332])


# ------------------------------------------------------------------------------------------------
def _r1_fixture(D, S, size, small32, cm, wseed, head_std, aug_r1, N):
    """The R1 penalty's gradient ALONE (train_stylegan2.py:106-113 / train_stylegan2_contraD.py:129-136 ->
    autograd.grad(r1, parameters)): in the full-step fixtures it is 0.5-1.4 % of the weight gradients and ~1e-6 of
    the bias gradients, so a 1e-3 check of the SUM cannot see an error in the double backward.  Here it is the whole
    signal, with the ``linear`` head scaled (``head_std``) so that r1 is O(0.1 ... 1).  (Inside one linear region of
    the leaky-relu network d D / d x does not depend on any bias; the bias gradients of r1 flow exclusively through the
    minibatch-stddev channel's sqrt(var + eps) -- its double backward is what they pin.)"""
    shapes = S.d_param_shapes(size, small32, cm)
    sd = S.det_fill_d(shapes, seed=wseed, head_std=head_std)
    D.load_state_dict({k: v.clone() for k, v in sd.items()})
    D.train()
    params = dict(D.named_parameters())
    xa = aug_r1.detach().clone().requires_grad_()
    d_real = D(xa)
    grad_real, = torch.autograd.grad(outputs=d_real.sum(), inputs=xa, create_graph=True, retain_graph=True)
    r1 = grad_real.pow(2).reshape(N, -1).sum(1).mean()
    gs = torch.autograd.grad(r1, list(params.values()), allow_unused=True)
    ref = {k: g for k, g in zip(params, gs)}

    osd = {k: v.clone() for k, v in sd.items()}
    for k in osd:
        if not k.endswith('kernel'):
            osd[k].requires_grad_()
    or1 = S.r1_penalty(lambda t: S.d_forward(osd, t, size)[0], aug_r1)
    names = [k for k in osd if not k.endswith('kernel')]
    ogs = torch.autograd.grad(or1, [osd[k] for k in names], allow_unused=True)
    check(or1, r1, 1e-5, 'r1 (%d)' % size)
    unused = []
    worst = 0.0
    for k, og in zip(names, ogs):
        if ref[k] is None or og is None:
            # (unused by r1 on one side, an exactly-zero gradient on the other: same thing)
            assert (og is None or og.abs().max().item() == 0) and (ref[k] is None or ref[k].abs().max().item() == 0), k
            ref[k] = None
            unused.append(k)
            continue
        e = ((og - ref[k]).norm() / ref[k].norm().clamp_min(1e-30)).item()
        worst = max(worst, e)
        assert e < 2e-5, ('r1 grad ' + k, e)
    print('  r1-only gradient (%d^2, N=%d): r1 = %.4f, |grad_real| = %.4f, oracle vs reference worst rel-L2 %.2e, '
          'no gradient: %s' % (size, N, r1.item(), grad_real.norm().item(), worst, unused))
    out = {'N': N, 'wseed': wseed, 'head_std': head_std, 'r1': r1, 'd_r1_logits': d_real.detach(),
           'grad_real_norm': grad_real.detach().norm(),
           'grad_real_head': grad_real.detach().reshape(N, -1)[:, :256],
           'grad_real_rowsum': grad_real.detach().double().sum(3)[:, :, ::max(1, size // 32)]}
    for k, g in ref.items():
        if g is None or g.abs().max().item() == 0:       # e.g. linear.l1.bias: d D / d x does not depend on it
            out['r1none/' + k] = 1
            continue
        out['r1gradnorm/' + k] = g.norm()
        if g.numel() <= 4096:
            out['r1grad/' + k] = g
        else:
            out['r1gradhead/' + k] = g.reshape(-1)[:512]
            # a strided sample as well: the first 512 entries of an OIHW weight all belong to output channel 0
            out['r1gradstride/' + k] = g.reshape(-1)[::max(1, g.numel() // 512)][:512]
    return out


def gen_stylegan2_r1():
    from oracle import stylegan2_oracle as S
    from models.gan.stylegan2.discriminator import ResidualDiscriminatorP
    # ---- 32^2 (BASELINE config 4: R1 every step) ----
    D = ResidualDiscriminatorP(size=32, small32=True, mlp_linear=True, d_hidden=512)
    N = 4
    aug_r1 = torch.rand(N, 3, 32, 32, generator=torch.Generator().manual_seed(171))
    out = _r1_fixture(D, S, 32, True, 2, 2025, 0.1, aug_r1, N)
    out['aug_r1'] = aug_r1
    save('stylegan2_r1', **out)
    # ---- 512^2 (BASELINE config 5: lazy R1), inputs regenerated from the seed on both sides ----
    from models.gan import get_architecture
    torch.manual_seed(0)
    _G, D = get_architecture('stylegan2_512', (512, 512, 3))
    N = 4        # one whole minibatch-stddev group (discriminator.py:22-33: groups of four, as in a training batch of 16): with N = 2
                 # the conv-bias gradients -- which reach r1 through sqrt(var + 1e-8) of the group alone -- are fp32-ill-conditioned
                 # (the reference's own fp32 result is 0.6e-3 ... 1.4e-3 from its float64 evaluation; at N = 4: 0.6e-4 ... 2.4e-4,
                 # tools/dev/r1_conditioning.py)
    aug_r1 = seeded_images(N, 512, 9004)
    out = _r1_fixture(D, S, 512, False, 1.0, 513, 0.3, aug_r1, N)
    out['seed_r1'] = 9004
    out['sum_r1'] = aug_r1.double().sum()
    save('stylegan2_512_r1', **out)


# ------------------------------------------------------------------------------------------------
def gen_stylegan2_gstep():
    """StyleGAN2 generator step (train_stylegan2.py:184-194; train_stylegan2_contraD.py:138-146 computes the same
    d_gen): G(z, style_mix) with grad, explicit noise -> loss_G_fn = softplus(-D(augment(G(z)))).mean() -> all G
    gradients (mapping network, modulated convs incl. demodulation, noise strengths, ToRGB + upsampled skips, const)."""
    import augment as A
    from oracle import stylegan2_oracle as S
    from models.gan.stylegan2.generator import Generator
    from models.gan.stylegan2.discriminator import ResidualDiscriminatorP
    from training.gan import contrad as ref_contrad
    from argparse import Namespace
    _refshim.bind_cifar_defaults()
    G = Generator(size=32, n_mlp=8, small32=True)
    D = ResidualDiscriminatorP(size=32, small32=True, mlp_linear=True, d_hidden=512)
    G.train(); D.train()
    gshapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    gsd = S.fill_kernels(S.det_fill_g(gshapes, seed=779), gshapes)
    # the reference initialises the noise strengths to 0 (their gradient is still non-zero); give them a value so the
    # forward depends on the injected noise as it does after training
    for k in gsd:
        if k.endswith('noise.weight'):
            gsd[k] = torch.full_like(gsd[k], 0.1)
    G.load_state_dict({k: v.clone() for k, v in gsd.items()})
    dsd = S.det_fill_d(S.d_param_shapes(32, True), seed=2025)
    D.load_state_dict({k: v.clone() for k, v in dsd.items()})
    for p in D.parameters():
        p.requires_grad = False
    B = 4
    g = torch.Generator().manual_seed(47)
    z = torch.randn(B, 512, generator=g)
    noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), generator=g) for i in range(G.num_layers)]
    for mseed in range(100):           # a seed whose draws mix some samples and leave others alone
        torch.manual_seed(mseed)
        z_mix = torch.randn(B, 512)
        nomix = torch.rand(B) >= 0.9
        mix_layer = torch.randint(G.n_latent, (B,)).masked_fill(nomix, G.n_latent)
        if 1 <= nomix.sum().item() < B:
            break
    torch.manual_seed(mseed)
    gen = G(z, style_mix=0.9, noise=noise)
    aug = A.simclr()
    seed = 33
    P = Namespace(augment_fn=aug, temp=0.1, lbd_a=1.0, distributed=False)
    torch.manual_seed(seed); np.random.seed(seed)
    g_loss = ref_contrad.loss_G_fn(P, D, {'loss': 'nonsat'}, None, gen)
    G.zero_grad()
    g_loss.backward()
    ref_grads = {k: v.grad.clone() for k, v in G.named_parameters()}

    osd = {k: v.clone() for k, v in gsd.items()}
    gparams = [k for k, _ in G.named_parameters()]
    for k in gparams:
        osd[k].requires_grad_()
    ogen = S.g_forward(osd, z, 32, noise, mix=(z_mix, mix_layer))
    check(ogen, gen, 1e-5, 'sg2 gstep gen')
    torch.manual_seed(seed); np.random.seed(seed)
    p = O.sample_simclr_params(B, 32, 32, O.SIMCLR_CIFAR)
    od = S.d_forward(dsd, O.simclr_apply(ogen, p), 32, sg_linear=False)[0]
    ol = O.gan_g_loss(od, 'nonsat')
    ol.backward()
    check(ol, g_loss, 1e-6, 'sg2 gstep loss')
    gerr = 0.0
    for k, gref in ref_grads.items():
        gerr = max(gerr, check(osd[k].grad, gref, 5e-5, 'sg2 gstep grad ' + k))
    print('  stylegan2 G-step: max grad err oracle vs reference %.2e, loss %.5f' % (gerr, g_loss.item()))
    out = {'z': z, 'z_mix': z_mix, 'mix_layer': mix_layer, 'seed': seed, 'B': B, 'gen': gen, 'g_loss': g_loss,
           'gseed': 779, 'dseed': 2025}
    for i, n in enumerate(noise):
        out['noise%d' % i] = n
    for k, gref in ref_grads.items():
        out['gradnorm/' + k] = gref.norm()
        if gref.numel() <= 2048:
            out['grad/' + k] = gref
        else:
            out['gradhead/' + k] = gref.reshape(-1)[:256]
    save('stylegan2_gstep', **out)


# ------------------------------------------------------------------------------------------------
def gen_checkpoint_manifest():
    """Structure of the checkpoint files the reference writes (train_gan.py:211-225, train_stylegan2.py:260-285):
    state-dict key ORDER / shapes / dtypes of gen.pt, dis.pt (and gen_ema.pt), and the layout of optim.pt as produced by
    ``torch.optim.Adam(...).state_dict()`` on the reference modules after one step.  The tensors themselves (74 MB for
    D_SNDCGAN) are not committed: tests rebuild files of exactly this structure with seeded values and load them
    through --resume / --finetune on the GPU box."""
    import json
    from models.gan import get_architecture
    out = {}
    for arch, size in (('sndcgan', 32), ('stylegan2', 32)):
        G, D = get_architecture(arch, (size, size, 3))
        ent = {}
        for tag, m in (('gen', G), ('dis', D)):
            ent[tag] = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
            opt = torch.optim.Adam(m.parameters(), lr=2e-4, betas=(0.5, 0.999))
            for prm in m.parameters():
                prm.grad = torch.zeros_like(prm)
            opt.step()
            sdo = opt.state_dict()
            groups = [{k: (list(v) if isinstance(v, tuple) else v) for k, v in g.items()} for g in sdo['param_groups']]
            st0 = sdo['state'][0]
            ent['optim_' + tag] = {
                'param_groups': groups,
                'n_state': len(sdo['state']),
                'state_keys': list(st0.keys()),
                'step': [str(type(st0['step']).__name__), str(getattr(st0['step'], 'dtype', '')),
                         list(getattr(st0['step'], 'shape', []))],
                'param_names': [k for k, _ in m.named_parameters()],
            }
        out[arch] = ent
    out['optim_pt_keys'] = ['epoch', 'optim_G', 'optim_D']
    out['torch'] = torch.__version__
    path = os.path.join(HERE, 'checkpoint_manifest.json')
    json.dump(out, open(path, 'w'), indent=0)
    print('wrote checkpoint_manifest.json %7.1f KiB' % (os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------
def gen_snresnet():
    """Scope row N4: D_SNResNet18 (models/gan/snresnet.py, built by get_architecture('snresnet18')) under the ContraD
    discriminator loss, and the simclr_only mode (training/gan/simclr_only.py) on the same network."""
    from models.gan import get_architecture
    from training.gan import contrad as ref_contrad
    from training.gan import simclr_only as ref_so
    from argparse import Namespace
    G, D = get_architecture('snresnet18', (32, 32, 3))
    D.train()
    shapes = O.snresnet18_param_shapes()
    ref_shapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    assert ref_shapes == shapes, set(ref_shapes) ^ set(shapes)
    assert list(ref_shapes) == list(shapes), 'snresnet state-dict ORDER'
    sd = O.det_fill(shapes, seed=1818, weight_std=0.05)
    N = 4
    g = torch.Generator().manual_seed(181)
    x = torch.rand(N, 3, 32, 32, generator=g)
    fake = torch.rand(N, 3, 32, 32, generator=g)
    aug = torch.rand(3 * N, 3, 32, 32, generator=g)
    out = {'x': x, 'fake': fake, 'aug': aug, 'N': N, 'wseed': 1818}
    for mode in ('contrad', 'simclr_only'):
        _load_sd(D, sd)
        P = Namespace(augment_fn=lambda t: aug[:t.size(0)], temp=0.1, lbd_a=1.0, distributed=False)
        D.zero_grad()
        fn = ref_contrad.loss_D_fn if mode == 'contrad' else ref_so.loss_D_fn
        d_loss, aux = fn(P, D, {'loss': 'nonsat'}, x, fake)
        (d_loss + aux['penalty']).backward()
        ref_grads = {k: (v.grad.clone() if v.grad is not None else torch.zeros_like(v)) for k, v in D.named_parameters()}
        ref_after = {k: v.clone() for k, v in D.state_dict().items()}
        osd = {k: v.clone() for k, v in sd.items()}
        for k in osd:
            if k.endswith('weight_orig') or k.endswith('bias'):
                osd[k].requires_grad_()
        if mode == 'contrad':
            closs, gloss, _, _ = O.contrad_loss_d(lambda t: O.snresnet18_forward(osd, t, sg_linear=True)[:3], aug, N)
            (closs + gloss).backward()
            check(closs, d_loss, 1e-6, 'snresnet contrad'); check(gloss, aux['penalty'], 1e-6, 'snresnet gan')
            out.update({'contrad_loss': d_loss, 'gan_loss': aux['penalty']})
        else:
            l = O.simclr_only_loss_d(lambda t: O.snresnet18_forward(osd, t)[:2], aug[:2 * N], N)
            l.backward()
            check(l, d_loss, 1e-6, 'simclr_only loss')
            assert aux['penalty'].item() == 0.0
            out['simclr_only_loss'] = d_loss
        gerr = 0.0
        for k, gref in ref_grads.items():
            og = osd[k].grad if osd[k].grad is not None else torch.zeros_like(gref)
            gerr = max(gerr, check(og, gref, 2e-5, 'snresnet %s grad %s' % (mode, k)))
        for k in osd:
            if k.endswith('weight_u'):
                check(osd[k], ref_after[k], 1e-6, 'snresnet sn buffer ' + k)
        print('  snresnet18 %s: max grad err oracle vs reference %.2e' % (mode, gerr))
        for k, gref in ref_grads.items():
            out['%s/gradnorm/%s' % (mode, k)] = gref.norm()
            if gref.numel() <= 2048:
                out['%s/grad/%s' % (mode, k)] = gref
        if mode == 'contrad':
            for k in ('conv1.weight_u', 'layer4.1.conv2.weight_u', 'linear.l1.weight_u'):
                out['after/' + k] = ref_after[k]
    _load_sd(D, sd)
    with torch.no_grad():
        logit, auxo = D(aug, sg_linear=True, projection=True, projection2=True, penultimate=True)
    out.update({'logit': logit, 'projection': auxo['projection'], 'projection2': auxo['projection2'],
                'penultimate': auxo['penultimate']})
    save('snresnet', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['losses', 'augment', 'sndcgan', 'sndcgan_eval', 'sndcgan_gstep', 'adam', 'stylegan2', 'stylegan2_g', 'stylegan2_512', 'stylegan2_r1', 'stylegan2_gstep', 'checkpoint_manifest', 'snresnet']
    for w in which:
        globals()['gen_' + w]()
    print('golden vectors OK')
