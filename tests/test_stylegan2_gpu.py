"""GPU parity of the StyleGAN2 rows (D4, D5, R1): the two native ops' HIP counterparts and the residual
discriminator incl. the R1 double backward, against the reference-generated goldens and the oracle."""
import math

import numpy as np
import pytest
import torch

from contrad_amd import ops
from contrad_amd.models.gan.stylegan2.discriminator import ResidualDiscriminatorP
from contrad_amd.training.gan import contrad as hip_contrad
from oracle import contrad_oracle as O
from oracle import stylegan2_oracle as S

pytestmark = pytest.mark.gpu
TOL = 1e-3
FLIP_TOL = float(__import__('os').environ.get('CONTRAD_FLIP_TOL', '1e-3'))      # leaky-relu sign flips vs the raw reference goldens: see tests/test_sndcgan_gpu.py
DEV = 'cuda'


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def l2(a, b):
    a, b = torch.as_tensor(a).double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('tag', ['blur22', 'blur11', 'up2', 'down2', 'neg', 'k2'])
def test_upfirdn2d_against_reference(golden, tag):
    g = golden('stylegan2_d')
    x = torch.from_numpy(g['ufd_x'])
    up, down, p0, p1 = [int(v) for v in g['ufd_%s_cfg' % tag]]
    k = torch.from_numpy(g['ufd_%s_k' % tag]).contiguous()
    out = ops.upfirdn2d(x.permute(0, 2, 3, 1).contiguous().to(DEV), k.to(DEV), up, down, (p0, p1, p0, p1))
    assert rel(out.permute(0, 3, 1, 2), g['ufd_%s_out' % tag]) < 1e-5


def test_upfirdn2d_adjoint_and_double_backward():
    """<blur(x), g> == <x, blur_backward(g)> and the backward of the backward is the forward (op/upfirdn2d.py:19-85)."""
    from contrad_amd import autograd_ops as A
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(3, 9, 9, 8, device=DEV, generator=g, requires_grad=True)
    k = A.make_blur_kernel().to(DEV)
    for pad in ((2, 2, 2, 2), (1, 1, 1, 1)):
        y = A.UpFirDn2dFn.apply(x, k, 1, 1, pad)
        gy = torch.randn(y.shape, device=DEV, generator=g, requires_grad=True)
        gx, = torch.autograd.grad(y, x, gy, create_graph=True)
        assert abs((y * gy).sum().item() - (x * gx).sum().item()) < 1e-3 * abs((y * gy).sum().item())
        h = torch.randn(gx.shape, device=DEV, generator=g)
        ggy, = torch.autograd.grad(gx, gy, h)
        assert rel(ggy, A.UpFirDn2dFn.apply(h, k, 1, 1, pad).detach()) < 1e-5


def test_fused_bias_act(golden):
    g = golden('stylegan2_d')
    x = torch.from_numpy(g['ufd_x']).permute(0, 2, 3, 1).contiguous().to(DEV)
    b = torch.from_numpy(g['flr_b']).to(DEV)
    y = ops.fused_bias_act(x, b, None, 3, 0, 0.2, math.sqrt(2))
    assert rel(y.permute(0, 3, 1, 2), g['flr_out']) < 1e-6
    gy = torch.randn_like(y)
    gx = ops.fused_bias_act(gy, None, y, 3, 1, 0.2, math.sqrt(2))
    ref = gy * torch.where(y > 0, math.sqrt(2), 0.2 * math.sqrt(2))
    assert rel(gx, ref) < 1e-6


@pytest.mark.parametrize('B,splits', [(8, None), (12, (4, 8)), (3, None), (16, (16,))])
def test_minibatch_stddev_node_family_against_torch_ops(B, splits):
    """The HIP minibatch-stddev channel (forward, backward, and the backward's backward that R1 needs) against the same layer
    written in differentiable torch ops (the round-3 implementation, discriminator.py:22-33): values, first-order input
    gradient, and both second-order gradients (w.r.t. x and w.r.t. the incoming gradient), 1e-4."""
    from contrad_amd.models.gan.stylegan2.discriminator import minibatch_stddev_batches, minibatch_stddev_nhwc
    g = torch.Generator(device='cuda').manual_seed(B)
    x0 = torch.randn(B, 4, 4, 32, device=DEV, generator=g)

    def ref(x):
        if splits is None or len(splits) == 1:
            return minibatch_stddev_nhwc(x)
        return torch.cat([minibatch_stddev_nhwc(t) for t in torch.split(x, list(splits), dim=0)], dim=0)

    outs = []
    for fn in (lambda t: minibatch_stddev_batches(t, splits), ref):
        x = x0.clone().requires_grad_()
        y = fn(x)
        gy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device='cuda').manual_seed(1)).requires_grad_()
        gx, = torch.autograd.grad(y, x, gy, create_graph=True)
        h = torch.randn(gx.shape, device=DEV, generator=torch.Generator(device='cuda').manual_seed(2))
        gx2, ggy = torch.autograd.grad(gx, (x, gy), h)
        outs.append((y.detach(), gx.detach(), gx2, ggy))
    assert outs[0][0].shape == outs[1][0].shape == (B, 4, 4, 48)
    for a, b, what in zip(outs[0], outs[1], ('y', 'gx', 'gx2', 'ggy')):
        assert rel(a, b) < 1e-4, (what, rel(a, b))


def test_r1_sum_of_squares_node():
    from contrad_amd import autograd_ops as A
    g = torch.randn(6, 3, 32, 32, device=DEV, requires_grad=True)
    r = A.SumSqMeanFn.apply(g)
    ref = g.detach().pow(2).reshape(6, -1).sum(1).mean()
    assert abs(r.item() - ref.item()) < 1e-5 * ref.item()
    (3.0 * r).backward()
    assert rel(g.grad, 3.0 * 2.0 / 6 * g.detach()) < 1e-6


def build_d():
    D = ResidualDiscriminatorP(32, small32=True, mlp_linear=True, d_hidden=512)
    D.load_state_dict(S.det_fill_d(S.d_param_shapes(32, True), seed=2024))
    return D.to(DEV).train()


def test_state_dict_contract():
    D = ResidualDiscriminatorP(32, small32=True)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == S.d_param_shapes(32, True)
    D = ResidualDiscriminatorP(512, channel_multiplier=1.0)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == S.d_param_shapes(512, False, 1.0)


class _P(object):
    def __init__(self, aug, aug_r1, N):
        self.augment_fn = lambda t: aug if t.size(0) == 3 * N else aug_r1
        self.temp, self.lbd_a, self.distributed = 0.1, 1.0, False


def hip_step(D, g, N):
    aug = torch.from_numpy(g['aug']).to(DEV)
    aug_r1 = torch.from_numpy(g['aug_r1']).to(DEV)
    P = _P(aug, aug_r1, N)
    d_loss, a = hip_contrad.loss_D_fn(P, D, {'loss': 'nonsat'}, torch.from_numpy(g['x']).to(DEV),
                                      torch.from_numpy(g['fake']).to(DEV))
    xa = aug_r1.detach().clone().requires_grad_()
    d_real = D(xa)
    grad_real, = torch.autograd.grad(outputs=d_real.sum(), inputs=xa, create_graph=True, retain_graph=True)
    r1 = grad_real.pow(2).reshape(N, -1).sum(1).mean()
    loss = d_loss + a['penalty'] + (0.5 * 0.1) * r1 * 1
    D.zero_grad()
    loss.backward()
    return d_loss, a, r1, grad_real, d_real


def test_discriminator_forward_and_r1_step_against_reference(golden, margin):
    g = golden('stylegan2_d')
    N = int(g['N'])
    D = build_d()
    aug = torch.from_numpy(g['aug']).to(DEV)
    with torch.no_grad():
        logit, aux = D(aug, sg_linear=True, projection=True, projection2=True, penultimate=True)
    assert rel(logit, g['logit']) < TOL
    assert rel(aux['projection'], g['projection']) < TOL and rel(aux['projection2'], g['projection2']) < TOL
    assert rel(aux['penultimate'][:, :64], g['penultimate_head']) < TOL
    assert rel(aux['penultimate'].sum(1), g['penultimate_sum']) < TOL

    d_loss, a, r1, grad_real, d_real = hip_step(D, g, N)
    assert abs(d_loss.item() - float(g['contrad_loss'])) < TOL * abs(float(g['contrad_loss']))
    assert abs(a['penalty'].item() - float(g['gan_loss'])) < TOL * abs(float(g['gan_loss']))
    assert rel(d_real, g['d_r1_logits']) < TOL
    margin('sg2_32 golden/r1', abs(r1.item() - float(g['r1'])) / float(g['r1']), TOL)
    margin('sg2_32 golden/grad_real_norm', abs(grad_real.norm().item() - float(g['grad_real_norm'])) / float(g['grad_real_norm']), TOL)
    grads = {k: p.grad for k, p in D.named_parameters()}
    for k in g.files:
        if k.startswith('gradnorm/'):
            name = k[len('gradnorm/'):]
            e = abs(grads[name].norm().item() - float(g[k])) / max(float(g[k]), 1e-30)
            margin('sg2_32 golden/gradnorm/' + name, e, TOL)
        elif k.startswith('grad/'):
            margin('sg2_32 golden/grad-l2/' + k[5:], l2(grads[k[5:]], g[k]), FLIP_TOL)


def test_r1_step_on_the_same_linear_region(golden):
    """Strict element-wise check of first AND second order gradients: oracle evaluated with the leaky-relu
    sign patterns recorded from the two HIP forwards (D-step batch and R1 batch)."""
    g = golden('stylegan2_d')
    N = int(g['N'])
    D = build_d()
    D._record_activations = True
    d_loss, a, r1, grad_real, d_real = hip_step(D, g, N)
    (rec_a, hl_a, hpq_a), (rec_b, hl_b, hpq_b) = D._recorded[0], D._recorded[1]

    def masks_of(rec):
        return [(t > 0).permute(0, 3, 1, 2).cpu() for t in rec]

    def head_masks(hl, hpq):
        hl, hpq = hl.reshape(hl.shape[0], -1).cpu(), hpq.reshape(hpq.shape[0], -1).cpu()
        return (hl > 0, hpq[:, :512] > 0, hpq[:, 512:] > 0)

    osd = S.det_fill_d(S.d_param_shapes(32, True), seed=2024)
    for k in osd:
        if not k.endswith('kernel'):
            osd[k].requires_grad_()
    aug, aug_r1 = torch.from_numpy(g['aug']), torch.from_numpy(g['aug_r1'])
    o_all, o_p, o_p2, _ = S.d_forward(osd, aug, 32, sg_linear=True, masks=masks_of(rec_a),
                                      head_masks=head_masks(hl_a, hpq_a))
    closs, gloss, _, _ = O.contrad_loss_d(lambda t: (o_all, o_p, o_p2), aug, N)
    or1 = S.r1_penalty(lambda t: S.d_forward(osd, t, 32, masks=masks_of(rec_b),
                                             head_masks=head_masks(hl_b, hpq_b))[0], aug_r1)
    (closs + gloss + 0.05 * or1).backward()
    assert abs(d_loss.item() - closs.item()) < TOL * abs(closs.item())
    assert abs(a['penalty'].item() - gloss.item()) < TOL * abs(gloss.item())
    assert abs(r1.item() - or1.item()) < TOL * or1.item()
    for k, prm in D.named_parameters():
        ref = osd[k].grad
        e = (prm.grad.cpu() - ref).abs().max().item() / ref.abs().max().clamp_min(1e-30).item()
        assert e < TOL, (k, e)


def test_generator_forward_against_reference(golden):
    """StyleGAN2 generator (row G0): mapping network, modulated convs (input-modulate / output-demodulate form),
    noise + activation epilogue, ToRGB with upsampled skip; explicit noise, with and without style mixing."""
    from contrad_amd.models.gan.stylegan2.generator import Generator
    g = golden('stylegan2_g')
    G = Generator(size=32, n_mlp=8, small32=True)
    shapes = S.g_param_shapes(32, True)
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == shapes
    G.load_state_dict(S.fill_kernels(S.det_fill_g(shapes, seed=777), shapes))
    G = G.to(DEV).train()
    z = torch.from_numpy(g['z']).to(DEV)
    noise = [torch.from_numpy(g['noise%d' % i]).to(DEV) for i in range(G.num_layers)]
    with torch.no_grad():
        lat = G._mapping(z, G._prepared())
        assert rel(lat, g['latent']) < TOL
        img0 = G(z, style_mix=0.0, noise=noise)
        assert rel(img0, g['img_nomix']) < TOL
        img1 = G(z, style_mix=0.9, noise=noise, _mix=(torch.from_numpy(g['z_mix']).to(DEV),
                                                      torch.from_numpy(g['mix_layer'])))
        assert rel(img1, g['img_mix']) < TOL
        # sampling path runs (device RNG for the second latent / noise, CPU RNG for the masks)
        out = G(G.sample_latent(5))
        assert out.shape == (5, 3, 32, 32) and torch.isfinite(out).all()


@pytest.mark.parametrize('size,small32,n', [(32, True, 5), (64, False, 3)])
def test_generator_fused_tail_equals_the_separate_passes(size, small32, n):
    """The forward-only generator with the upsampling StyledConv's blur + demod / noise / bias / lrelu epilogue in one launch
    and the next layer's weight modulation folded into the producer's store (generator.FUSE_TAIL) is BITWISE the sequence
    blur -> modconv_epilogue -> nhwc_scale it replaces (same operations in the same order per element)."""
    import contrad_amd.models.gan.stylegan2.generator as gen
    G = gen.Generator(size=size, n_mlp=8, small32=small32).to(DEV).train()
    torch.manual_seed(3)
    z = torch.randn(n, G.style_dim, device=DEV)
    noise = [torch.randn(n, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=DEV) for i in range(G.num_layers)]
    mix = (torch.randn(n, G.style_dim, device=DEV), torch.randint(G.n_latent, (n,)))
    saved = gen.FUSE_TAIL
    try:
        outs = []
        for flag in (True, False):
            gen.FUSE_TAIL = flag
            with torch.no_grad():
                outs.append(G(z, style_mix=0.9, noise=noise, _mix=mix).clone())
    finally:
        gen.FUSE_TAIL = saved
    assert torch.isfinite(outs[0]).all() and outs[0].abs().max() > 0
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('shape', [(2, 9, 9, 8), (3, 17, 33, 32), (1, 65, 65, 64)])
@pytest.mark.parametrize('with_opt', [True, False])
def test_upfirdn2d_modconv_against_plain_torch(shape, with_opt):
    """contrad_upfirdn2d_modconv (blur pad (1,1) + demod + noise + bias + lrelu * sqrt2 [* post]) against torch ops."""
    from contrad_amd import ops
    import torch.nn.functional as F
    B, H, W, C = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, C, generator=g)
    k1 = torch.tensor([1., 3., 3., 1.]); k = (k1[:, None] * k1[None, :]); k = k / k.sum() * 4.0
    demod = torch.rand(B, C, generator=g) + 0.5 if with_opt else None
    post = torch.randn(B, C, generator=g) if with_opt else None
    nw = torch.tensor([0.3])
    bias = torch.randn(C, generator=g)
    oh, ow = H - 1, W - 1
    noise = torch.randn(B, 1, oh, ow, generator=g) if with_opt else None
    dv = lambda t: None if t is None else t.to(DEV)
    got = ops.upfirdn2d_modconv(dv(x), dv(k), (1, 1, 1, 1), dv(demod), dv(noise), dv(nw), dv(bias), dv(post)).cpu()
    ref = _upfirdn_ref(x, k, 1, 1, (1, 1, 1, 1))
    if demod is not None:
        ref = ref * demod[:, None, None, :]
    if noise is not None:
        ref = ref + nw * noise.view(B, oh, ow, 1)
    ref = F.leaky_relu(ref + bias, 0.2) * math.sqrt(2.0)
    if post is not None:
        ref = ref * post[:, None, None, :]
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


def test_modconv_tables_against_torch():
    """contrad_modconv_tables: packed shared weights in both orientations + the demodulation tables of several layers
    (ragged channel counts, 1x1 and 3x3, a 3-channel ToRGB) from one launch, against torch ops on the same weights."""
    g = torch.Generator().manual_seed(8)
    cfgs = [(64, 32, 3, False, True), (32, 64, 3, True, True), (3, 48, 1, True, False), (40, 36, 3, False, True),
            (512, 512, 3, True, True), (128, 256, 1, False, False)]          # (Cout, Cin, k, transposed, demodulate)
    layers, refs = [], []
    for co, ci, k, tr, dm in cfgs:
        w = torch.randn(co, ci, k, k, generator=g)
        scale = 1.0 / math.sqrt(ci * k * k)
        rows, cols = (k * k * co, ci) if tr else (k * k * ci, co)
        wp = torch.full((rows, cols), float('nan'), device=DEV)
        wsq = torch.full((ci, co), float('nan'), device=DEV) if dm else None
        layers.append((w.to(DEV), wp, wsq, tr, scale))
        ws = w * scale
        if tr:       # rows (tap, cout), cols cin
            ref_wp = ws.permute(2, 3, 0, 1).reshape(k * k * co, ci)
        else:        # rows (tap, cin), cols cout
            ref_wp = ws.permute(2, 3, 1, 0).reshape(k * k * ci, co)
        refs.append((ref_wp, ws.pow(2).sum((2, 3)).t() if dm else None))
    ops.modconv_tables(layers)
    for (w, wp, wsq, tr, scale), (ref_wp, ref_wsq) in zip(layers, refs):
        assert torch.equal(wp.cpu(), ref_wp)                      # one multiply per element: exact
        if wsq is not None:
            assert rel(wsq, ref_wsq) < 1e-6


@pytest.mark.parametrize('B', [5, 16, 70])
def test_modconv_demod_against_torch(B):
    """contrad_modconv_demod: rsqrt(style^2 @ wsq + 1e-8) of several layers (ragged channel counts, Cin above one LDS
    chunk) from one launch."""
    g = torch.Generator().manual_seed(9)
    jobs, refs = [], []
    for cin, k in [(512, 512), (32, 64), (100, 36), (1024, 300), (64, 3)]:
        st = torch.randn(B, cin, generator=g)
        wsq = torch.rand(cin, k, generator=g) / cin
        out = torch.full((B, k), float('nan'), device=DEV)
        jobs.append((st.to(DEV), wsq.to(DEV), out))
        refs.append(torch.rsqrt((st.double() ** 2) @ wsq.double() + 1e-8).float())
    ops.modconv_demod(jobs, 1e-8)
    for (st, wsq, out), ref in zip(jobs, refs):
        assert rel(out, ref) < 1e-5


def _upfirdn_ref(x, k, up, down, pad):
    """Plain-torch upfirdn2d on NHWC (B,H,W,C) with pad = (x0, x1, y0, y1): zero insertion, (possibly negative) padding,
    correlation with the flipped kernel, decimation -- the contract of op/upfirdn2d_kernel.cu:209-243."""
    import torch.nn.functional as F
    B, H, W, C = x.shape
    z = x.new_zeros(B, H * up, W * up, C)
    z[:, ::up, ::up] = x
    px0, px1, py0, py1 = pad
    z = F.pad(z, (0, 0, max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)))
    z = z[:, max(-py0, 0): z.shape[1] - max(-py1, 0), max(-px0, 0): z.shape[2] - max(-px1, 0)]
    zz = z.permute(0, 3, 1, 2).reshape(B * C, 1, z.shape[1], z.shape[2])
    out = F.conv2d(zz, torch.flip(k, [0, 1]).view(1, 1, *k.shape))
    out = out[:, :, ::down, ::down]
    return out.reshape(B, C, out.shape[2], out.shape[3]).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('cfg', [(1, 1, (2, 2, 2, 2)), (1, 1, (1, 1, 1, 1)), (1, 1, (1, 2, 2, 1)), (1, 1, (-1, 2, 0, 1)),
                                 (1, 2, (1, 1, 1, 1)), (1, 2, (2, 2, 2, 2)), (1, 2, (0, 1, 1, 0)),
                                 (2, 1, (2, 1, 2, 1)), (2, 1, (1, 2, 1, 2)), (2, 1, (0, 3, 3, 0)), (2, 1, (2, 2, 1, 1))])
@pytest.mark.parametrize('shape', [(2, 9, 11, 4), (3, 16, 16, 32), (1, 33, 7, 8), (2, 4, 5, 64), (6, 9, 11, 1), (3, 16, 17, 1)])
def test_fir4_specialisations_against_plain_torch(cfg, shape):
    """The 4x4-FIR kernels (up1/down1 tile 4x2, down 2 tile 2x2, up 2 quad) incl. odd sizes, asymmetric and negative pads."""
    up, down, pad = cfg
    g = torch.Generator().manual_seed(sum(shape) + up * 7 + down)
    x = torch.randn(*shape, generator=g)
    k = torch.rand(4, 4, generator=g)                       # NOT symmetric: catches a missing flip
    want = _upfirdn_ref(x, k, up, down, pad)
    if want.shape[1] <= 0 or want.shape[2] <= 0:
        pytest.skip('empty output')
    got = ops.upfirdn2d(x.to(DEV), k.to(DEV), up, down, pad)
    assert tuple(got.shape) == tuple(want.shape)
    assert rel(got, want) < 1e-5
    # fused epilogue: addend, second output with the activation derivative of a reference tensor
    addend = torch.randn(want.shape, generator=g)
    ref = torch.randn(want.shape, generator=g)
    o1, o2 = ops.upfirdn2d_fused(x.to(DEV), k.to(DEV), up, down, pad, addend=addend.to(DEV), act_ref=ref.to(DEV),
                                 slope=0.2, gain=math.sqrt(2), want_out=True, want_out2=True)
    v = want + addend
    assert rel(o1, v) < 1e-5
    assert rel(o2, v * torch.where(ref > 0, math.sqrt(2), 0.2 * math.sqrt(2))) < 1e-5
    none, o2b = ops.upfirdn2d_fused(x.to(DEV), k.to(DEV), up, down, pad, act_ref=ref.to(DEV), slope=0.2, gain=1.0,
                                    want_out=False, want_out2=True)
    assert none is None and rel(o2b, want * torch.where(ref > 0, 1.0, 0.2)) < 1e-5


@pytest.mark.parametrize('cfg', [(1, 1, (2, 1, 1, 2)), (1, 2, (1, 1, 1, 1)), (2, 1, (2, 1, 2, 1))])
def test_fir4_pointer_forms_beyond_65535_images(cfg):
    """More images than a grid's y dimension holds: the launch falls back from the branch-free kernels (one image per
    grid.y) to the pointer forms; same contract, and the same values as the branch-free kernels give on a slice."""
    up, down, pad = cfg
    g = torch.Generator().manual_seed(77 + up + down)
    x = torch.randn(65600, 5, 6, 4, generator=g)
    k = torch.rand(4, 4, generator=g)
    want = _upfirdn_ref(x, k, up, down, pad)
    addend = torch.randn(want.shape, generator=g)
    ref = torch.randn(want.shape, generator=g)
    o1, o2 = ops.upfirdn2d_fused(x.to(DEV), k.to(DEV), up, down, pad, addend=addend.to(DEV), act_ref=ref.to(DEV),
                                 slope=0.2, gain=1.3, want_out=True, want_out2=True)
    v = want + addend
    assert rel(o1, v) < 1e-5
    assert rel(o2, v * torch.where(ref > 0, 1.3, 0.2 * 1.3)) < 1e-5
    sl = slice(65000, 65600)
    s1, s2 = ops.upfirdn2d_fused(x[sl].contiguous().to(DEV), k.to(DEV), up, down, pad, addend=addend[sl].contiguous().to(DEV),
                                 act_ref=ref[sl].contiguous().to(DEV), slope=0.2, gain=1.3, want_out=True, want_out2=True)
    assert torch.equal(o1[sl], s1) and torch.equal(o2[sl], s2)


@pytest.mark.parametrize('size,small32,N', [(32, True, 8), (64, False, 4)])
def test_fused_trunk_equals_the_node_per_op_graph(size, small32, N):
    """_TrunkFn (one first-order node: folded 1/sqrt2, activation derivatives and gradient sums in the blur epilogues)
    against the any-order node family on the same weights and inputs: forward values and every parameter gradient."""
    import copy
    torch.manual_seed(size)
    D = ResidualDiscriminatorP(size, small32=small32, channel_multiplier=1.0).to(DEV).train()
    with torch.no_grad():
        for k, p in D.named_parameters():
            if k.endswith('bias'):
                p.normal_(0, 0.1)
    D2 = copy.deepcopy(D)
    D2.fuse_trunk = False
    x = torch.rand(N, 3, size, size, device=DEV)
    outs = []
    for m in (D, D2):
        logit, aux = m(x, sg_linear=True, projection=True, projection2=True)
        loss = (logit * 0.3).sum() + aux['projection'].pow(2).sum() + aux['projection2'].sin().sum()
        loss.backward()
        outs.append((logit.detach(), aux['projection'].detach()))
    assert rel(outs[0][0], outs[1][0]) < 1e-4 and rel(outs[0][1], outs[1][1]) < 1e-4
    for (k, p), q in zip(D.named_parameters(), D2.parameters()):
        assert p.grad is not None and q.grad is not None, k
        assert l2(p.grad, q.grad) < 1e-4, (k, l2(p.grad, q.grad))


def test_call_batches_equals_separate_calls():
    """ResidualDiscriminatorP.call_batches: train_stylegan2_contraD.py's two discriminator calls (fakes N, real views 2N)
    as one pass -- the minibatch-stddev statistics stay inside each call's batch, everything else is per sample."""
    import copy
    torch.manual_seed(5)
    D = ResidualDiscriminatorP(32, small32=True).to(DEV).train()
    D2 = copy.deepcopy(D)
    a, b = torch.rand(4, 3, 32, 32, device=DEV), torch.rand(8, 3, 32, 32, device=DEV)
    flags = dict(sg_linear=True, projection=True, projection2=True)
    (la, xa), (lb, xb) = D.call_batches([a, b], **flags)
    la2, xa2 = D2(a, **flags)
    lb2, xb2 = D2(b, **flags)
    assert la.shape == (4, 1) and lb.shape == (8, 1) and xa['projection'].shape == (4, 128)
    assert rel(la, la2) < 1e-5 and rel(lb, lb2) < 1e-5
    assert rel(xa['projection2'], xa2['projection2']) < 1e-5 and rel(xb['projection'], xb2['projection']) < 1e-5
    # NOT what one plain call on the concatenated batch gives (its stddev groups mix the two calls)
    lc, _ = copy.deepcopy(D)(torch.cat([a, b]), **flags)
    assert rel(lc[:4], la2) > 1e-4
    w = torch.randn(12, 1, device=DEV)
    ((torch.cat([la, lb]) * w).sum() + xa['projection'].pow(2).sum() + xb['projection2'].sin().sum()).backward()
    ((torch.cat([la2, lb2]) * w).sum() + xa2['projection'].pow(2).sum() + xb2['projection2'].sin().sum()).backward()
    for (k, p), q in zip(D.named_parameters(), D2.parameters()):
        assert l2(p.grad, q.grad) < 1e-4, (k, l2(p.grad, q.grad))


def test_call_merged_equals_call_batches_bit_for_bit():
    """ResidualDiscriminatorP.call_merged (the product path of the StyleGAN2 ContraD step: real views and fakes already side
    by side in one buffer) is call_batches without the split and the concatenation: logits, both projections and every
    parameter gradient bitwise equal; a wrong split is an error, not a silently mis-segmented minibatch stddev."""
    import copy
    torch.manual_seed(6)
    D = ResidualDiscriminatorP(32, small32=True).to(DEV).train()
    D2 = copy.deepcopy(D)
    a, b = torch.rand(8, 3, 32, 32, device=DEV), torch.rand(4, 3, 32, 32, device=DEV)
    flags = dict(sg_linear=True, projection=True, projection2=True)
    lm, xm = D.call_merged(torch.cat([a, b]), [8, 4], **flags)
    (la, xa), (lb, xb) = D2.call_batches([a, b], **flags)
    assert torch.equal(lm, torch.cat([la, lb]))
    for k in ('projection', 'projection2'):
        assert torch.equal(xm[k], torch.cat([xa[k], xb[k]]))
    w = torch.randn(12, 1, device=DEV)
    ((lm * w).sum() + xm['projection'].pow(2).sum() + xm['projection2'].sin().sum()).backward()
    ((torch.cat([la, lb]) * w).sum() + torch.cat([xa['projection'], xb['projection']]).pow(2).sum() +
     torch.cat([xa['projection2'], xb['projection2']]).sin().sum()).backward()
    for (k, p), q in zip(D.named_parameters(), D2.parameters()):
        assert torch.equal(p.grad, q.grad), k
    with pytest.raises(ValueError):
        D.call_merged(torch.cat([a, b]), [8, 8], **flags)


@pytest.mark.parametrize('size,small32,N', [(32, True, 8), (64, False, 4)])
def test_fused_r1_trunk_equals_the_node_family(size, small32, N):
    """_TrunkR1Fn / _TrunkVJPFn (round 3: the R1 call and the generator step on two fused nodes -- the trunk's backward
    chain as a differentiable node, its backward = the tangent pass of the linearised trunk) against the any-order node
    family on identical weights: r1, d D / d images, and every parameter gradient of (a) r1 alone, (b) a first-order loss
    plus r1, (c) the first-order image gradient of the generator step."""
    import copy
    from contrad_amd.engine import r1_loss
    torch.manual_seed(size + 1)
    D = ResidualDiscriminatorP(size, small32=small32, channel_multiplier=1.0).to(DEV).train()
    with torch.no_grad():
        for k, p in D.named_parameters():
            if k.endswith('bias'):
                p.normal_(0, 0.1)
            elif k.startswith('linear'):
                p.mul_(8.0)                                   # r1 = O(0.1 ... 1): the second-order terms carry the signal
    D2 = copy.deepcopy(D)
    D2.fuse_r1 = False
    x = torch.rand(N, 3, size, size, device=DEV)

    res = []
    for m in (D, D2):
        out = {}
        # (a) r1 alone
        m.zero_grad()
        r1 = r1_loss(m, x, lambda t: t)
        r1.backward()
        out['r1'] = r1.detach().clone()
        out['a'] = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
        # (b) first-order loss + r1 through the reference's own call pattern (no input_grad_only around autograd.grad)
        m.zero_grad()
        xa = x.clone().requires_grad_()
        d_real, aux = m(xa, projection=True, projection2=True)
        grad_real, = torch.autograd.grad(outputs=d_real.sum(), inputs=xa, create_graph=True, retain_graph=True)
        loss = torch.nn.functional.softplus(-d_real).mean() + aux['projection'].pow(2).mean() + \
            2.0 * grad_real.pow(2).reshape(N, -1).sum(1).mean()
        loss.backward()
        out['grad_real'] = grad_real.detach().clone()
        out['b'] = {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}
        # (c) generator step: D frozen, first-order gradient w.r.t. the images
        for p in m.parameters():
            p.requires_grad_(False)
        xg = x.clone().requires_grad_()
        o, a2 = m(xg, projection=True, projection2=True)
        (torch.nn.functional.softplus(-o).mean() + a2['projection2'].sin().mean()).backward()
        out['c'] = xg.grad.clone()
        assert all(p.grad is None or True for p in m.parameters())
        for p in m.parameters():
            p.requires_grad_(True)
        res.append(out)
    f, n = res
    assert float(n['r1']) > 0.02
    assert abs(float(f['r1']) - float(n['r1'])) < 1e-4 * float(n['r1'])
    assert l2(f['grad_real'], n['grad_real']) < 1e-4 and l2(f['c'], n['c']) < 1e-4
    for tag in ('a', 'b'):
        for k in n[tag]:
            gn, gf = n[tag][k], f[tag][k]
            if gn is None or gn.abs().max().item() == 0:
                assert gf is None or gf.abs().max().item() == 0, (tag, k)
                continue
            assert gf is not None, (tag, k)
            # (biases of r1 alone flow through the stddev curvature only: tiny, fp32-conditioned -- tests/test_r1_gradient_gpu.py)
            tol = 2e-3 if (tag == 'a' and k.endswith('bias')) else 1e-4
            assert l2(gf, gn) < tol, (tag, k, l2(gf, gn))


def test_fir_nontemporal_store_path_equals_the_plain_one():
    """FIR outputs of >= 256 MB are stored non-temporally (stylegan2_ops.hip, UpfirdnArgs.nt_store): the blur and the
    upsampling FIR with the fused epilogue on a 268 MB output equal the same calls on the two halves of the batch (plain
    stores) bit for bit -- the arithmetic per output does not depend on the batch."""
    g = torch.Generator(device=DEV).manual_seed(6)
    k1 = torch.tensor([1., 3., 3., 1.]); k = (k1[:, None] * k1[None, :] / 64).to(DEV)
    N, H, C = 16, 256, 64
    x = torch.randn(N, H, H, C, device=DEV, generator=g)
    ref = torch.randn(N, H, H, C, device=DEV, generator=g)
    _, o = ops.upfirdn2d_fused(x, k, 1, 1, (2, 1, 2, 1), act_ref=ref, slope=0.2, gain=1.4, want_out=False, want_out2=True)
    assert o.numel() * 4 >= 256 << 20
    h = N // 2
    for sl in (slice(0, h), slice(h, N)):
        _, oh = ops.upfirdn2d_fused(x[sl].contiguous(), k, 1, 1, (2, 1, 2, 1), act_ref=ref[sl].contiguous(), slope=0.2,
                                    gain=1.4, want_out=False, want_out2=True)
        assert torch.equal(o[sl], oh)
    xs = x[:, ::2, ::2].contiguous()
    up = ops.upfirdn2d(xs, k, 2, 1, (2, 1, 2, 1))
    assert up.numel() * 4 >= 256 << 20
    for sl in (slice(0, h), slice(h, N)):
        assert torch.equal(up[sl], ops.upfirdn2d(xs[sl].contiguous(), k, 2, 1, (2, 1, 2, 1)))
    assert rel(up[:1], _upfirdn_ref(xs[:1].cpu(), k.cpu(), 2, 1, (2, 1, 2, 1))) < 1e-5
