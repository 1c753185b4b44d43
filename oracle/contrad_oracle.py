"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the ContraD discriminator-step hot path.

A restatement, in plain PyTorch-CPU / numpy functional code, of the algorithm the reference
(jh-jeong/ContraD, mounted at /root/reference while this was written) runs on its hot path.
Every function cites the reference file:line it follows.  The arithmetic of conv / linear /
grid_sample / spectral_norm / Adam lives in PyTorch itself (the reference pins no version); the
oracle calls the same PyTorch-CPU primitives (torch 2.10 in this image).

Pinning: the reference ships NO tests and NO golden vectors for this path,
so the oracle is pinned against outputs of the reference itself, imported in the build container
by ``tests/golden/make_golden.py`` (which asserts oracle == reference at generation time and
commits the resulting vectors under ``tests/golden/*.npz``).  Exception: ``gaussian_blur``
restates kornia's ``filter2D``/``get_gaussian_kernel2d`` API contract (kornia is neither vendored
nor installed) -- that one function is "parity unpinned" and validated by properties only.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  The product (``contrad_amd``) must never import it.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# A. SimCLR augmentation  (augment/__init__.py:106-122)
# ----------------------------------------------------------------------------------------------

SIMCLR_CIFAR = dict(scale=(0.2, 1.0), ratio=(3. / 4., 4. / 3.), brightness=0.4, contrast=0.4,
                    saturation=0.4, hue=0.1, p_jitter=0.8, p_gray=0.2)          # configs/defaults/augment.gin:11-18
SIMCLR_HQ_AFHQ = dict(scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.), brightness=0.8, contrast=0.8,
                      saturation=0.8, hue=0.2, p_jitter=0.8, p_gray=0.2, p_blur=0.5,
                      sigma_range=(0.1, 2.0))                                   # afhq_dog_style64.gin:16-20


SIMCLR_HQ_CUTOUT_AFHQ = dict(SIMCLR_HQ_AFHQ, p_cutout=0.5, cutout_length=15)   # augment/__init__.py:124-133, augment.gin:16


def _jitter_range(value, center=1.0, clip_first_on_zero=True):
    """ColorJitterLayer._check_input for a scalar (augment/color_jitter.py:25-42)."""
    lo, hi = center - value, center + value
    if clip_first_on_zero:
        lo = max(lo, 0)
    if lo == hi == center:
        return None
    return [lo, hi]


def sample_resized_crop_theta(B, dim2, dim3, scale, ratio):
    """Host part of RandomResizeCropLayer.forward (augment/spatial.py:111-143): numpy global RNG.

    The reference names ``inputs.shape[2]`` "width" and ``shape[3]`` "height"; theta[0,0] (the x /
    last-axis scale) is w/shape[2].  Reproduced as written.
    """
    width, height = dim2, dim3
    theta = torch.eye(2, 3).repeat(B, 1, 1)
    area = height * width
    target_area = np.random.uniform(*scale, B * 10) * area
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    aspect_ratio = np.exp(np.random.uniform(*log_ratio, B * 10))
    w = np.round(np.sqrt(target_area * aspect_ratio))
    h = np.round(np.sqrt(target_area / aspect_ratio))
    cond = (0 < w) * (w <= width) * (0 < h) * (h <= height)
    w = w[cond]
    h = h[cond]
    if len(w) > B:
        inds = np.random.choice(len(w), B, replace=False)
        w = w[inds]
        h = h[inds]
    n = len(w)
    r_w_bias = np.random.randint(w - width, width - w + 1) / width
    r_h_bias = np.random.randint(h - height, height - h + 1) / height
    w = w / width
    h = h / height
    theta[:n, 0, 0] = torch.tensor(w)
    theta[:n, 1, 1] = torch.tensor(h)
    theta[:n, 0, 2] = torch.tensor(r_w_bias)
    theta[:n, 1, 2] = torch.tensor(r_h_bias)
    return theta


def sample_simclr_params(B, H, W, cfg):
    """Random parameters of one ``simclr()`` / ``simclr_hq()`` call, in the reference's draw order
    (SURVEY.md section 8a row A7): np(crop) -> torch bernoulli(flip) -> torch bernoulli(jitter
    mask) -> np.rand (order) -> torch uniforms in executed order -> torch bernoulli(gray mask)
    [-> torch bernoulli(blur mask) -> np.uniform sigma]."""
    p = {}
    p['theta'] = sample_resized_crop_theta(B, H, W, cfg['scale'], cfg['ratio'])
    p['flip_sign'] = torch.bernoulli(torch.ones(B) * 0.5) * 2 - 1            # spatial.py:89
    p['jitter_mask'] = torch.bernoulli(torch.full((B,), cfg['p_jitter']))     # augment/__init__.py:101-102
    p['contrast_first'] = bool(np.random.rand() > 0.5)                        # color_jitter.py:67
    r_c = _jitter_range(cfg['contrast'])
    r_h = _jitter_range(cfg['hue'], center=0, clip_first_on_zero=False)
    r_s = _jitter_range(cfg['saturation'])
    r_v = _jitter_range(cfg['brightness'])

    def _draw_contrast():
        if r_c:
            p['f_contrast'] = torch.empty(B, 1, 1, 1).uniform_(*r_c).view(B)   # color_jitter.py:46
        else:
            p['f_contrast'] = None

    def _draw_hsv():
        f_h = torch.zeros(B, 1, 1)
        f_s = torch.ones(B, 1, 1)
        f_v = torch.ones(B, 1, 1)
        if r_h:
            f_h.uniform_(*r_h)                                                # color_jitter.py:56-61
        if r_s:
            f_s = f_s.uniform_(*r_s)
        if r_v:
            f_v = f_v.uniform_(*r_v)
        p['f_h'], p['f_s'], p['f_v'] = f_h.view(B), f_s.view(B), f_v.view(B)

    if p['contrast_first']:
        _draw_contrast(); _draw_hsv()
    else:
        _draw_hsv(); _draw_contrast()
    p['gray_mask'] = torch.bernoulli(torch.full((B,), cfg['p_gray']))
    if 'p_blur' in cfg:
        p['blur_mask'] = torch.bernoulli(torch.full((B,), cfg['p_blur']))
        p['sigma'] = float(np.random.uniform(*cfg['sigma_range']))            # augment/__init__.py:73
    if 'p_cutout' in cfg:
        p['cut_mask'] = torch.bernoulli(torch.full((B,), cfg['p_cutout']))
        p['cut_h'] = torch.randint(H, (B, 1)).view(B)                         # spatial.py:169-170
        p['cut_w'] = torch.randint(W, (B, 1)).view(B)
        p['cut_length'] = cfg['cutout_length']
    return p


def resized_crop(x, theta):
    """Device part of RandomResizeCropLayer (spatial.py:145-146)."""
    grid = F.affine_grid(theta, x.size(), align_corners=False)
    return F.grid_sample(x, grid, padding_mode='reflection', align_corners=False)


def hflip(x, sign):
    """HorizontalFlipLayer (spatial.py:84-93)."""
    theta = torch.eye(2, 3).repeat(x.size(0), 1, 1)
    theta[:, 0, 0] = sign
    grid = F.affine_grid(theta, x.size(), align_corners=False)
    return F.grid_sample(x, grid, padding_mode='reflection', align_corners=False)


def rgb2hsv(rgb):
    """augment/utils.py:6-37 (atan2 hue, not the lookup form)."""
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    cmax = rgb.max(1)[0]
    cmin = rgb.min(1)[0]
    hue = torch.atan2(math.sqrt(3) * (g - b), 2 * r - g - b)
    hue = (hue % (2 * math.pi)) / (2 * math.pi)
    sat = 1 - cmin / (cmax + 1e-8)
    hsv = torch.stack([hue, sat, cmax], dim=1)
    hsv[~torch.isfinite(hsv)] = 0.
    return hsv


def hsv2rgb(hsv):
    """augment/utils.py:40-63."""
    h, s, v = hsv[:, [0]], hsv[:, [1]], hsv[:, [2]]
    c = v * s
    n = hsv.new_tensor([5, 3, 1]).view(3, 1, 1)
    k = (n + h * 6) % 6
    t = torch.clamp(torch.min(k, 4. - k), 0, 1)
    return v - c * t


class _StraightThroughHSV(torch.autograd.Function):
    """RandomHSVFunction (color_jitter.py:81-104): forward = the HSV jitter, backward = identity on x."""

    @staticmethod
    def forward(ctx, x, f_h, f_s, f_v):
        return _adjust_hsv_values(x, f_h, f_s, f_v)

    @staticmethod
    def backward(ctx, g):
        return g.clone(), None, None, None


def adjust_hsv(x, f_h, f_s, f_v):
    """RandomHSVFunction.apply: values as color_jitter.py:83-95, gradient straight-through (:97-104)."""
    if x.requires_grad:
        return _StraightThroughHSV.apply(x, f_h, f_s, f_v)
    return _adjust_hsv_values(x, f_h, f_s, f_v)


def _adjust_hsv_values(x, f_h, f_s, f_v):
    """RandomHSVFunction.forward (color_jitter.py:83-95), incl. the 255/360 hue quirk."""
    B = x.size(0)
    hsv = rgb2hsv(x)
    h = hsv[:, 0] + f_h.view(B, 1, 1) * 255. / 360.
    hsv[:, 0] = h % 1
    hsv[:, 1] = hsv[:, 1] * f_s.view(B, 1, 1)
    hsv[:, 2] = hsv[:, 2] * f_v.view(B, 1, 1)
    hsv = torch.clamp(hsv, 0, 1)
    return hsv2rgb(hsv)


def adjust_contrast(x, f):
    """ColorJitterLayer.adjust_contrast (color_jitter.py:44-49); clamp applies even when f is None."""
    if f is not None:
        means = torch.mean(x, dim=[2, 3], keepdim=True)
        x = (x - means) * f.view(-1, 1, 1, 1) + means
    return torch.clamp(x, 0, 1)


def color_jitter(x, contrast_first, f_contrast, f_h, f_s, f_v):
    """ColorJitterLayer.transform (color_jitter.py:65-75)."""
    if contrast_first:
        return adjust_hsv(adjust_contrast(x, f_contrast), f_h, f_s, f_v)
    return adjust_contrast(adjust_hsv(x, f_h, f_s, f_v), f_contrast)


def color_gray(x):
    """RandomColorGrayLayer (augment/__init__.py:82-91)."""
    w = torch.tensor([[0.299, 0.587, 0.114]]).view(1, 3, 1, 1)
    l = F.conv2d(x, w)
    return torch.cat([l, l, l], dim=1)


def blend(x, fx, mask):
    """RandomApply.forward (augment/__init__.py:100-103)."""
    m = mask.view(-1, 1, 1, 1)
    return x * (1 - m) + fx * m


def gaussian_kernel1d(ksize, sigma):
    """kornia.filters.get_gaussian_kernel1d contract (odd ksize): exp(-(i-k//2)^2/(2 s^2)) / sum."""
    xs = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    g = torch.exp(-xs.pow(2) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_blur(x, sigma):
    """GaussianBlur.forward (augment/__init__.py:64-78) on kornia's filter2D contract:
    2-D kernel = outer(g, g), reflect padding, depthwise correlation.  PARITY UNPINNED (kornia absent)."""
    B, C, H, W = x.shape
    radius = int((H // 10) / 2)
    ksize = radius * 2 + 1
    g = gaussian_kernel1d(ksize, sigma)
    k2 = torch.outer(g, g).view(1, 1, ksize, ksize).repeat(C, 1, 1, 1)
    xp = F.pad(x, [radius, radius, radius, radius], mode='reflect')
    return F.conv2d(xp, k2, groups=C)


def cutout(x, h_center, w_center, length):
    """CutOut.forward (augment/spatial.py:163-181) with explicit centres."""
    N, _, h, w = x.shape
    mask_h = x.new_zeros(N, h).scatter_(1, h_center.view(N, 1), 1).unsqueeze(1)
    mask_w = x.new_zeros(N, w).scatter_(1, w_center.view(N, 1), 1).unsqueeze(1)
    wgt = torch.ones(1, 1, length)
    mask_h = F.conv1d(mask_h, wgt, padding=(length - 1) // 2)
    mask_w = F.conv1d(mask_w, wgt, padding=(length - 1) // 2)
    return x * (1. - torch.einsum('bci,bcj->bcij', mask_h, mask_w))


def simclr_apply(x, p):
    """nn.Sequential of simclr()/simclr_hq() with explicit parameters ``p``."""
    x = resized_crop(x, p['theta'])
    x = hflip(x, p['flip_sign'])
    x = blend(x, color_jitter(x, p['contrast_first'], p['f_contrast'], p['f_h'], p['f_s'], p['f_v']),
              p['jitter_mask'])
    x = blend(x, color_gray(x), p['gray_mask'])
    if 'blur_mask' in p:
        x = blend(x, gaussian_blur(x, p['sigma']), p['blur_mask'])
    if 'cut_mask' in p:
        x = blend(x, cutout(x, p['cut_h'], p['cut_w'], p['cut_length']), p['cut_mask'])
    return x


# ----------------------------------------------------------------------------------------------
# L. Losses  (training/criterion.py:24-45, training/gan/contrad.py:8-32)
# ----------------------------------------------------------------------------------------------

def nt_xent(out1, out2, temperature=0.1):
    """training/criterion.py:24-45 (inputs already L2-normalised; single process)."""
    N = out1.size(0)
    z = torch.cat([out1, out2], dim=0)
    sim = (z @ z.t()) / temperature
    sim = sim.clone()
    sim.fill_diagonal_(-5e4)
    lsm = F.log_softmax(sim, dim=1)
    return -torch.sum(lsm[:N, N:].diag() + lsm[N:, :N].diag()) / (2 * N)


def supcon_fake(out1, out2, others, temperature):
    """training/gan/contrad.py:8-32."""
    N = out1.size(0)
    z = torch.cat([out1, out2, others], dim=0)
    sim = (z @ z.t()) / temperature
    sim = sim.clone()
    sim.fill_diagonal_(-5e4)
    mask = torch.zeros_like(sim)
    mask[2 * N:, 2 * N:] = 1
    mask.fill_diagonal_(0)
    sim = sim[2 * N:]
    mask = mask[2 * N:]
    mask = mask / mask.sum(1, keepdim=True)
    lsm = F.log_softmax(sim, dim=1) * mask
    return -lsm.sum(1).mean()


def gan_d_loss(d_real, d_gen, kind):
    """training/gan/contrad.py:52-64."""
    if kind == 'nonsat':
        return F.softplus(d_gen).mean() + F.softplus(-d_real).mean()
    if kind == 'wgan':
        return d_gen.mean() - d_real.mean()
    if kind == 'hinge':
        return F.relu(1. + d_gen).mean() + F.relu(1. - d_real).mean()
    if kind == 'lsgan':
        return 0.5 * (((d_real - 1.0) ** 2).mean() + (d_gen ** 2).mean())
    raise NotImplementedError()


# ----------------------------------------------------------------------------------------------
# D. SNDCGAN discriminator (models/gan/sndcgan.py:69-148, models/gan/base.py:79-150)
# ----------------------------------------------------------------------------------------------

# (cin, cout, k, stride, pad) of D_SNDCGAN.main (sndcgan.py:91-109); conv i lives at main.{2*i}
SNDCGAN_D_CONVS = [(3, 64, 3, 1, 1), (64, 128, 4, 2, 1), (128, 128, 3, 1, 1), (128, 256, 4, 2, 1),
                   (256, 256, 3, 1, 1), (256, 512, 4, 2, 1), (512, 512, 3, 1, 1)]


def sndcgan_d_param_shapes(image_hw=32, d_hidden=512, d_project=128):
    """State-dict names and shapes of D_SNDCGAN(mlp_linear=True, d_hidden=512)."""
    feat = 512 * (image_hw // 8) ** 2
    shapes = {}

    def sn(prefix, wshape):
        out = wshape[0]
        inn = int(np.prod(wshape[1:]))
        shapes[prefix + '.bias'] = (out,)
        shapes[prefix + '.weight_orig'] = tuple(wshape)
        shapes[prefix + '.weight_u'] = (out,)
        shapes[prefix + '.weight_v'] = (inn,)

    sn('linear.l1', (d_hidden, feat))
    sn('linear.l2', (1, d_hidden))
    sn('projection.0', (d_hidden, feat))
    sn('projection.2', (d_project, d_hidden))
    sn('projection2.0', (d_hidden, feat))
    sn('projection2.2', (d_project, d_hidden))
    for i, (ci, co, k, s, p) in enumerate(SNDCGAN_D_CONVS):
        sn('main.%d' % (2 * i), (co, ci, k, k))
    return shapes


def spectral_norm_weight(sd, prefix, training=True, eps=1e-12):
    """torch.nn.utils.spectral_norm pre-forward hook (applied at sndcgan.py:111-118): ONE power
    iteration per forward in train mode, u/v updated in place under no_grad, sigma = u^T W v with
    u, v constant in the graph; returns W_orig / sigma."""
    w = sd[prefix + '.weight_orig']
    u = sd[prefix + '.weight_u']
    v = sd[prefix + '.weight_v']
    wm = w.reshape(w.size(0), -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    sigma = torch.dot(u.clone(), torch.mv(wm, v.clone()))
    return w / sigma


def _lrelu(y, slope, mask=None):
    """leaky_relu; with ``mask`` (bool, True where the unit is in its positive region) the linear region is
    imposed from outside instead of taken from sign(y).  Used by the GPU parity tests to evaluate the CPU
    gradient on the SAME piecewise-linear region as the kernel under test: leaky_relu's derivative is
    discontinuous at 0, and a pre-activation of ~1e-8 (pure fp32 summation-order noise) otherwise flips a slope
    1 <-> 0.1 and pollutes an element-wise gradient comparison with an error that is not a kernel error."""
    if mask is None:
        return F.leaky_relu(y, slope)
    return y * torch.where(mask, torch.ones((), dtype=y.dtype), torch.full((), slope, dtype=y.dtype))


def sndcgan_d_features(sd, x, training=True, act_masks=None):
    """D_SNDCGAN.penultimate (sndcgan.py:122-128)."""
    h = x * 2. - 1.
    for i, (ci, co, k, s, p) in enumerate(SNDCGAN_D_CONVS):
        pre = 'main.%d' % (2 * i)
        h = F.conv2d(h, spectral_norm_weight(sd, pre, training), sd[pre + '.bias'], stride=s, padding=p)
        h = _lrelu(h, 0.1, None if act_masks is None else act_masks[i])
    return h.reshape(h.size(0), -1)


def _sn_linear(sd, prefix, x, training):
    return F.linear(x, spectral_norm_weight(sd, prefix, training), sd[prefix + '.bias'])


def d_heads(sd, features, sg_linear, training=True, hidden_masks=None):
    """BaseDiscriminator.forward heads (base.py:122-133) with TinyDiscriminator (base.py:14-35)."""
    fd = features.detach() if sg_linear else features
    hm = hidden_masks if hidden_masks is not None else (None, None, None)
    out = _sn_linear(sd, 'linear.l2', _lrelu(_sn_linear(sd, 'linear.l1', fd, training), 0.1, hm[0]), training)
    proj = _sn_linear(sd, 'projection.2',
                      _lrelu(_sn_linear(sd, 'projection.0', features, training), 0.1, hm[1]), training)
    proj2 = _sn_linear(sd, 'projection2.2',
                       _lrelu(_sn_linear(sd, 'projection2.0', features, training), 0.1, hm[2]), training)
    out = out + (proj.mean() + proj2.mean()) * 0.
    return out, proj, proj2


def sndcgan_d_forward(sd, x, sg_linear=False, training=True, act_masks=None, hidden_masks=None):
    feats = sndcgan_d_features(sd, x, training, act_masks)
    out, proj, proj2 = d_heads(sd, feats, sg_linear, training, hidden_masks)
    return out, proj, proj2, feats


# ----------------------------------------------------------------------------------------------
# D_SNResNet18 (models/gan/snresnet.py:21-89, built at models/gan/__init__.py:8-12 with d_hidden=1024)
# ----------------------------------------------------------------------------------------------
SNRESNET18_STAGES = [(64, 1), (128, 2), (256, 2), (512, 2)]      # (planes, stride of the first block), 2 blocks each


def snresnet18_blocks():
    """[(prefix, in_planes, planes, stride, has_shortcut)] in module order (snresnet.py:68-75)."""
    out, inp = [], 64
    for li, (planes, stride) in enumerate(SNRESNET18_STAGES, 1):
        for bi, s in enumerate([stride, 1]):
            out.append(('layer%d.%d' % (li, bi), inp, planes, s, s != 1 or inp != planes))
            inp = planes
    return out


def snresnet18_param_shapes(d_hidden=1024, d_project=128):
    shapes = {}

    def sn(prefix, wshape):
        out = wshape[0]
        shapes[prefix + '.bias'] = (out,)
        shapes[prefix + '.weight_orig'] = tuple(wshape)
        shapes[prefix + '.weight_u'] = (out,)
        shapes[prefix + '.weight_v'] = (int(np.prod(wshape[1:])),)

    sn('linear.l1', (d_hidden, 512)); sn('linear.l2', (1, d_hidden))
    sn('projection.0', (d_hidden, 512)); sn('projection.2', (d_project, d_hidden))
    sn('projection2.0', (d_hidden, 512)); sn('projection2.2', (d_project, d_hidden))
    sn('conv1', (64, 3, 3, 3))
    for pre, inp, planes, s, sc in snresnet18_blocks():
        sn(pre + '.conv1', (planes, inp, 3, 3))
        sn(pre + '.conv2', (planes, planes, 3, 3))
        if sc:
            sn(pre + '.shortcut.0', (planes, inp, 1, 1))
    return shapes


def snresnet18_features(sd, x, training=True, act_masks=None):
    """SNResNet.penultimate (snresnet.py:77-89) with BasicBlock.forward (:36-41).  act_masks: optional list of boolean
    masks (one per LeakyReLU, in execution order) that fix the linear region instead of the sign of the pre-activation
    (same-region parity tests)."""
    masks = iter(act_masks) if act_masks is not None else None

    def act(y):
        return _lrelu(y, 0.1, next(masks) if masks is not None else None)

    def conv(pre, h, stride, pad):
        return F.conv2d(h, spectral_norm_weight(sd, pre, training), sd[pre + '.bias'], stride=stride, padding=pad)
    h = act(conv('conv1', x * 2. - 1., 1, 1))
    for pre, inp, planes, s, sc in snresnet18_blocks():
        o = act(conv(pre + '.conv1', h, s, 1))
        o = conv(pre + '.conv2', o, 1, 1)
        o = o + (conv(pre + '.shortcut.0', h, s, 0) if sc else h)
        h = act(o)
    h = F.avg_pool2d(h, 4)
    return h.reshape(h.size(0), -1)


def snresnet18_forward(sd, x, sg_linear=False, training=True, act_masks=None, hidden_masks=None):
    feats = snresnet18_features(sd, x, training, act_masks)
    out, proj, proj2 = d_heads(sd, feats, sg_linear, training, hidden_masks)
    return out, proj, proj2, feats


def simclr_only_loss_d(d_forward, images_aug2n, N, temp=0.1):
    """simclr_only.loss_D_fn (training/gan/simclr_only.py:9-21) on already-augmented cat([x, x]):
    d_forward(x) -> (logits, projection)."""
    d, proj = d_forward(images_aug2n)
    views = F.normalize(proj + d.mean() * 0)
    return nt_xent(views[:N], views[N:], temperature=temp)


def gan_g_loss(d_gen, kind):
    """contrad.loss_G_fn's loss (training/gan/contrad.py:75-81)."""
    if kind == 'nonsat':
        return F.softplus(-d_gen).mean()
    if kind == 'lsgan':
        return 0.5 * ((d_gen - 1.0) ** 2).mean()
    return -d_gen.mean()


def contrad_loss_d(d_forward, images_aug3n, N, temp=0.1, lbd_a=1.0, loss='nonsat'):
    """contrad.loss_D_fn (training/gan/contrad.py:35-70) on already-augmented cat([x,x,G(z)]);
    returns (simclr + lbd_a*sup, gan_loss, d_real_mean, d_gen_mean)."""
    d_all, proj, proj2 = d_forward(images_aug3n)
    views = F.normalize(proj)
    simclr = nt_xent(views[:N], views[N:2 * N], temperature=temp)
    reals = F.normalize(proj2)
    sup = supcon_fake(reals[:N], reals[N:2 * N], reals[2 * N:], temperature=temp)
    d_real, d_gen = d_all[:N], d_all[2 * N:3 * N]
    return simclr + lbd_a * sup, gan_d_loss(d_real, d_gen, loss), d_real.mean(), d_gen.mean()


# ----------------------------------------------------------------------------------------------
# G. SNDCGAN generator forward (models/gan/sndcgan.py:13-52)
# ----------------------------------------------------------------------------------------------

# ConvTranspose2d (cin, cout, k, stride, pad) of G_SNDCGAN.main at indices 0,3,6,9
SNDCGAN_G_CONVT = [(512, 256, 4, 2, 1), (256, 128, 4, 2, 1), (128, 64, 4, 2, 1), (64, 3, 3, 1, 1)]


def sndcgan_g_param_shapes(image_hw=32, nz=128):
    hb = image_hw // 8
    f = 512 * hb * hb
    shapes = {'linear.weight': (f, nz), 'linear.bias': (f,),
              'norm_init.weight': (f,), 'norm_init.bias': (f,),
              'norm_init.running_mean': (f,), 'norm_init.running_var': (f,)}
    for j, (ci, co, k, s, p) in enumerate(SNDCGAN_G_CONVT):
        shapes['main.%d.weight' % (3 * j)] = (ci, co, k, k)
        shapes['main.%d.bias' % (3 * j)] = (co,)
        if j < 3:
            bn = 'main.%d' % (3 * j + 1)
            for nm in ('weight', 'bias', 'running_mean', 'running_var'):
                shapes[bn + '.' + nm] = (co,)
    return shapes


def _bn_train(sd, prefix, x, momentum=0.1, eps=1e-5, training=True):
    """nn.BatchNorm2d in train mode: biased batch variance to normalise, running stats updated with
    the unbiased one (torch semantics; G stays in .train() during the D-step, train_gan.py:142).
    ``training=False``: eval mode, the running statistics normalise (sampling from a checkpoint, train_gan.py:181)."""
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'],
                        sd[prefix + '.weight'], sd[prefix + '.bias'], training, momentum, eps)


def sndcgan_g_forward(sd, z, image_hw=32, training=True):
    """G_SNDCGAN.forward (sndcgan.py:41-48); ``training=False`` = G.eval()."""
    hb = image_hw // 8
    h = F.linear(z, sd['linear.weight'], sd['linear.bias'])
    h = h.view(h.size(0), h.size(1), 1, 1)
    h = F.relu(_bn_train(sd, 'norm_init', h, training=training))
    h = h.view(-1, 512, hb, hb)
    for j, (ci, co, k, s, p) in enumerate(SNDCGAN_G_CONVT):
        h = F.conv_transpose2d(h, sd['main.%d.weight' % (3 * j)], sd['main.%d.bias' % (3 * j)],
                               stride=s, padding=p)
        if j < 3:
            h = F.relu(_bn_train(sd, 'main.%d' % (3 * j + 1), h, training=training))
    return 0.5 * torch.tanh(h) + 0.5


def sample_latent_sndcgan(n, nz=128):
    """G_SNDCGAN.sample_latent (sndcgan.py:50-52): U(-1,1) from the CPU generator."""
    return torch.empty(n, nz).uniform_(-1, 1)


# ----------------------------------------------------------------------------------------------
# O. Adam (torch.optim.Adam as constructed at train_gan.py:273-274) and LR warm-up (:88-93)
# ----------------------------------------------------------------------------------------------

def adam_step(p, g, m, v, step, lr, beta1, beta2, eps=1e-8):
    """One torch.optim.Adam update (no weight decay, no amsgrad), in place; ``step`` is 1-based."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def warmup_lr(cur_step, warmup, lr):
    """_update_warmup (train_gan.py:88-93)."""
    if warmup > 0:
        return min(1., (cur_step + 1) / warmup) * lr
    return lr


# ----------------------------------------------------------------------------------------------
# Deterministic fills shared by the golden generator and the tests (fixture hygiene)
# ----------------------------------------------------------------------------------------------

def det_fill(shapes, seed=1234, weight_std=0.02, bias_std=0.02):
    """Deterministic parameter/buffer fill keyed on the (ordered) name list: tensor i is drawn from
    a torch.Generator seeded ``seed + i``.  u/v buffers are unit vectors, running_var is positive."""
    sd = {}
    for i, (name, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(seed + i)
        t = torch.randn(*shape, generator=g)
        if name.endswith('weight_u') or name.endswith('weight_v'):
            t = F.normalize(t, dim=0, eps=1e-12)
        elif name.endswith('running_var'):
            t = t.abs() * 0.5 + 0.5
        elif name.endswith('running_mean'):
            t = t * 0.1
        elif name.endswith('bias'):
            t = t * bias_std
        elif 'norm' in name or (name.startswith('main.') and name.endswith('.weight') and len(shape) == 1):
            t = 1.0 + 0.1 * t          # BatchNorm gamma
        else:
            t = t * weight_std
        sd[name] = t
    return sd
