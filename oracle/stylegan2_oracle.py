"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the StyleGAN2 part of the ContraD hot path
(SURVEY.md 8a rows D4, D5, R1, G0): plain PyTorch-CPU restatement of models/gan/stylegan2/{layers,
discriminator,generator}.py and op/{upfirdn2d,fused_act}.py, functional over a state dict with the reference's
key names.  Pinned by tests/golden/make_golden.py against the imported reference (CPU path: upfirdn2d_native,
pure-torch fused_leaky_relu); see oracle/contrad_oracle.py for the conventions.  Never imported by the product.
"""
import math

import torch
import torch.nn.functional as F

from .contrad_oracle import _lrelu


def make_kernel(k=(1, 3, 3, 1)):
    """stylegan2/layers.py:24-32."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d_native (op/upfirdn2d.py:159-200) on NCHW: zero-insert, pad (may be negative), correlate with the
    flipped kernel, decimate.  pad = (pad0, pad1) applied to both axes like the reference wrapper (:144-156)."""
    B, C, H, W = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    out = x.reshape(B * C, H, 1, W, 1)
    out = F.pad(out, [0, up - 1, 0, 0, 0, up - 1])
    out = out.reshape(B * C, 1, H * up, W * up)
    out = F.pad(out, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    out = out[:, :, max(-p0, 0): out.shape[2] - max(-p1, 0), max(-p0, 0): out.shape[3] - max(-p1, 0)]
    out = F.conv2d(out, torch.flip(kernel, [0, 1]).view(1, 1, kh, kw))
    out = out[:, :, ::down, ::down]
    return out.reshape(B, C, out.shape[2], out.shape[3])


def fused_leaky_relu(x, bias, slope=0.2, scale=2 ** 0.5, mask=None):
    """op/fused_act.py:86-94 (the live pure-torch branch)."""
    rest = [1] * (x.ndim - 2)
    return _lrelu(x + bias.view(1, -1, *rest), slope, mask) * scale


def equal_conv(x, w, stride=1, padding=0):
    """EqualConv2d.forward (layers.py:114-123): weight * 1/sqrt(fan_in) at run time, no bias here."""
    scale = 1 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
    return F.conv2d(x, w * scale, None, stride=stride, padding=padding)


def minibatch_stddev(x, group=4):
    """_minibatch_stddev_layer (discriminator.py:22-33)."""
    B, C, H, W = x.shape
    g = min(B, group)
    s = x.view(g, -1, 1, C, H, W)
    s = torch.sqrt(s.var(0, unbiased=False) + 1e-8)
    s = s.mean([2, 3, 4], keepdim=True).mean(2)
    s = s.repeat(g, 1, H, W)
    return torch.cat([x, s], 1)


def d_channels(size, small32, channel_multiplier=2):
    if small32:
        return {4: 512, 8: 512, 16: 256, 32: 128}
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: int(256 * channel_multiplier), 128: int(128 * channel_multiplier),
            256: int(64 * channel_multiplier), 512: int(32 * channel_multiplier), 1024: int(16 * channel_multiplier)}


def d_param_shapes(size, small32, channel_multiplier=2, d_hidden=512, d_project=128):
    """State-dict names/shapes of ResidualDiscriminatorP (discriminator.py:191-235, base.py:79-101)."""
    ch = d_channels(size, small32, channel_multiplier)
    feat = ch[4] * 16
    s = {}
    for pre, (o, i) in (('linear.l1', (d_hidden, feat)), ('linear.l2', (1, d_hidden)),
                        ('projection.0', (d_hidden, feat)), ('projection.2', (d_project, d_hidden)),
                        ('projection2.0', (d_hidden, feat)), ('projection2.2', (d_project, d_hidden))):
        s[pre + '.weight'] = (o, i)
        s[pre + '.bias'] = (o,)
    s['layers.0.0.weight'] = (ch[size], 3, 1, 1)
    s['layers.0.1.bias'] = (ch[size],)
    cin = ch[size]
    k = 1
    for i in range(int(math.log2(size)), 2, -1):
        cout = ch[2 ** (i - 1)]
        p = 'layers.%d.' % k
        s[p + 'conv1.0.weight'] = (cin, cin, 3, 3)
        s[p + 'conv1.1.bias'] = (cin,)
        s[p + 'conv2.0.kernel'] = (4, 4)
        s[p + 'conv2.1.weight'] = (cout, cin, 3, 3)
        s[p + 'conv2.2.bias'] = (cout,)
        s[p + 'skip.0.kernel'] = (4, 4)
        s[p + 'skip.1.weight'] = (cout, cin, 1, 1)
        cin = cout
        k += 1
    s['last_conv.0.weight'] = (ch[4], cin + 1, 3, 3)
    s['last_conv.1.bias'] = (ch[4],)
    return s


def det_fill_d(shapes, seed=2024, head_std=0.02):
    """Deterministic fill: weights N(0,1) (EqualConv init), head weights N(0, head_std), biases N(0, 0.1), blur kernels
    exact.  (The R1-gradient fixtures use a larger ``head_std`` so that the penalty is O(0.1 ... 1) instead of 1e-3 ...
    1e-5: d D / d x scales with the product of the two ``linear`` head layers.)"""
    sd = {}
    for i, (name, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(seed + i)
        if name.endswith('kernel'):
            sd[name] = make_kernel()
            continue
        t = torch.randn(*shape, generator=g)
        if name.endswith('bias'):
            t = t * 0.1
        elif name.startswith(('linear', 'projection')):
            t = t * head_std
        sd[name] = t
    return sd


def d_features(sd, x, size, masks=None):
    """ResidualDiscriminatorP.penultimate (discriminator.py:225-235) with ResBlock (:60-76) and ConvLayer
    (layers.py:174-198).  ``masks``: optional list of leaky-relu linear-region masks in execution order."""
    mi = [0]

    def act(y, bias):
        m = None
        if masks is not None:
            m = masks[mi[0]]
            mi[0] += 1
        return fused_leaky_relu(y, bias, mask=m)

    h = x * 2. - 1.
    h = act(equal_conv(h, sd['layers.0.0.weight']), sd['layers.0.1.bias'])
    k = 1
    for _ in range(int(math.log2(size)), 2, -1):
        p = 'layers.%d.' % k
        o = act(equal_conv(h, sd[p + 'conv1.0.weight'], padding=1), sd[p + 'conv1.1.bias'])
        o = upfirdn2d(o, sd[p + 'conv2.0.kernel'], pad=(2, 2))
        o = act(equal_conv(o, sd[p + 'conv2.1.weight'], stride=2), sd[p + 'conv2.2.bias'])
        s = upfirdn2d(h, sd[p + 'skip.0.kernel'], pad=(1, 1))
        s = equal_conv(s, sd[p + 'skip.1.weight'], stride=2)
        h = (o + s) / math.sqrt(2)
        k += 1
    h = minibatch_stddev(h)
    h = act(equal_conv(h, sd['last_conv.0.weight'], padding=1), sd['last_conv.1.bias'])
    return h.reshape(h.size(0), -1)


def d_forward(sd, x, size, sg_linear=False, masks=None, head_masks=None):
    """BaseDiscriminator.forward (base.py:107-150) with plain nn.Linear heads (no spectral norm in StyleGAN2)."""
    f = d_features(sd, x, size, masks)
    fd = f.detach() if sg_linear else f
    hm = head_masks if head_masks is not None else (None, None, None)

    def lin(pre, t):
        return F.linear(t, sd[pre + '.weight'], sd[pre + '.bias'])

    out = lin('linear.l2', _lrelu(lin('linear.l1', fd), 0.1, hm[0]))
    proj = lin('projection.2', _lrelu(lin('projection.0', f), 0.1, hm[1]))
    proj2 = lin('projection2.2', _lrelu(lin('projection2.0', f), 0.1, hm[2]))
    return out, proj, proj2, f


def r1_penalty(d_out_fn, images_aug):
    """r1_loss (train_stylegan2.py:106-113) on already-augmented images; d_out_fn(x) -> logits (default flags)."""
    xa = images_aug.detach().requires_grad_()
    d_real = d_out_fn(xa)
    grad_real, = torch.autograd.grad(outputs=d_real.sum(), inputs=xa, create_graph=True, retain_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


# ----------------------------------------------------------------------------------------------
# Generator (models/gan/stylegan2/generator.py) -- forward only, functional over the state dict
# ----------------------------------------------------------------------------------------------
def g_channels(size, small32, channel_multiplier=2):
    return d_channels(size, small32, channel_multiplier)


def equal_linear(x, w, b, lr_mul=1.0, bias_init=0.0, activation=False):
    """EqualLinear.forward (layers.py:146-153)."""
    scale = (1 / math.sqrt(w.shape[1])) * lr_mul
    bias = b * lr_mul + bias_init
    if activation:
        return fused_leaky_relu(F.linear(x, w * scale), bias)
    return F.linear(x, w * scale, bias=bias)


def mapping(sd, z, n_mlp=8, lr_mlp=0.01):
    """PixelNorm + n_mlp x EqualLinear(fused_lrelu) (generator.py:154-160)."""
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(1, n_mlp + 1):
        x = equal_linear(x, sd['style.%d.weight' % i], sd['style.%d.bias' % i], lr_mul=lr_mlp, activation=True)
    return x


def modulated_conv(sd, pre, x, style, demodulate=True, upsample=False):
    """ModulatedConv2d.forward (generator.py:52-82), grouped-conv form as written."""
    B, cin, H, W = x.shape
    w = sd[pre + '.weight']
    _, cout, _, k, _ = w.shape
    s = equal_linear(style, sd[pre + '.modulation.weight'], sd[pre + '.modulation.bias'], bias_init=1.0)
    scale = 1 / math.sqrt(cin * k * k)
    weight = scale * w * s.view(B, 1, cin, 1, 1)
    if demodulate:
        demod = torch.rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8)
        weight = weight * demod.view(B, cout, 1, 1, 1)
    xin = x.reshape(1, B * cin, H, W)
    if upsample:
        wt = weight.transpose(1, 2).reshape(B * cin, cout, k, k)
        out = F.conv_transpose2d(xin, wt, padding=0, stride=2, groups=B)
        out = out.view(B, cout, out.shape[2], out.shape[3])
        out = upfirdn2d(out, sd[pre + '.blur.kernel'], pad=(1, 1))
    else:
        out = F.conv2d(xin, weight.view(B * cout, cin, k, k), padding=k // 2, groups=B)
        out = out.view(B, cout, out.shape[2], out.shape[3])
    return out


def styled_layer(sd, pre, x, style, noise, upsample=False):
    """StyleLayer.forward (generator.py:120-124)."""
    out = modulated_conv(sd, pre + '.conv', x, style, upsample=upsample)
    out = out + sd[pre + '.noise.weight'] * noise
    return fused_leaky_relu(out, sd[pre + '.activate.bias'])


def to_rgb(sd, pre, x, style, skip=None):
    """ToRGB.forward (generator.py:136-144)."""
    out = modulated_conv(sd, pre + '.conv', x, style, demodulate=False) + sd[pre + '.bias']
    if skip is not None:
        out = out + upfirdn2d(skip, sd[pre + '.upsample.kernel'], up=2, pad=(2, 1))
    return out


def g_forward(sd, z, size, noise, mix=None):
    """Generator.forward in train mode (generator.py:236-291).  noise: list of (B or 1,1,H,W) tensors;
    mix = (z_mix, mix_layer) reproduces the style-mixing branch (:252-266) with explicit randomness."""
    log_size = int(math.log2(size))
    n_latent = log_size * 2 - 2
    latent = mapping(sd, z)
    latents = latent.unsqueeze(1).repeat(1, n_latent, 1)
    if mix is not None:
        latent_mix = mapping(sd, mix[0]).unsqueeze(1)
        mask = (torch.arange(n_latent)[None] < mix[1].unsqueeze(1)).float().unsqueeze(-1)
        latents = latents * mask + latent_mix * (1 - mask)
    B = z.shape[0]
    out = sd['input.const'].repeat(B, 1, 1, 1)
    out = styled_layer(sd, 'conv1', out, latents[:, 0], noise[0])
    skip = to_rgb(sd, 'to_rgb1', out, latents[:, 1])
    idx = 1
    for j in range(log_size - 2):
        out = styled_layer(sd, 'layers.%d' % (2 * j), out, latents[:, idx], noise[1 + 2 * j], upsample=True)
        out = styled_layer(sd, 'layers.%d' % (2 * j + 1), out, latents[:, idx + 1], noise[2 + 2 * j])
        skip = to_rgb(sd, 'to_rgbs.%d' % j, out, latents[:, idx + 2], skip)
        idx += 2
    return 0.5 * skip + 0.5


def det_fill_g(ref_state, seed=777):
    """Deterministic fill keyed on the reference generator's state-dict order (names + shapes passed in)."""
    sd = {}
    for i, (name, shape) in enumerate(ref_state.items()):
        g = torch.Generator().manual_seed(seed + i)
        if name.endswith('kernel'):
            continue
        t = torch.randn(*shape, generator=g)
        if name.endswith('noise.weight'):
            t = t * 0.1
        elif name.endswith('bias'):
            t = t * 0.1
        elif name.startswith('style.'):
            t = t * 100.0 if name.endswith('weight') else t       # EqualLinear(lr_mul=0.01): randn / lr_mul
        sd[name] = t
    return sd


def g_param_shapes(size, small32, channel_multiplier=2, style_dim=512, n_mlp=8):
    """State-dict names/shapes of the reference Generator, in its registration order."""
    ch = g_channels(size, small32, channel_multiplier)
    s = {}
    for i in range(1, n_mlp + 1):
        s['style.%d.weight' % i] = (style_dim, style_dim)
        s['style.%d.bias' % i] = (style_dim,)
    s['input.const'] = (1, ch[4], 4, 4)

    def modconv(pre, cin, cout, k, upsample):
        s[pre + '.weight'] = (1, cout, cin, k, k)
        if upsample:
            s[pre + '.blur.kernel'] = (4, 4)
        s[pre + '.modulation.weight'] = (cin, style_dim)
        s[pre + '.modulation.bias'] = (cin,)

    def styled(pre, cin, cout, upsample):
        modconv(pre + '.conv', cin, cout, 3, upsample)
        s[pre + '.noise.weight'] = (1,)
        s[pre + '.activate.bias'] = (cout,)

    def torgb(pre, cin, upsample):
        s[pre + '.bias'] = (1, 3, 1, 1)            # own parameter first, then sub-modules (state_dict order)
        if upsample:
            s[pre + '.upsample.kernel'] = (4, 4)
        modconv(pre + '.conv', cin, 3, 1, False)

    styled('conv1', ch[4], ch[4], False)
    torgb('to_rgb1', ch[4], False)
    cin = ch[4]
    log_size = int(math.log2(size))
    for j, i in enumerate(range(3, log_size + 1)):
        cout = ch[2 ** i]
        styled('layers.%d' % (2 * j), cin, cout, True)
        styled('layers.%d' % (2 * j + 1), cout, cout, False)
        cin = cout
    cin = ch[4]
    for j, i in enumerate(range(3, log_size + 1)):
        torgb('to_rgbs.%d' % j, ch[2 ** i], True)
    return s


def fill_kernels(sd, shapes):
    for name in shapes:
        if name.endswith('blur.kernel') or name.endswith('upsample.kernel'):
            sd[name] = make_kernel() * 4
    return sd
