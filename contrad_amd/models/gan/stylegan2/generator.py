"""StyleGAN2 generator forward on the HIP library -- counterpart of models/gan/stylegan2/generator.py:146-291.

Under no_grad (the discriminator step's fake batch) it runs on fused forward-only launches; with gradients enabled
(the generator step, train_stylegan2.py:184-194 / train_stylegan2_contraD.py:138-146,207) the same network is composed
from the differentiable HIP nodes of contrad_amd.autograd_ops.  Same state-dict names as the reference.
Design (NHWC feature maps):
  * mapping network: PixelNorm kernel + 8 EqualLinear(lr_mul 0.01)+fused-lrelu as GEMMs with bias/activation in the
    epilogue;
  * ModulatedConv2d is evaluated in its equivalent "modulate the input, demodulate the output" form, so the conv uses
    the SHARED scaled weight on the MFMA engine instead of the reference's grouped conv over B*Cout per-sample filters
    (generator.py:52-82):   y_b = demod_b * conv(x_b * s_b, scale*W),  demod_b[k] = rsqrt(sum_c s_b[c]^2 Wsq[c,k] + 1e-8);
  * upsampling layers: transposed conv stride 2 (= the conv engine's dgrad kernel) followed by the 4x4 blur (upfirdn2d);
  * demodulation + noise injection + bias + leaky-relu*sqrt2 fused in one elementwise pass;
  * ToRGB: 1x1 modulated conv onto 3 channels + bias + upsampled skip in one kernel (rgb_conv_dgrad with per-sample
    channel modulation and a residual input); Upsample of the RGB skip = upfirdn2d(up=2).
Packed weights / Wsq tables are cached and rebuilt only when a parameter's version counter changes (G is frozen during
the D-step).
"""
import math
import os

import torch
import torch.nn as nn

from .... import ops
from .... import autograd_ops as A
from ....hostio import upload


# Switches for the same-box A/B runs of bench.py (tools/dev/r5b.sh … r5d.sh, profiles/r05_ab_g*.txt) and for the test that
# the two forms give the same bits; the defaults are the product path.
# FUSE_TAIL: the forward-only path with the upsampling StyledConv's blur + epilogue in one launch and the next layer's
#   modulation folded into the producer's store (True), or as separate passes (False; bitwise the same images).
# LEGACY_PREP: the round-4 per-layer preparation of the demodulation tables instead of the batched launches.
LEGACY_PREP = os.environ.get('CONTRAD_DEV_G_PREP', '') == 'legacy'
FUSE_TAIL = os.environ.get('CONTRAD_DEV_G_FUSE', '1') != '0'


class _EqualLinearParams(nn.Module):
    """EqualLinear (stylegan2/layers.py:132-154)."""

    def __init__(self, in_dim, out_dim, bias_init=0, lr_mul=1):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul, self.bias_init = lr_mul, bias_init


class _Blur(nn.Module):
    def __init__(self, pad, upsample_factor=1):
        super().__init__()
        k = A.make_blur_kernel((1, 3, 3, 1))
        if upsample_factor > 1:
            k = k * (upsample_factor ** 2)
        self.register_buffer('kernel', k)
        self.pad = pad


class _ModConv(nn.Module):
    """ModulatedConv2d's tensors (generator.py:17-50)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False):
        super().__init__()
        self.in_channel, self.out_channel, self.kernel_size = in_channel, out_channel, kernel_size
        self.upsample, self.demodulate = upsample, demodulate
        if upsample:
            p = (4 - 2) - (kernel_size - 1)
            self.blur = _Blur(((p + 1) // 2 + 2 - 1, p // 2 + 1), upsample_factor=2)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = _EqualLinearParams(style_dim, in_channel, bias_init=1)


class _Noise(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))


class _ActBias(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(c))


class _StyleLayer(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False):
        super().__init__()
        self.conv = _ModConv(in_channel, out_channel, kernel_size, style_dim, upsample=upsample)
        self.noise = _Noise()
        self.activate = _ActBias(out_channel)


class _Upsample(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('kernel', A.make_blur_kernel((1, 3, 3, 1)) * 4)
        self.pad = (2, 1)


class _ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True):
        super().__init__()
        if upsample:
            self.upsample = _Upsample()
        self.conv = _ModConv(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))


class _Const(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.const = nn.Parameter(torch.randn(1, channel, size, size))


class _PixelNorm(nn.Module):
    pass


class Generator(nn.Module):
    def __init__(self, size, style_dim=512, n_mlp=8, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01,
                 small32=False):
        super().__init__()
        if tuple(blur_kernel) != (1, 3, 3, 1):
            raise NotImplementedError('blur kernel [1,3,3,1]')
        self.size, self.style_dim = size, style_dim
        layers = [_PixelNorm()]
        for _ in range(n_mlp):
            layers.append(_EqualLinearParams(style_dim, style_dim, lr_mul=lr_mlp))
        self.style = nn.Sequential(*layers)
        if small32:
            self.channels = {4: 512, 8: 512, 16: 256, 32: 128}
        else:
            self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: int(256 * channel_multiplier),
                             128: int(128 * channel_multiplier), 256: int(64 * channel_multiplier),
                             512: int(32 * channel_multiplier), 1024: int(16 * channel_multiplier)}
        self.input = _Const(self.channels[4])
        self.conv1 = _StyleLayer(self.channels[4], self.channels[4], 3, style_dim)
        self.to_rgb1 = _ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.layers = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.layers.append(_StyleLayer(in_channel, out_channel, 3, style_dim, upsample=True))
            self.layers.append(_StyleLayer(out_channel, out_channel, 3, style_dim))
            self.to_rgbs.append(_ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        self._cache_key, self._cache = None, None
        self._gather_key, self._gather_idx = None, None      # flat gather index of _all_styles (per batch size)

    @property
    def device(self):
        return self.input.const.device

    def sample_latent(self, num_samples):
        return torch.randn(num_samples, self.style_dim, device=self.device)

    def input_sizes(self, N):
        """Element counts of everything a training-mode forward draws on the device: z, the mixing latent, per-layer noise."""
        return [N * self.style_dim, N * self.style_dim] + [N * 4 ** ((i + 5) // 2) for i in range(self.num_layers)]

    def input_views(self, flat, N):
        parts = torch.split(flat, self.input_sizes(N))
        z, z_mix = parts[0].view(N, self.style_dim), parts[1].view(N, self.style_dim)
        noise = [parts[2 + i].view(N, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2)) for i in range(self.num_layers)]
        return z, z_mix, noise

    def draw_inputs(self, N, style_mix, flat=None):
        """Everything random in one training-mode forward -- ``G(G.sample_latent(N), style_mix)`` draws z, then inside
        forward() the mixing latent, the mixing layers (CPU generator) and one noise map per layer (generator.py:91-92,
        233-234, 252-259) -- with ONE device launch: a single flat normal_() cut into views (the training loops' D- and
        G-steps and the hipGraph replays all sample through here, so eager and replayed steps consume the same stream).
        Returns kwargs for forward(): ``G(**G.draw_inputs(N, style_mix))``."""
        if flat is None:
            flat = torch.empty(sum(self.input_sizes(N)), device=self.device)
        flat.normal_()
        z, z_mix, noise = self.input_views(flat, N)
        kw = dict(input=z, style_mix=style_mix, noise=noise)
        if self.training and style_mix > 0:
            nomix_mask = torch.rand(N) >= style_mix
            mix_layer = torch.randint(self.n_latent, (N,)).masked_fill(nomix_mask, self.n_latent)
            kw['_mix'] = (z_mix, mix_layer)
        return kw

    def _layer_idx(self, device):
        t = getattr(self, '_layer_idx_cache', None)
        if t is None or t.device != device:
            t = self._layer_idx_cache = torch.arange(self.n_latent, device=device, dtype=torch.float32)[None]
        return t

    # ---- cached weight preparation -----------------------------------------------------------------------
    def _modconvs(self):
        out = [self.conv1.conv, self.to_rgb1.conv]
        for l in self.layers:
            out.append(l.conv)
        for t in self.to_rgbs:
            out.append(t.conv)
        return out

    def invalidate_cache(self):
        """Forget the packed tables (see G_SNDCGAN.invalidate_cache)."""
        self._cache_key, self._cache = None, None

    def _prepared(self, differentiable=False):
        """Packed weights / biases / Wsq tables.  Forward-only: built under no_grad and cached on the parameters'
        version counters.  ``differentiable``: rebuilt inside the autograd graph on every call."""
        if not differentiable:
            params = list(self.parameters())
            key = tuple((p.data_ptr(), p._version) for p in params)
            if key == self._cache_key:
                return self._cache
        dev = self.device
        ws, entries, groups = [], [], []

        def add(w, K, C, T, scale):
            groups.append((T * C, ops.round_up(K, 4)))
            ws.append(w)
            entries.append((K, C, T, scale, len(groups) - 1, 0))
            return len(groups) - 1

        c = {'style': [], 'mod': {}, 'conv': {}, 'wsq': {}, 'differentiable': differentiable}
        if not differentiable and not LEGACY_PREP:
            with torch.no_grad():
                self._prepare_forward_only(c, add, ws, entries, groups, dev)
            self._cache_key, self._cache = key, c
            return c
        if not differentiable:          # dev A/B only: the per-layer ATen construction of rounds 1 - 4 under no_grad
            with torch.no_grad():
                self._prepare_legacy(c, add, ws, entries, groups, dev)
            self._cache_key, self._cache = key, c
            return c
        self._prepare_per_layer(c, add)
        c['packed'] = A.PackWeightsFn.apply(A.PackMeta(entries, groups), *ws)
        return c

    def _prepare_per_layer(self, c, add):
        for m in list(self.style)[1:]:
            c['style'].append((add(m.weight, m.weight.shape[0], m.weight.shape[1], 1, m.scale),
                               (m.bias * m.lr_mul).contiguous()))
        for mc in self._modconvs():
            mod = mc.modulation
            c['mod'][mc] = (add(mod.weight, mod.weight.shape[0], mod.weight.shape[1], 1, mod.scale),
                            (mod.bias * mod.lr_mul + mod.bias_init).contiguous())
            w = mc.weight[0]                                  # (Cout, Cin, k, k)
            k = mc.kernel_size
            if mc.upsample or mc.out_channel == 3:            # transposed conv / ToRGB == dgrad of the
                #                                               (K = Cin, C = Cout) conv: pack rows (tap, cout), cols cin
                c['conv'][mc] = add(w.transpose(0, 1).contiguous(), mc.in_channel, mc.out_channel, k * k, mc.scale)
            else:
                c['conv'][mc] = add(w.contiguous(), mc.out_channel, mc.in_channel, k * k, mc.scale)
            if mc.demodulate:                                 # Wsq[c][k] = scale^2 * sum_taps W[k,c,:,:]^2
                c['wsq'][mc] = ((w * mc.scale).pow(2).sum((2, 3)).t().contiguous())

    def _prepare_legacy(self, c, add, ws, entries, groups, dev):
        self._prepare_per_layer(c, add)
        mcs = self._modconvs()
        tot = sum(mc.in_channel for mc in mcs)
        groups.append((self.style_dim, ops.round_up(tot, 4)))
        gi, off, c['mod_cols'] = len(groups) - 1, 0, {}
        for mc in mcs:
            mod = mc.modulation
            ws.append(mod.weight)
            entries.append((mc.in_channel, self.style_dim, 1, mod.scale, gi, off))
            c['mod_cols'][mc] = (off, mc.in_channel)
            off += mc.in_channel
        c['mod_all'] = gi
        pad = ops.round_up(tot, 4) - tot
        c['mod_bias_all'] = torch.cat([c['mod'][mc][1] for mc in mcs] + ([torch.zeros(pad, device=dev)] if pad else []))
        c['packed'] = A.PackWeightsFn.apply(A.PackMeta(entries, groups), *ws)

    def _prepare_forward_only(self, c, add, ws, entries, groups, dev):
        """The tables of the no-grad forward in a handful of launches (G moves between two discriminator steps, so a captured
        D-step rebuilds them on every replay): every bias of the mapping network and of the modulation linears from ONE
        concatenation + ONE multiply-add against cached constant vectors (was: 1 - 2 elementwise launches per layer); the
        mapping weights and all modulation linears side by side -- ONE [style_dim][sum C_in] matrix, so every layer's
        style vector comes from a single GEMM over the (B * n_latent) latent rows -- through the batched pack kernel; the
        shared conv weights in their GEMM layouts and the demodulation tables Wsq from ONE launch
        (contrad_modconv_tables; was: a transposed copy per upsampling / ToRGB layer and mul, pow, sum, transpose-copy
        per demodulated layer: ~130 launches at 512^2, ~65 at 32^2)."""
        mcs = self._modconvs()
        styles = list(self.style)[1:]
        # -- biases
        raw = [m.bias for m in styles] + [mc.modulation.bias for mc in mcs]
        sizes = [b.numel() for b in raw]
        tot = sum(mc.in_channel for mc in mcs)
        pad = ops.round_up(tot, 4) - tot
        kc = (tuple(sizes), pad, str(dev))
        if getattr(self, '_bias_consts_key', None) != kc:
            mul = torch.cat([torch.full((m.bias.numel(),), float(m.lr_mul)) for m in styles] +
                            [torch.full((mc.modulation.bias.numel(),), float(mc.modulation.lr_mul)) for mc in mcs] +
                            [torch.zeros(pad)])
            addv = torch.cat([torch.zeros(m.bias.numel()) for m in styles] +
                             [torch.full((mc.modulation.bias.numel(),), float(mc.modulation.bias_init)) for mc in mcs] +
                             [torch.zeros(pad)])
            self._bias_consts, self._bias_consts_key = (mul.to(dev), addv.to(dev)), kc
        mul, addv = self._bias_consts
        flat = torch.cat(raw + ([addv[:pad]] if pad else []))
        biases = torch.addcmul(addv, flat, mul)                     # bias * lr_mul + bias_init, all layers at once
        off = 0
        for m in styles:
            n = m.bias.numel()
            c['style'].append((add(m.weight, m.weight.shape[0], m.weight.shape[1], 1, m.scale), biases[off:off + n]))
            off += n
        c['mod_bias_all'] = biases[off:]
        # -- all modulation linears as one packed matrix
        groups.append((self.style_dim, ops.round_up(tot, 4)))
        gi, off, c['mod_cols'] = len(groups) - 1, 0, {}
        for mc in mcs:
            mod = mc.modulation
            ws.append(mod.weight)
            entries.append((mc.in_channel, self.style_dim, 1, mod.scale, gi, off))
            c['mod_cols'][mc] = (off, mc.in_channel)
            off += mc.in_channel
        c['mod_all'] = gi
        packed = list(A.PackWeightsFn.apply(A.PackMeta(entries, groups), *ws))
        # -- shared conv weights + demodulation tables: one flat buffer each, one launch
        shapes, wsq_n = [], 0
        for mc in mcs:
            T = mc.kernel_size ** 2
            tr = bool(mc.upsample or mc.out_channel == 3)
            rows, cols = (T * mc.out_channel, mc.in_channel) if tr else (T * mc.in_channel, mc.out_channel)
            shapes.append((rows, ops.round_up(cols, 4), cols, tr))
            wsq_n += mc.in_channel * mc.out_channel if mc.demodulate else 0
        if any(ld != cols for _, ld, cols, _ in shapes):
            raise NotImplementedError('modulated conv with a channel count that is not a multiple of 4')
        wflat = torch.empty(sum(r * ld for r, ld, _, _ in shapes), device=dev, dtype=torch.float32)
        qflat = torch.empty(max(wsq_n, 4), device=dev, dtype=torch.float32)
        layers, wo, qo = [], 0, 0
        for mc, (rows, ld, cols, tr) in zip(mcs, shapes):
            wp = wflat[wo:wo + rows * ld].view(rows, ld)
            wo += rows * ld
            wsq = None
            if mc.demodulate:
                wsq = qflat[qo:qo + mc.in_channel * mc.out_channel].view(mc.in_channel, mc.out_channel)
                qo += mc.in_channel * mc.out_channel
                c['wsq'][mc] = wsq
            packed.append(wp)
            c['conv'][mc] = len(packed) - 1
            layers.append((mc.weight[0], wp, wsq, tr, mc.scale))
        ops.modconv_tables(layers)
        c['packed'] = packed

    def _all_styles(self, latents, c):
        """Forward-only path: ({modconv: (B, C_in) style}, {modconv: (B, C_out) demodulation factor}).  One GEMM over the
        B * n_latent latent rows gives every layer's style; ONE gather (a cached flat index) lays the row block of each
        layer's latent index out contiguously, layer after layer (was: one strided-slice copy per layer); the
        demodulation factors rsqrt(sum_c s^2 Wsq + 1e-8) (generator.py:62-64) of all layers come from ONE launch
        (contrad_modconv_demod; rounds 2 - 4: a squaring pass, a split-K GEMM + reduce per layer, one add + rsqrt)."""
        B, L, D = latents.shape
        wp = c['packed'][c['mod_all']]
        cols = wp.shape[1]
        out = ops.conv2d_fwd(latents.contiguous().view(B * L, 1, 1, D), wp, c['mod_bias_all'], cols, 1, 1, 1, 0)
        lat_idx = {self.conv1.conv: 0, self.to_rgb1.conv: 1}
        idx = 1
        for j in range(len(self.to_rgbs)):
            lat_idx[self.layers[2 * j].conv] = idx
            lat_idx[self.layers[2 * j + 1].conv] = idx + 1
            lat_idx[self.to_rgbs[j].conv] = idx + 2
            idx += 2
        order = list(c['mod_cols'].items())
        key = (B, L, cols, str(out.device))
        if self._gather_key != key:
            if torch.cuda.is_current_stream_capturing():
                # the index is a pageable host -> device copy and would live in the graph's private pool: it has to exist
                # before the capture (one eager forward at this batch size; the Graphed*Step constructors run one)
                raise RuntimeError('Generator: first forward at batch size %d inside a stream capture; run one eager '
                                   'forward at this batch size before capturing' % B)
            rows = torch.arange(B, dtype=torch.int64).view(B, 1)
            pieces = [((rows * L + lat_idx[mc]) * cols + o + torch.arange(n, dtype=torch.int64).view(1, n)).reshape(-1)
                      for mc, (o, n) in order]
            self._gather_idx, self._gather_key = torch.cat(pieces).to(out.device), key
        flat = out.view(-1).index_select(0, self._gather_idx)
        styles, off = {}, 0
        for mc, (o, n) in order:
            styles[mc] = flat[off:off + B * n].view(B, n)
            off += B * n
        dem = [mc for mc, _ in order if mc.demodulate]
        dbuf = torch.empty(B * sum(mc.out_channel for mc in dem), device=out.device, dtype=torch.float32)
        demods, doff = {}, 0
        if LEGACY_PREP:       # (dev A/B: squaring pass + one split-K GEMM and reduce per layer + add + rsqrt)
            sq = flat * flat
            offs, off = {}, 0
            for mc, (o, n) in order:
                offs[mc] = off
                off += B * n
        jobs = []
        for mc in dem:
            n_in, K = c['mod_cols'][mc][1], mc.out_channel
            d = dbuf[doff:doff + B * K].view(B, K)
            if LEGACY_PREP:
                ops.conv2d_fwd(sq[offs[mc]:offs[mc] + B * n_in].view(B, 1, 1, n_in), c['wsq'][mc], None, K, 1, 1, 1, 0,
                               out=d.view(B, 1, 1, K))
            else:
                jobs.append((styles[mc], c['wsq'][mc], d))
            demods[mc] = d
            doff += B * K
        if LEGACY_PREP:
            torch.rsqrt_(dbuf.add_(1e-8))
        else:
            ops.modconv_demod(jobs, 1e-8)       # every layer's rsqrt(s^2 @ Wsq + 1e-8) in one launch
        return styles, demods

    # ---- pieces ----------------------------------------------------------------------------------------------
    def _mapping(self, z, c):
        x = ops.pixelnorm(z.contiguous().float())          # z is a sample, not a parameter: no gradient needed
        B = x.shape[0]
        for gi, bias in c['style']:
            if c['differentiable']:
                x = A.ConvBiasActFn.apply(x.view(B, 1, 1, -1), c['packed'][gi], bias, (self.style_dim, 1, 1, 1, 0),
                                          0.2, math.sqrt(2.0)).view(B, -1)
            else:
                x = ops.conv2d_fwd(x.view(B, 1, 1, -1), c['packed'][gi], bias, self.style_dim, 1, 1, 1, 0, 0.2,
                                   math.sqrt(2.0)).view(B, -1)
        return x

    def _style(self, mc, w_lat, c):
        B = w_lat.shape[0]
        if mc not in c['mod']:
            # forward-only cache (_prepare_forward_only): the modulation linears exist only side by side in ONE matrix; a
            # caller that did not take its style from _all_styles gets this layer's columns of the full product
            off, n = c['mod_cols'][mc]
            wp = c['packed'][c['mod_all']]
            full = ops.conv2d_fwd(w_lat.contiguous().view(B, 1, 1, -1), wp, c['mod_bias_all'], wp.shape[1], 1, 1, 1, 0)
            return full.view(B, -1)[:, off:off + n].contiguous()
        gi, bias = c['mod'][mc]
        if c['differentiable']:
            return A.ConvBiasActFn.apply(w_lat.contiguous().view(B, 1, 1, -1), c['packed'][gi], bias,
                                         (mc.in_channel, 1, 1, 1, 0), 1.0, 1.0).view(B, mc.in_channel)
        return ops.conv2d_fwd(w_lat.contiguous().view(B, 1, 1, -1), c['packed'][gi], bias, mc.in_channel, 1, 1, 1,
                              0).view(B, mc.in_channel)

    def _styled_conv_grad(self, layer, x, w_lat, noise, c):
        """StyledConv (generator.py:97-118) out of differentiable nodes: style -> demodulation factors -> modulate the
        input -> shared-weight conv (or transposed conv + blur) -> demod + noise + bias + activation."""
        mc = layer.conv
        B, H, W, _ = x.shape
        s = self._style(mc, w_lat, c)
        d = A.Conv2dFn.apply((s * s).view(B, 1, 1, -1), c['wsq'][mc], (mc.out_channel, 1, 1, 1, 0)).view(B, -1)
        demod = torch.rsqrt(d + 1e-8)
        xm = A.NhwcScaleFn.apply(x, s)
        wp = c['packed'][c['conv'][mc]]
        if mc.upsample:
            y = A.ConvDgradFn.apply(xm, wp, (B, 2 * H + 1, 2 * W + 1, mc.out_channel), (mc.in_channel, 3, 3, 2, 0))
            p0, p1 = mc.blur.pad
            y = A.UpFirDn2dFn.apply(y, mc.blur.kernel, 1, 1, (p0, p1, p0, p1))
        else:
            y = A.Conv2dFn.apply(xm, wp, (mc.out_channel, 3, 3, 1, 1))
        if noise is None:
            noise = torch.empty(B, 1, y.shape[1], y.shape[2], device=y.device).normal_()
        noise = noise.expand(B, 1, y.shape[1], y.shape[2]).contiguous()
        return A.ModconvEpilogueFn.apply(y, demod, noise, layer.noise.weight, layer.activate.bias)

    def _to_rgb_grad(self, trgb, x, w_lat, skip, c):
        """ToRGB (generator.py:121-143): 1x1 modulated conv without demodulation + bias + upsampled skip."""
        mc = trgb.conv
        s = self._style(mc, w_lat, c)
        xm = A.NhwcScaleFn.apply(x, s)
        out = A.RgbDgradFn.apply(xm, c['packed'][c['conv'][mc]], 3, (1, 1.0)) + trgb.bias
        if skip is not None:
            B, C, H, W = skip.shape
            p0, p1 = trgb.upsample.pad
            up = A.UpFirDn2dFn.apply(skip.reshape(B * C, H, W, 1), trgb.upsample.kernel, 2, 1, (p0, p1, p0, p1))
            out = out + up.view(B, C, 2 * H, 2 * W)
        return out

    def _styled_conv(self, layer, x, w_lat, noise, c, s=None, demod=None, post=None, prescaled=False):
        """Forward-only StyledConv.  ``post``: the style vector of the ONE layer that consumes the output -- its weight
        modulation rides on this layer's epilogue store (the upsampling conv of a resolution feeds only that resolution's
        second conv; outputs that also feed a ToRGB stay unscaled); ``prescaled``: x arrived already modulated that way.
        The upsampling layer's blur and its epilogue are one launch (ops.upfirdn2d_modconv): per resolution two passes
        over the activation less than blur -> epilogue -> nhwc_scale (FUSE_TAIL = False restores them: same bits)."""
        mc = layer.conv
        B, H, W, _ = x.shape
        if s is None:
            s = self._style(mc, w_lat, c)
        if mc.demodulate and demod is None:
            wsq = c['wsq'][mc]
            d = ops.conv2d_fwd((s * s).view(B, 1, 1, -1), wsq, None, mc.out_channel, 1, 1, 1, 0).view(B, -1)
            demod = torch.rsqrt(d + 1e-8)
        xm = x if prescaled else ops.nhwc_scale(x, s)
        wp = c['packed'][c['conv'][mc]]
        oh, ow = (2 * H, 2 * W) if mc.upsample else (H, W)
        if noise is None:
            noise = torch.empty(B, 1, oh, ow, device=x.device).normal_()                   # generator.py:91-92
        noise = noise.expand(B, 1, oh, ow).contiguous()
        if mc.upsample:
            y = ops.conv2d_dgrad(xm, wp, (B, 2 * H + 1, 2 * W + 1, mc.out_channel), 3, 3, 2, 0)
            p0, p1 = mc.blur.pad
            if FUSE_TAIL and mc.out_channel % 4 == 0:
                return ops.upfirdn2d_modconv(y, mc.blur.kernel, (p0, p1, p0, p1), demod, noise, layer.noise.weight,
                                             layer.activate.bias, post)
            y = ops.upfirdn2d(y, mc.blur.kernel, 1, 1, (p0, p1, p0, p1))
        else:
            y = ops.conv2d_fwd(xm, wp, None, mc.out_channel, 3, 3, 1, 1)
        if post is not None and not FUSE_TAIL:
            return ops.nhwc_scale(ops.modconv_epilogue_(y, demod, noise, layer.noise.weight, layer.activate.bias), post)
        return ops.modconv_epilogue_(y, demod, noise, layer.noise.weight, layer.activate.bias, post_scale=post)

    def _to_rgb(self, trgb, x, w_lat, skip, c, final=False, s=None):
        mc = trgb.conv
        if s is None:
            s = self._style(mc, w_lat, c)
        res = None
        if skip is not None:
            B, C, H, W = skip.shape
            p0, p1 = trgb.upsample.pad
            res = ops.upfirdn2d(skip.reshape(B * C, H, W, 1), trgb.upsample.kernel, 2, 1, (p0, p1, p0, p1))
            res = res.view(B, C, 2 * H, 2 * W)
        return ops.rgb_conv_dgrad(x, c['packed'][c['conv'][mc]], trgb.bias.view(-1), 3, 1, act=0,
                                  out_scale=0.5 if final else 1.0, out_shift=0.5 if final else 0.0,
                                  mod=s, residual=res)

    def forward(self, input, return_latents=False, style_mix=0.9, input_is_latent=False, noise=None, _mix=None):
        if not input.is_cuda:
            raise RuntimeError('contrad_amd StyleGAN2 generator runs on the MI355X HIP path only (no CPU fallback)')
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        c = self._prepared(differentiable=grad)
        mixing = self.training and style_mix > 0
        z_mix = mix_layer = None
        if mixing:                                                 # generator.py:252-266 (masks from the CPU generator)
            B = input.size(0)
            if _mix is None:
                z_mix = self.sample_latent(B)
                nomix_mask = torch.rand(B) >= style_mix
                mix_layer = torch.randint(self.n_latent, (B,))
                mix_layer = mix_layer.masked_fill(nomix_mask, self.n_latent)
            else:
                z_mix, mix_layer = _mix
        if mixing and not input_is_latent:
            # both latent batches through the mapping network in ONE pass of 2B rows (8 GEMM launches instead of 16)
            both = self._mapping(torch.cat([input, z_mix], dim=0), c)
            latent, latent_mix = both[:input.size(0)], both[input.size(0):]
        else:
            latent = self._mapping(input, c) if not input_is_latent else input
            latent_mix = self._mapping(z_mix, c) if mixing else None
        if noise is None:
            noise = [None] * self.num_layers
        latents = latent.unsqueeze(1).repeat(1, self.n_latent, 1) if latent.ndim < 3 else latent
        if mixing:
            if not mix_layer.is_cuda:       # CPU draw -> device without a host-side wait (contrad_amd/hostio.py)
                mix_layer = upload(mix_layer.float().view(-1, 1), latents.device).view(-1)
            layer_idx = self._layer_idx(latents.device)
            mask = (layer_idx < mix_layer.float().unsqueeze(1)).float().unsqueeze(-1)
            latents = latents * mask + latent_mix.unsqueeze(1) * (1 - mask)
        B = latents.shape[0]
        x = self.input.const.permute(0, 2, 3, 1).expand(B, -1, -1, -1).contiguous()        # NHWC const input
        if grad:
            x = self._styled_conv_grad(self.conv1, x, latents[:, 0], noise[0], c)
            skip = self._to_rgb_grad(self.to_rgb1, x, latents[:, 1], None, c)
            idx = 1
            for j in range(len(self.to_rgbs)):
                x = self._styled_conv_grad(self.layers[2 * j], x, latents[:, idx], noise[1 + 2 * j], c)
                x = self._styled_conv_grad(self.layers[2 * j + 1], x, latents[:, idx + 1], noise[2 + 2 * j], c)
                skip = self._to_rgb_grad(self.to_rgbs[j], x, latents[:, idx + 2], skip, c)
                idx += 2
            image = 0.5 * skip + 0.5                                                        # generator.py:283
            if return_latents:
                return image, latents
            return image
        st, dm = self._all_styles(latents, c)
        x = self._styled_conv(self.conv1, x, None, noise[0], c, s=st[self.conv1.conv], demod=dm.get(self.conv1.conv))
        last = len(self.to_rgbs) == 0
        skip = self._to_rgb(self.to_rgb1, x, None, None, c, final=last, s=st[self.to_rgb1.conv])
        for j in range(len(self.to_rgbs)):
            la, lb, tr = self.layers[2 * j], self.layers[2 * j + 1], self.to_rgbs[j]
            # la's output feeds lb only: lb's weight modulation is la's post-scale, lb takes its input as it comes
            x = self._styled_conv(la, x, None, noise[1 + 2 * j], c, s=st[la.conv], demod=dm.get(la.conv),
                                  post=st[lb.conv])
            x = self._styled_conv(lb, x, None, noise[2 + 2 * j], c, s=st[lb.conv], demod=dm.get(lb.conv),
                                  prescaled=True)
            skip = self._to_rgb(tr, x, None, skip, c, final=(j == len(self.to_rgbs) - 1), s=st[tr.conv])
        image = skip                                                                        # 0.5*x+0.5 fused above
        if not self.training:
            image = image.clamp(0, 1)
        if return_latents:
            return image, latents
        return image
