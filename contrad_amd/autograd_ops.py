"""Op-level autograd nodes over the HIP library that are differentiable to ANY order: the backward of every node
is built only from other nodes of this file, exactly the scheme the reference uses for its native upfirdn2d op
(models/gan/stylegan2/op/upfirdn2d.py:19-142) -- needed because the R1 penalty (train_stylegan2.py:106-113)
differentiates ``||dD/dx||^2``, i.e. runs a backward through the backward.

The convolution family is closed under differentiation (each op is bilinear in its two tensor arguments):

    Conv2dFn(x, wp)      -> y        d/dx: ConvDgradFn(gy, wp)     d/dwp: ConvWgradFn(x, gy)
    ConvDgradFn(gy, wp)  -> dx       d/dgy: Conv2dFn(h, wp)        d/dwp: ConvWgradFn(h, gy)
    ConvWgradFn(x, gy)   -> dwp      d/dx: ConvDgradFn(gy, h)      d/dgy: Conv2dFn(x, h)

and so is the RGB family (RgbConvFn / RgbDgradFn / RgbWgradFn).  Activations are NHWC; ``wp`` is the packed GEMM
weight produced by PackWeightsFn (whose backward is UnpackWeightsFn and vice versa).
The SNDCGAN discriminator does NOT use these (it is one fused node, models/gan/sndcgan.py); the StyleGAN2
discriminator does.
"""
import math

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


def _cont(t):
    return t if t.is_contiguous() else t.contiguous()


# ``input_grad_only()``: inside it, the backward of every conv / RGB node of this family returns the gradient w.r.t. its
# ACTIVATION input only.  A custom Function cannot see which of its input gradients an ``autograd.grad(outputs, inputs)``
# call actually asked for (``ctx.needs_input_grad`` is static), so the first backward of the R1 penalty --
# ``autograd.grad(d_real.sum(), images, create_graph=True)``, train_stylegan2.py:108-111 -- also computed every weight
# and bias gradient of the discriminator and threw them away: 17 weight-gradient GEMMs + their split-K reduces per
# StyleGAN2-32 step.  engine.r1_loss wraps exactly that call.  (A module global, not a thread-local: the backward runs
# on the autograd engine's device thread while the caller blocks in autograd.grad.)
_INPUT_GRAD_ONLY = False


class input_grad_only(object):
    def __enter__(self):
        global _INPUT_GRAD_ONLY
        self.saved, _INPUT_GRAD_ONLY = _INPUT_GRAD_ONLY, True
        return self

    def __exit__(self, *exc):
        global _INPUT_GRAD_ONLY
        _INPUT_GRAD_ONLY = self.saved
        return False


_CONSTS = {}


def _device_constant(values, device):
    """Small read-only float tensor on the device, built once per (values, device): a host -> device copy inside a
    captured hipGraph is not permitted (and costs a synchronisation outside one)."""
    key = (values, str(device))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(values, device=device, dtype=torch.float32)
    return t


def _packed_buffer(shape, k_written, device):
    """Output buffer in the packed [(tap, c)][ldw] layout: the kernels write columns [0, k_written) of every row, so a
    zero fill is needed only when padding columns exist (ldw > K) -- ~110 fill launches per StyleGAN2-32 step otherwise."""
    if shape[1] == k_written:
        return torch.empty(shape, device=device, dtype=torch.float32)
    return torch.zeros(shape, device=device, dtype=torch.float32)


# ----------------------------------------------------------------------------------------------------------
# convolution family.  geom = (K, KH, KW, stride, pad)
# ----------------------------------------------------------------------------------------------------------
class Conv2dFn(Function):
    @staticmethod
    def forward(ctx, x, wp, geom):
        K, KH, KW, s, p = geom
        x = _cont(x)
        ctx.save_for_backward(x, wp)
        ctx.geom = geom
        return ops.conv2d_fwd(x, wp, None, K, KH, KW, s, p)

    @staticmethod
    def backward(ctx, gy):
        x, wp = ctx.saved_tensors
        gx = ConvDgradFn.apply(gy, wp, tuple(x.shape), ctx.geom) if ctx.needs_input_grad[0] else None
        gw = ConvWgradFn.apply(x, gy, ctx.geom, tuple(wp.shape)) if (ctx.needs_input_grad[1]
                                                                      and not _INPUT_GRAD_ONLY) else None
        return gx, gw, None


class ConvDgradFn(Function):
    @staticmethod
    def forward(ctx, gy, wp, x_shape, geom):
        K, KH, KW, s, p = geom
        gy = _cont(gy)
        ctx.save_for_backward(gy, wp)
        ctx.geom, ctx.x_shape = geom, x_shape
        return ops.conv2d_dgrad(gy, wp, x_shape, KH, KW, s, p)

    @staticmethod
    def backward(ctx, h):
        gy, wp = ctx.saved_tensors
        g_gy = Conv2dFn.apply(h, wp, ctx.geom) if ctx.needs_input_grad[0] else None
        g_wp = ConvWgradFn.apply(h, gy, ctx.geom, tuple(wp.shape)) if ctx.needs_input_grad[1] else None
        return g_gy, g_wp, None, None


class ConvWgradFn(Function):
    @staticmethod
    def forward(ctx, x, gy, geom, wp_shape):
        K, KH, KW, s, p = geom
        x, gy = _cont(x), _cont(gy)
        ctx.save_for_backward(x, gy)
        ctx.geom = geom
        out = _packed_buffer(wp_shape, K, x.device)   # padding columns (ldw > K) stay 0
        return ops.conv2d_wgrad(x, gy, KH, KW, s, p, out=out)

    @staticmethod
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        h = _cont(h)
        g_x = ConvDgradFn.apply(gy, h, tuple(x.shape), ctx.geom) if ctx.needs_input_grad[0] else None
        g_gy = Conv2dFn.apply(x, h, ctx.geom) if ctx.needs_input_grad[1] else None
        return g_x, g_gy, None, None


class ConvWgradBiasFn(Function):
    """(dwp, dbias) = (wgrad(x, gy), sum_{n,h,w} gy) in ONE kernel: the bias gradient is accumulated from the gy
    tiles the weight-gradient kernel streams anyway.  Both outputs are linear in gy, so the backward is again made
    of the family's nodes."""

    @staticmethod
    def forward(ctx, x, gy, geom, wp_shape):
        K, KH, KW, s, p = geom
        x, gy = _cont(x), _cont(gy)
        ctx.save_for_backward(x, gy)
        ctx.geom = geom
        out = _packed_buffer(wp_shape, K, x.device)
        db = torch.empty(K, device=x.device, dtype=torch.float32)
        ops.conv2d_wgrad(x, gy, KH, KW, s, p, out=out, dbias=db)
        return out, db

    @staticmethod
    def backward(ctx, h, hb):
        x, gy = ctx.saved_tensors
        g_x = g_gy = None
        if h is not None:
            h = _cont(h)
            g_x = ConvDgradFn.apply(gy, h, tuple(x.shape), ctx.geom) if ctx.needs_input_grad[0] else None
            g_gy = Conv2dFn.apply(x, h, ctx.geom) if ctx.needs_input_grad[1] else None
        if hb is not None and ctx.needs_input_grad[1]:
            e = hb.view((1,) * (gy.dim() - 1) + (-1,)).expand(gy.shape)
            g_gy = e if g_gy is None else g_gy + e
        return g_x, g_gy, None, None


def _fused_bias_ok(x, K):
    return x.shape[-1] % 4 == 0 and K % 4 == 0


class ActBwdFn(Function):
    """g_pre = g_post * lrelu'(.) * gain, with the linear region read from the activation OUTPUT ``y``
    (FusedLeakyReLUFunctionBackward, op/fused_act.py:20-55: linear in g, its own backward is the same op)."""

    @staticmethod
    def forward(ctx, g, y, slope, gain):
        ctx.save_for_backward(y)
        ctx.cfg = (slope, gain)
        return ops.fused_bias_act(_cont(g), None, y, 3, 1, slope, gain)

    @staticmethod
    def backward(ctx, h):
        y, = ctx.saved_tensors
        slope, gain = ctx.cfg
        return ActBwdFn.apply(h, y, slope, gain), None, None, None


class ActFn(Function):
    """y = gain * lrelu_slope(x) (F.leaky_relu after a residual add, models/gan/snresnet.py:40); any-order."""

    @staticmethod
    def forward(ctx, x, slope, gain):
        y = ops.fused_bias_act(_cont(x), None, None, 3, 0, slope, gain)
        ctx.save_for_backward(y)
        ctx.cfg = (slope, gain)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        slope, gain = ctx.cfg
        return ActBwdFn.apply(g, y, slope, gain), None, None


class ColSumFn(Function):
    """Bias gradient: sum over all leading dims of an (..., K) tensor."""

    @staticmethod
    def forward(ctx, g):
        g = _cont(g)
        ctx.shape = tuple(g.shape)
        return ops.colstats(g.view(-1, g.shape[-1]))[0]

    @staticmethod
    def backward(ctx, h):
        return h.view((1,) * (len(ctx.shape) - 1) + (-1,)).expand(ctx.shape)


class ConvBiasActFn(Function):
    """y = gain * lrelu_slope(conv(x, wp) + bias): conv + FusedLeakyReLU (stylegan2/layers.py:174-198,
    op/fused_act.py:74-92) / nn.Linear + LeakyReLU in ONE kernel (bias + activation in the GEMM epilogue).
    Optional 7th argument ``xch = (comm, meta, group)``: in a first-order backward the packed weight gradient is
    all-reduced the moment it is produced (data-parallel exchange overlapped with the backward, PackWeightsFn.backward)."""

    @staticmethod
    def forward(ctx, x, wp, bias, geom, slope, gain, xch=None):
        K, KH, KW, s, p = geom
        x = _cont(x)
        y = ops.conv2d_fwd(x, wp, bias, K, KH, KW, s, p, slope, gain)
        ctx.save_for_backward(x, wp, y)
        ctx.cfg = (geom, slope, gain)
        ctx.xch = xch
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wp, y = ctx.saved_tensors
        geom, slope, gain = ctx.cfg
        g_pre = gy if (slope == 1.0 and gain == 1.0) else ActBwdFn.apply(gy, y, slope, gain)
        gx = ConvDgradFn.apply(g_pre, wp, tuple(x.shape), geom) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if _INPUT_GRAD_ONLY:
            pass
        elif ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and _fused_bias_ok(x, geom[0]):
            gw, gb = ConvWgradBiasFn.apply(x, g_pre, geom, tuple(wp.shape))
        else:
            gw = ConvWgradFn.apply(x, g_pre, geom, tuple(wp.shape)) if ctx.needs_input_grad[1] else None
            gb = ColSumFn.apply(g_pre) if ctx.needs_input_grad[2] else None
        if ctx.xch is not None and gw is not None and not torch.is_grad_enabled():
            exchange_packed(ctx.xch, gw)
        return (gx, gw, gb, None, None, None) + ((None,) if ctx.xch is not None else ())


def exchange_packed(xch, gw):
    """All-reduce a packed weight gradient at its production site and tell its PackWeightsFn node (which waits for the
    collective before it unpacks, and reduces whatever nobody reduced before)."""
    comm, meta, group = xch
    if meta.shared:             # other calls contribute to the same packed gradient: the pack node reduces the total
        return
    comm.reduce_async(gw)
    meta.reduced.add(group)


# ----------------------------------------------------------------------------------------------------------
# RGB family (image NCHW (N,3,H,W) <-> activation NHWC (N,H,W,K)); rgb = (k, in_scale, in_shift)
# ----------------------------------------------------------------------------------------------------------
class RgbConvFn(Function):
    @staticmethod
    def forward(ctx, img, wp, K, rgb):
        k, a, b = rgb
        img = _cont(img)
        ctx.save_for_backward(img, wp)
        ctx.cfg = (K, rgb)
        return ops.rgb_conv_fwd(img, wp, None, K, k, a, b, 1.0, 1.0)

    @staticmethod
    def backward(ctx, gy):
        img, wp = ctx.saved_tensors
        K, (k, a, b) = ctx.cfg
        g_img = RgbDgradFn.apply(gy, wp, img.shape[1], (k, a)) if ctx.needs_input_grad[0] else None
        g_wp = RgbWgradFn.apply(img, gy, (k, a, b), tuple(wp.shape)) if (ctx.needs_input_grad[1]
                                                                          and not _INPUT_GRAD_ONLY) else None
        return g_img, g_wp, None, None


class RgbDgradFn(Function):
    """d_img[n,c,h,w] = scale * sum_{taps,k} g[n,h+p-kh,w+p-kw,k] wp[(tap,c),k]."""

    @staticmethod
    def forward(ctx, g, wp, C, ks):
        k, scale = ks
        g = _cont(g)
        ctx.save_for_backward(g, wp)
        ctx.cfg = (C, ks)
        return ops.rgb_conv_dgrad(g, wp, None, C, k, act=0, out_scale=scale)

    @staticmethod
    def backward(ctx, h):
        g, wp = ctx.saved_tensors
        C, (k, scale) = ctx.cfg
        K = g.shape[3]
        g_g = RgbConvFn.apply(h, wp, K, (k, scale, 0.0)) if ctx.needs_input_grad[0] else None
        g_wp = RgbWgradFn.apply(h, g, (k, scale, 0.0), tuple(wp.shape)) if ctx.needs_input_grad[1] else None
        return g_g, g_wp, None, None


class RgbWgradFn(Function):
    @staticmethod
    def forward(ctx, img, g, rgb, wp_shape):
        k, a, b = rgb
        img, g = _cont(img), _cont(g)
        ctx.save_for_backward(img, g)
        ctx.cfg = rgb
        out = _packed_buffer(wp_shape, g.shape[3], img.device)
        return ops.rgb_conv_wgrad(img, g, k, a, b, out)

    @staticmethod
    def backward(ctx, h):
        img, g = ctx.saved_tensors
        k, a, b = ctx.cfg
        h = _cont(h)
        g_img = RgbDgradFn.apply(g, h, img.shape[1], (k, a)) if ctx.needs_input_grad[0] else None
        g_g = RgbConvFn.apply(img, h, g.shape[3], (k, a, b)) if ctx.needs_input_grad[1] else None
        return g_img, g_g, None, None


class RgbConvBiasActFn(Function):
    """FromRGB / first conv with the input rescale, bias and activation fused (one read of the image)."""

    @staticmethod
    def forward(ctx, img, wp, bias, K, rgb, slope, gain):
        k, a, b = rgb
        img = _cont(img)
        y = ops.rgb_conv_fwd(img, wp, bias, K, k, a, b, slope, gain)
        ctx.save_for_backward(img, wp, y)
        ctx.cfg = (K, rgb, slope, gain)
        return y

    @staticmethod
    def backward(ctx, gy):
        img, wp, y = ctx.saved_tensors
        K, (k, a, b), slope, gain = ctx.cfg
        g_pre = ActBwdFn.apply(gy, y, slope, gain)
        g_img = RgbDgradFn.apply(g_pre, wp, img.shape[1], (k, a)) if ctx.needs_input_grad[0] else None
        only_in = _INPUT_GRAD_ONLY
        g_wp = RgbWgradFn.apply(img, g_pre, (k, a, b), tuple(wp.shape)) if (ctx.needs_input_grad[1] and not only_in) else None
        g_b = ColSumFn.apply(g_pre) if (ctx.needs_input_grad[2] and not only_in) else None
        return g_img, g_wp, g_b, None, None, None, None


# ----------------------------------------------------------------------------------------------------------
# upfirdn2d (Blur / Upsample / Downsample), NHWC
# ----------------------------------------------------------------------------------------------------------
class UpFirDn2dFn(Function):
    """UpFirDn2d (op/upfirdn2d.py:88-142) with pad = (x0, x1, y0, y1)."""

    @staticmethod
    def forward(ctx, x, kernel, up, down, pad):
        x = _cont(x)
        kh, kw = kernel.shape
        B, H, W, C = x.shape
        out = ops.upfirdn2d(x, kernel, up, down, pad)
        oh, ow = out.shape[1], out.shape[2]
        px0, px1, py0, py1 = pad
        g_pad = (kw - px0 - 1, W * up - ow * down + px0 - up + 1,
                 kh - py0 - 1, H * up - oh * down + py0 - up + 1)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, g_pad, (H, W), (oh, ow))
        return out

    @staticmethod
    def backward(ctx, gy):
        kernel, = ctx.saved_tensors
        up, down, pad, g_pad, in_hw, out_hw = ctx.cfg
        return UpFirDn2dBackwardFn.apply(gy, kernel, up, down, pad, g_pad, in_hw), None, None, None, None


_FLIPPED = {}


def flipped_kernel(kernel):
    """torch.flip(kernel, [0, 1]) of a FIR buffer, built once per (storage, version): the transposed blur runs 13 times
    per StyleGAN2-32 step and each flip was its own 4 us launch on a 4 x 4 tensor."""
    key = (kernel.data_ptr(), kernel._version, tuple(kernel.shape), str(kernel.device))
    hit = _FLIPPED.get(key)
    if hit is None:
        if torch.cuda.is_current_stream_capturing():
            # a tensor created inside a capture lives in the graph's private pool and holds data only after a replay:
            # never let eager code find it in the cache
            return torch.flip(kernel.detach(), [0, 1]).contiguous()
        if len(_FLIPPED) > 64:
            _FLIPPED.clear()
        hit = _FLIPPED[key] = (torch.flip(kernel.detach(), [0, 1]).contiguous(), kernel)   # (keeps the storage alive)
    return hit[0]


class UpFirDn2dBackwardFn(Function):
    """UpFirDn2dBackward (op/upfirdn2d.py:19-85): upfirdn with the flipped kernel and up/down swapped; its own
    backward is the forward op again."""

    @staticmethod
    def forward(ctx, gy, kernel, up, down, pad, g_pad, in_hw):
        gk = flipped_kernel(kernel)
        gx = ops.upfirdn2d(_cont(gy), gk, down, up, g_pad)
        assert (gx.shape[1], gx.shape[2]) == tuple(in_hw), (gx.shape, in_hw)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad)
        return gx

    @staticmethod
    def backward(ctx, h):
        kernel, = ctx.saved_tensors
        up, down, pad = ctx.cfg
        return UpFirDn2dFn.apply(h, kernel, up, down, pad), None, None, None, None, None, None


class MinibatchStddevFn(Function):
    """The minibatch-stddev channel (stylegan2/discriminator.py:22-33) as a node of the any-order family: x (B,H,W,C) ->
    (B,H,W,Cp) = [x | group statistic | zero padding].  Its backward is MinibatchStddevBwdFn, which is differentiable once
    more -- the R1 penalty reaches the conv biases ONLY through this channel's curvature."""

    @staticmethod
    def forward(ctx, x, cpad, splits):
        if not x.is_contiguous():
            # a silent copy here would be saved WITHOUT autograd history: the double backward's gradient w.r.t. x (R1's
            # only path to the trunk through this channel) would be dropped
            raise ValueError('MinibatchStddevFn expects a dense NHWC activation (got strides %s)' % (x.stride(),))
        ctx.save_for_backward(x)
        ctx.splits = splits
        return ops.minibatch_stddev(0, x, cpad=cpad, splits=splits)

    @staticmethod
    def backward(ctx, gy):
        x, = ctx.saved_tensors
        return MinibatchStddevBwdFn.apply(x, gy, ctx.splits), None, None


class MinibatchStddevBwdFn(Function):
    @staticmethod
    def forward(ctx, x, gy, splits):
        gy = _cont(gy)
        ctx.save_for_backward(x, gy)
        ctx.splits = splits
        return ops.minibatch_stddev(1, x, gy=gy, splits=splits)

    @staticmethod
    @once_differentiable
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        gx2, ggy = ops.minibatch_stddev(2, x, gy=gy, h=_cont(h), splits=ctx.splits)
        return gx2, ggy, None


class SumSqMeanFn(Function):
    """mean_n sum_chw g^2 (r1_loss, train_stylegan2.py:112) in two launches; backward = g * (2 / N * grad_out), with grad_out
    read on the device."""

    @staticmethod
    def forward(ctx, g):
        g = _cont(g)
        ctx.save_for_backward(g)
        ctx.n = g.shape[0]
        return ops.sumsq(g, 1.0 / g.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        g, = ctx.saved_tensors
        return ops.scale_dev(g, _cont(gout), 2.0 / ctx.n)


class ZeroGradFn(Function):
    """Identity on ``out`` that hands exact-zero gradients to ``params`` -- the effect of the reference's
    ``output + (projection.mean() + projection2.mean()) * 0.`` (models/gan/base.py:139-141) for a call that skips the
    projection GEMMs: every parameter of the module still receives a gradient, so optimizer state and step counts stay
    uniform across training modes."""

    @staticmethod
    def forward(ctx, out, *params):
        ctx.shapes = [p.shape for p in params]
        return out.view(out.shape)

    @staticmethod
    def backward(ctx, g):
        need = ctx.needs_input_grad[1:]
        if _INPUT_GRAD_ONLY or not any(need):          # frozen D (generator step), R1's first backward: nothing to fill
            return (g,) + (None,) * len(ctx.shapes)
        flat = torch.zeros(sum(int(torch.Size(s).numel()) for s, nd in zip(ctx.shapes, need) if nd),
                           device=g.device, dtype=g.dtype)
        outs, o = [], 0
        for s, nd in zip(ctx.shapes, need):
            if not nd:
                outs.append(None)
                continue
            n = int(torch.Size(s).numel())
            outs.append(flat[o:o + n].view(s))
            o += n
        return (g,) + tuple(outs)


class LinCombFn(Function):
    """y = a*x + b*z  (ResBlock merge, discriminator.py:72-74)."""

    @staticmethod
    def forward(ctx, x, z, a, b):
        ctx.cfg = (a, b)
        return ops.lincomb(_cont(x), _cont(z), a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.cfg
        if a == 1.0 and b == 1.0:
            return g, g, None, None       # plain sum (the ResBlock merge with its 1/sqrt2 folded into both producers)
        ga = g * a
        return ga, (ga if b == a else g * b), None, None


# ----------------------------------------------------------------------------------------------------------
# weight packing (fixed runtime scale): OIHW parameters -> packed GEMM layout, batched over layers
# ----------------------------------------------------------------------------------------------------------
class PackMeta(object):
    """entries: list of (K, C, T, scale, group, col_off); groups: list of (rows, cols)."""

    def __init__(self, entries, groups):
        self.entries, self.groups = entries, groups
        self.comm = None            # engine.OverlappedGradReducer when the gradient exchange happens at the packed level
        self.reduced = set()        # groups whose packed gradient a producer has already all-reduced in this backward
        self.shared = False         # several discriminator calls use this pack: no production-site exchange
        self.dead = False           # its backward has run (the graph may be freed): do not reuse
        self.on_backward = None     # callback of the owner (drops its cached reference)


def _specs_from(meta, ws):
    return [ops.SnSpec(w, fixed_scale=e[3], view_kct=(e[0], e[1], e[2])) for e, w in zip(meta.entries, ws)]


class PackWeightsFn(Function):
    """(w_0 .. w_{n-1}) -> one packed tensor per group; wp = w * scale in layout [(tap*C + c)][K] at the group's
    column offset (EqualConv2d / EqualLinear runtime scale folded in, stylegan2/layers.py:104,117,141)."""

    @staticmethod
    def forward(ctx, meta, *ws):
        dev = ws[0].device
        ws = [_cont(w) for w in ws]
        kcols = [sum(e[0] for e in meta.entries if e[4] == gi) for gi in range(len(meta.groups))]
        outs = [_packed_buffer((r, c), kcols[gi], dev) for gi, (r, c) in enumerate(meta.groups)]
        specs = _specs_from(meta, ws)
        wps = [outs[e[4]][:, e[5]:e[5] + e[0]] for e in meta.entries]
        ldws = [meta.groups[e[4]][1] for e in meta.entries]
        offs, n = ops.sn_scratch_floats(specs)
        scratch = torch.empty(n, device=dev, dtype=torch.float32)
        sigma = torch.empty(len(specs), device=dev, dtype=torch.float32)
        ops.sn_weight_prep(specs, wps, ldws, False, scratch, offs, sigma)
        ctx.meta = meta
        ctx.shapes = [tuple(w.shape) for w in ws]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        if _INPUT_GRAD_ONLY:
            # input_grad_only() is a process-wide switch around ONE autograd.grad call that asks for d / d images only;
            # a weight gradient arriving here means another backward is running inside it and is silently losing its
            # parameter gradients -- refuse instead
            raise RuntimeError('PackWeightsFn.backward inside autograd_ops.input_grad_only(): a backward that needs '
                               'parameter gradients is running while they are switched off')
        meta = ctx.meta
        meta.dead = True
        if meta.on_backward is not None:
            meta.on_backward()
        gouts = [g if g is not None else None for g in gouts]
        comm = getattr(meta, 'comm', None)
        if comm is not None and not torch.is_grad_enabled():
            # data-parallel exchange at the packed level (the fixed EqualConv scale is the same on every rank, so reducing
            # the packed gradient is reducing the parameter gradient): groups a first-order producer already all-reduced
            # while the backward went on (meta.reduced) are only waited for, everything else -- the any-order graph of
            # an R1 call, where a packed weight collects several contributions -- is reduced here
            for gi, g in enumerate(gouts):
                if g is not None and gi not in meta.reduced:
                    comm.reduce_async(g)
            comm.wait()
        gws = UnpackWeightsFn.apply(meta, ctx.shapes, *gouts)
        return (None,) + tuple(gws)


class UnpackWeightsFn(Function):
    """Packed gradients (one per group) -> OIHW gradients * scale.  Inverse-layout twin of PackWeightsFn."""

    @staticmethod
    def forward(ctx, meta, shapes, *gouts):
        dev = next(g for g in gouts if g is not None).device
        gouts = [(_cont(g) if g is not None else torch.zeros(meta.groups[i], device=dev)) for i, g in enumerate(gouts)]
        # one flat allocation, per-parameter views: the data-parallel exchange (engine.GradAllReducer) then moves all
        # weight gradients of a network in ONE collective instead of one per parameter
        sizes = [int(math.prod(s)) for s in shapes]
        offs, tot = [], 0
        for n_ in sizes:
            offs.append(tot)
            tot += ops.round_up(n_, 4)
        flat = torch.empty(max(tot, 4), device=dev, dtype=torch.float32)
        gws = [flat[o:o + n_].view(s) for o, n_, s in zip(offs, sizes, shapes)]
        specs = _specs_from(meta, gws)           # .w is only a non-null placeholder on the fixed-scale path
        gwps = [gouts[e[4]][:, e[5]:e[5] + e[0]] for e in meta.entries]
        ldws = [meta.groups[e[4]][1] for e in meta.entries]
        offs, n = ops.sn_scratch_floats(specs)
        scratch = torch.empty(n, device=dev, dtype=torch.float32)
        sigma = _device_constant(tuple(1.0 / e[3] for e in meta.entries), dev)
        ops.sn_weight_grad(specs, gwps, ldws, gwps, gws, scratch, offs, sigma)
        ctx.meta, ctx.shapes = meta, shapes
        return tuple(gws)

    @staticmethod
    def backward(ctx, *hs):
        dev = next(h for h in hs if h is not None).device
        hs = [h if h is not None else torch.zeros(s, device=dev) for h, s in zip(hs, ctx.shapes)]
        outs = PackWeightsFn.apply(ctx.meta, *hs)
        return (None, None) + tuple(outs)


# ----------------------------------------------------------------------------------------------------------
# modulated-convolution pieces of the StyleGAN2 generator (first-order backward: the generator step never needs
# a double backward -- R1 differentiates the discriminator w.r.t. its input images only)
# ----------------------------------------------------------------------------------------------------------
class NhwcScaleFn(Function):
    """y[n,h,w,c] = x[n,h,w,c] * s[n,c]  (weight modulation moved onto the activations, generator.py:52-60)."""

    @staticmethod
    def forward(ctx, x, s):
        x, s = _cont(x), _cont(s)
        ctx.save_for_backward(x, s)
        return ops.nhwc_scale(x, s)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        g = _cont(g)
        gx = ops.nhwc_scale(g, s) if ctx.needs_input_grad[0] else None
        gs = ops.nhwc_dot(g, x) if ctx.needs_input_grad[1] else None
        return gx, gs


class ModconvEpilogueFn(Function):
    """out = sqrt2 * lrelu_0.2(y * demod[n,k] + noise_w * noise[n,h,w] + bias[k]): demodulation (generator.py:62-64),
    NoiseInjection (:85-94) and FusedLeakyReLU (:113-118) in one pass; the backward returns the gradients of the conv
    output, the demodulation factors, the noise strength and the bias."""

    @staticmethod
    def forward(ctx, y, demod, noise, noise_w, bias):
        y, demod, noise = _cont(y), _cont(demod), _cont(noise)
        out = ops.modconv_epilogue_(y, demod, noise, noise_w, bias, out=torch.empty_like(y))
        ctx.save_for_backward(y, demod, noise, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        y, demod, noise, out = ctx.saved_tensors
        g_pre = ops.fused_bias_act(_cont(g), None, out, 3, 1, 0.2, SQRT2)
        gy = ops.nhwc_scale(g_pre, demod) if ctx.needs_input_grad[0] else None
        gdemod = ops.nhwc_dot(g_pre, y) if ctx.needs_input_grad[1] else None
        gnw = ops.nhwc_dot(g_pre, noise, per_channel=False).sum().reshape(1) if ctx.needs_input_grad[3] else None
        gb = ops.colstats(g_pre.view(-1, g_pre.shape[-1]))[0] if ctx.needs_input_grad[4] else None
        return gy, gdemod, None, gnw, gb


class SnPackWeightsFn(Function):
    """Spectral-norm weight preparation as an autograd node (torch.nn.utils.spectral_norm's pre-forward hook + its
    autograd, applied to every Conv2d / Linear of a discriminator: sndcgan.py:111-118, snresnet.py:59-66): ONE batched
    launch runs the power iteration of every layer (train mode; ``weight_u`` / ``weight_v`` updated in place), computes
    sigma = u^T W v and writes W / sigma in the packed GEMM layout; the backward maps the packed gradients to
    d loss / d weight_orig = G / sigma - (<G, W> / sigma^2) u v^T with the u, v of THIS forward.  First-order."""

    @staticmethod
    def forward(ctx, mods, training, *ws):
        dev = ws[0].device
        specs = [ops.SnSpec(w, m.weight_u, m.weight_v) for m, w in zip(mods, ws)]
        ldws = [ops.round_up(sp.K, 4) for sp in specs]
        outs = [_packed_buffer((sp.T * sp.C, ld), sp.K, dev) for sp, ld in zip(specs, ldws)]
        offs, n = ops.sn_scratch_floats(specs)
        scratch = torch.empty(n, device=dev, dtype=torch.float32)
        sigma = torch.empty(len(specs), device=dev, dtype=torch.float32)
        u_snaps = [torch.empty(sp.K, device=dev, dtype=torch.float32) for sp in specs]
        v_snaps = [torch.empty(sp.C * sp.T, device=dev, dtype=torch.float32) for sp in specs]
        ops.sn_weight_prep(specs, outs, ldws, training, scratch, offs, sigma, u_snaps, v_snaps)
        ctx.state = (specs, ldws, offs, n, sigma, u_snaps, v_snaps, outs)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        specs, ldws, offs, n, sigma, u_snaps, v_snaps, outs = ctx.state
        dev = outs[0].device
        gwps = [(_cont(g) if g is not None else torch.zeros_like(o)) for g, o in zip(gouts, outs)]
        gws = [torch.empty_like(sp.w) for sp in specs]
        scratch = torch.empty(n, device=dev, dtype=torch.float32)
        ops.sn_weight_grad(specs, outs, ldws, gwps, gws, scratch, offs, sigma, u_snaps, v_snaps)
        return (None, None) + tuple(gws)


def make_blur_kernel(k=(1, 3, 3, 1)):
    """make_kernel (stylegan2/layers.py:24-32)."""
    k = torch.tensor(k, dtype=torch.float32)
    k = k[None, :] * k[:, None]
    return k / k.sum()


SQRT2 = math.sqrt(2.0)
