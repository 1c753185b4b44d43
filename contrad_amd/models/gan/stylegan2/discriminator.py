"""ResidualDiscriminatorP (models/gan/stylegan2/discriminator.py:191-235) on the HIP library.

FromRGB 1x1 (+ input rescale x*2-1, bias, lrelu(0.2)*sqrt2 fused) -> ResBlocks [conv3x3+act; blur(2,2) -> conv3x3
stride 2 + act; skip: blur(1,1) -> conv1x1 stride 2; (out+skip)/sqrt2] -> minibatch-stddev channel -> last_conv ->
8192 features -> the BaseDiscriminator heads (plain nn.Linear, no spectral norm here).  Built from the
any-order-differentiable nodes of contrad_amd.autograd_ops, so ``autograd.grad(..., create_graph=True)`` -- the R1
penalty (train_stylegan2.py:106-113) -- works as it does in the reference.  Activations are NHWC internally, every
EqualConv2d scale 1/sqrt(fan_in) is folded into the packed weights, state-dict names match the reference.
"""
import math
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import autograd_ops as A
from .... import ops
from ..base import BaseDiscriminator, PlainParams, TinyHead, _Act, make_projection

_SLOPE, _GAIN = 0.2, math.sqrt(2.0)
# per-discriminator pack of the current step (ResidualDiscriminatorP._pack); outside the module so that deepcopy / pickling
# of a discriminator never meets autograd-graph tensors
_PACK_CACHE = weakref.WeakKeyDictionary()
_HEAD_SLOPE = 0.1


class _EqualConvParams(nn.Module):
    """EqualConv2d's parameter (stylegan2/layers.py:95-123): ``weight`` ~ N(0,1), runtime scale 1/sqrt(fan_in)."""

    def __init__(self, in_channel, out_channel, kernel_size):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)


class _ActBias(nn.Module):
    """FusedLeakyReLU's ``bias`` (op/fused_act.py:74-83)."""

    def __init__(self, channel):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))


class _BlurBuf(nn.Module):
    """Blur's ``kernel`` buffer (stylegan2/layers.py:76-92)."""

    def __init__(self, pad):
        super().__init__()
        self.register_buffer('kernel', A.make_blur_kernel((1, 3, 3, 1)))
        self.pad = pad


def _conv_layer(in_channel, out_channel, kernel_size, downsample=False, activate=True):
    """Same child indices as the reference's ConvLayer (layers.py:174-198): [Blur,] EqualConv2d [, FusedLeakyReLU]."""
    layers = []
    if downsample:
        p = (4 - 2) + (kernel_size - 1)
        layers.append(_BlurBuf(((p + 1) // 2, p // 2)))
    layers.append(_EqualConvParams(in_channel, out_channel, kernel_size))
    if activate:
        layers.append(_ActBias(out_channel))
    return nn.Sequential(*layers)


class _ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv1 = _conv_layer(in_channel, in_channel, 3)
        self.conv2 = _conv_layer(in_channel, out_channel, 3, downsample=True)
        self.skip = _conv_layer(in_channel, out_channel, 1, downsample=True, activate=False)
        self.cin, self.cout = in_channel, out_channel


def minibatch_stddev_batches(x, batch_splits, stddev_group=4):
    """The stddev channel with the statistics confined to consecutive sub-batches: ``batch_splits`` = sizes of the
    discriminator calls that were merged into this one (see ResidualDiscriminatorP.call_batches).  One HIP launch per
    sub-batch (autograd_ops.MinibatchStddevFn: forward, backward and the backward's backward for R1); the channel count is
    padded with zeros to a multiple of 16 (513 -> 528: a K-tile of the lean conv loop must not straddle a filter tap)."""
    assert stddev_group == 4
    B, H, W, C = x.shape
    splits = tuple(batch_splits) if batch_splits and len(batch_splits) > 1 else (B,)
    assert sum(splits) == B
    return A.MinibatchStddevFn.apply(x, C + 1 + ((-(C + 1)) % 16), splits)


def minibatch_stddev_nhwc(x, stddev_group=4):
    """_minibatch_stddev_layer (discriminator.py:22-33) on NHWC in differentiable torch ops -- the cross-check of the HIP node
    family (tests/test_stylegan2_gpu.py); the model itself calls minibatch_stddev_batches.  One extra channel holding, for
    sample b, the mean over (C,H,W) of the std over the group {b mod M, + M, + 2M, ...}, M = B / group.  The channel
    dimension is padded with zeros to a multiple of 16 (513 -> 528): a K-tile of the lean conv loop must not straddle a
    filter tap, and the 516-channel layout of round 1 sent last_conv to the general kernel (37 TF/s)."""
    B, H, W, C = x.shape
    group = min(B, stddev_group)
    y = x.reshape(group, B // group, H, W, C)
    std = torch.sqrt(y.var(0, unbiased=False) + 1e-8)
    std = std.mean([1, 2, 3], keepdim=True)                       # (M,1,1,1)
    std = std.repeat(group, H, W, 1)                              # (B,H,W,1)
    pad = (-(C + 1)) % 16
    parts = [x, std]
    if pad:
        parts.append(x.new_zeros(B, H, W, pad))
    return torch.cat(parts, dim=3)


class _TrunkFn(torch.autograd.Function):
    """FromRGB + all ResBlocks as ONE first-order autograd node (the ContraD discriminator calls: augmented images are
    constants there, so neither d/d images nor a double backward is needed; the R1 call and the generator step keep the
    any-order node family).  What the fusion buys over the node-per-op graph:
      * the residual merge (out + skip) / sqrt2 (discriminator.py:72-74) costs no scaling pass: 1/sqrt2 is folded into
        the conv2 activation gain (sqrt2 / sqrt2 = 1) and into the packed skip weights;
      * no FusedLeakyReLU backward pass (op/fused_act.py:20-55) and no gradient-accumulation add exists: the
        transposed blur of the conv1 branch multiplies by act'(o) in its epilogue, and the transposed decimating blur
        of the skip branch adds the conv1 branch's input gradient and emits BOTH the plain sum (for the next skip
        branch) and the sum times act'(y2) (for the next conv2 branch) -- contrad_upfirdn2d_fused.
    Inputs: images, blur kernel, then (wp, bias) of FromRGB and per block (wp1, b1, wp2, b2, wp_skip)."""

    @staticmethod
    def forward(ctx, D, images, kernel, *wb):
        blocks = list(D.layers)[1:]
        wp_rgb, b_rgb = wb[0], wb[1]
        K0 = D.layers[0][0].weight.shape[0]
        r0 = ops.rgb_conv_fwd(images, wp_rgb, b_rgb, K0, 1, 2.0, -1.0, _SLOPE, _GAIN)
        saved, geo = [r0], []
        x = r0
        for bi, blk in enumerate(blocks):
            wp1, b1, wp2, b2, wps = wb[2 + 5 * bi: 7 + 5 * bi]
            ci, co = blk.cin, blk.cout
            o = ops.conv2d_fwd(x, wp1, b1, ci, 3, 3, 1, 1, _SLOPE, _GAIN)
            p0, p1 = blk.conv2[0].pad
            ob = ops.upfirdn2d(o, kernel, 1, 1, (p0, p1, p0, p1))
            y2 = ops.conv2d_fwd(ob, wp2, b2, co, 3, 3, 2, 0, _SLOPE, 1.0)         # gain sqrt2 * (1/sqrt2 of the merge)
            q0, q1 = blk.skip[0].pad
            sb = ops.upfirdn2d(x, kernel, 1, 2, (q0, q1, q0, q1))
            # skip weights carry the 1/sqrt2; the merge y2 + skip is the skip conv's epilogue (no lincomb pass)
            xn = ops.conv2d_fwd(sb, wps, None, co, 1, 1, 1, 0, addend=y2)
            saved += [x, o, ob, y2, sb]
            geo.append((ci, co, (p0, p1), (q0, q1)))
            x = xn
        if getattr(D, '_record_activations', False):             # test hook: the linear regions actually used
            D._trunk_rec = [r0] + [t for bi in range(len(blocks)) for t in (saved[2 + 5 * bi], saved[4 + 5 * bi])]
        ctx.save_for_backward(images, kernel, *wb, *saved)
        ctx.n_wb, ctx.geo = len(wb), geo
        ctx.xch = getattr(D, '_trunk_xch', None)      # (comm, meta, group ids): all-reduce each weight gradient as produced
        return x

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        t = ctx.saved_tensors
        grads, _ = _trunk_backward_chain(t[0], t[1], t[2:2 + ctx.n_wb], t[2 + ctx.n_wb:], ctx.geo, g, True, False, ctx.xch)
        return (None, None, None) + tuple(grads)


def _trunk_backward_chain(images, kernel, wb, saved, geo, g, want_w, want_img, xch=None, keep=None, img_c=3):
    """The trunk's vector-Jacobian product from the gradient ``g`` at its output: every packed weight / bias gradient
    (``want_w``) and / or the gradient w.r.t. the images (``want_img``).  ``keep`` (a dict) receives, per block, the
    signals the double backward needs: the gradient at the block output, plain (``g``) and masked by act'(y2) (``gm``),
    and the gradient at conv1's pre-activation (``g_o``); key -1: the gradient at FromRGB's pre-activation."""
    n_wb = len(wb)
    r0 = saved[0]
    nb = len(geo)
    gk = A.flipped_kernel(kernel)
    grads = [None] * n_wb
    g = g.contiguous()
    gm = None                                   # g * act'(y2 of the current block)
    for bi in range(nb - 1, -1, -1):
        x, o, ob, y2, sb = saved[1 + 5 * bi: 6 + 5 * bi]
        wp1, b1, wp2, b2, wps = wb[2 + 5 * bi: 7 + 5 * bi]
        ci, co, (p0, p1), (q0, q1) = geo[bi]
        if gm is None:
            gm = ops.fused_bias_act(g, None, y2, 3, 1, _SLOPE, 1.0)
        # conv2 (3x3 stride 2 on the blurred map)
        if want_w:
            gw2 = A._packed_buffer(wp2.shape, co, wp2.device); gb2 = torch.empty_like(b2)
            ops.conv2d_wgrad(ob, gm, 3, 3, 2, 0, out=gw2, dbias=gb2)
        g_ob = ops.conv2d_dgrad(gm, wp2, tuple(ob.shape), 3, 3, 2, 0)
        # blur^T, times act'(o) * sqrt2
        gp = (4 - p0 - 1, o.shape[2] - ob.shape[2] + p0, 4 - p0 - 1, o.shape[1] - ob.shape[1] + p0)
        _, g_o = ops.upfirdn2d_fused(g_ob, gk, 1, 1, gp, act_ref=o, slope=_SLOPE, gain=_GAIN, want_out=False,
                                     want_out2=True)
        del g_ob
        if want_w:
            gw1 = A._packed_buffer(wp1.shape, ci, wp1.device); gb1 = torch.empty_like(b1)
            ops.conv2d_wgrad(x, g_o, 3, 3, 1, 1, out=gw1, dbias=gb1)
        g_x1 = ops.conv2d_dgrad(g_o, wp1, tuple(x.shape), 3, 3, 1, 1)
        # skip branch (1x1 conv on the blurred + decimated map), driven by the un-masked g
        if want_w:
            gws = A._packed_buffer(wps.shape, co, wps.device)
            ops.conv2d_wgrad(sb, g, 1, 1, 1, 0, out=gws)
        g_sb = ops.conv2d_dgrad(g, wps, tuple(sb.shape), 1, 1, 1, 0)
        if keep is not None:
            keep[bi] = (g, gm, g_o)
        del g_o
        gq = (4 - q0 - 1, x.shape[2] - sb.shape[2] * 2 + q0, 4 - q0 - 1, x.shape[1] - sb.shape[1] * 2 + q0)
        if bi > 0:      # gradient of this block's input = previous block's output: plain sum + masked sum
            y2_prev = saved[1 + 5 * (bi - 1) + 3]
            g, gm = ops.upfirdn2d_fused(g_sb, gk, 2, 1, gq, addend=g_x1, act_ref=y2_prev, slope=_SLOPE, gain=1.0,
                                        want_out=True, want_out2=True)
        else:           # FromRGB output: only the masked sum is needed
            _, gm = ops.upfirdn2d_fused(g_sb, gk, 2, 1, gq, addend=g_x1, act_ref=r0, slope=_SLOPE, gain=_GAIN,
                                        want_out=False, want_out2=True)
        del g_sb, g_x1
        if want_w:
            grads[2 + 5 * bi: 7 + 5 * bi] = [gw1, gb1, gw2, gb2, gws]
            if xch is not None:       # data parallel: these three slabs are final -- start their all-reduce now
                comm, meta, gids = xch
                for k_, gw_ in ((2, gw1), (4, gw2), (6, gws)):
                    A.exchange_packed((comm, meta, gids[5 * bi + k_]), gw_)
    if keep is not None:
        keep[-1] = gm
    if want_w:
        gw0 = A._packed_buffer(wb[0].shape, wb[1].numel(), wb[0].device); gb0 = torch.empty_like(wb[1])
        ops.rgb_conv_wgrad(images, gm, 1, 2.0, -1.0, gw0, gb0)
        if xch is not None:
            A.exchange_packed((xch[0], xch[1], xch[2][0]), gw0)
        grads[0], grads[1] = gw0, gb0
    g_img = ops.rgb_conv_dgrad(gm, wb[0], None, img_c, 1, act=0, out_scale=2.0) if want_img else None
    return grads, g_img


class _TrunkR1Fn(torch.autograd.Function):
    """The trunk for images that NEED a gradient: the R1 penalty's D(x) (train_stylegan2.py:106-113) and the generator
    step.  Same forward kernels as _TrunkFn; what differs is the backward:
      * first-order (generator step, and the final backward of an R1 step -- the path through the minibatch-stddev
        curvature, which is where the bias gradients of r1 come from): the fused chain of _TrunkFn plus FromRGB's data
        gradient (d / d images);
      * under ``create_graph`` (R1's ``autograd.grad(d_real.sum(), images, create_graph=True)``): d / d images is
        returned by a second node, _TrunkVJPFn, so that ``r1 = |d D / d x|^2`` can be differentiated again.
    Until round 3 this call ran on the any-order node family: ~100 autograd nodes per direction, every residual fan-out an
    ATen ``add`` pass, every LeakyReLU derivative its own pass (55 adds, 19 muls, ~20 mask passes per StyleGAN2-32 step).
    The packed weights are inputs of both nodes, so autograd sums their two contributions (term (i): through the
    Jacobian's dependence on the weights, from _TrunkVJPFn.backward; term (ii): through the head's curvature, from this
    node's first-order backward)."""

    @staticmethod
    def forward(ctx, D, images, kernel, *wb):
        return _TrunkFn.forward(ctx, D, images, kernel, *wb)

    @staticmethod
    def backward(ctx, g):
        t = ctx.saved_tensors
        images, kernel = t[0], t[1]
        wb, saved = t[2:2 + ctx.n_wb], t[2 + ctx.n_wb:]
        need_img = ctx.needs_input_grad[1]
        need_w = any(ctx.needs_input_grad[3:]) and not A._INPUT_GRAD_ONLY
        if torch.is_grad_enabled():
            # create_graph: the image gradient must stay differentiable.  (Parameter gradients of THIS backward are not
            # differentiable here -- nothing on the path differentiates them -- and are skipped when the caller asked
            # for d / d images only, autograd_ops.input_grad_only.)
            g_img = _TrunkVJPFn.apply(g, ctx.n_wb, ctx.geo, tuple(images.shape), kernel, *wb, *saved) if need_img else None
            grads = [None] * ctx.n_wb
            if need_w:
                with torch.no_grad():
                    grads, _ = _trunk_backward_chain(images, kernel, [w.detach() for w in wb], saved, ctx.geo, g.detach(),
                                                     True, False)
            return (None, g_img, None) + tuple(grads)
        grads, g_img = _trunk_backward_chain(images, kernel, wb, saved, ctx.geo, g, need_w, need_img, None, None,
                                             images.shape[1])
        return (None, g_img, None) + tuple(grads)


class _TrunkVJPFn(torch.autograd.Function):
    """g_top (gradient at the trunk output) -> d / d images, as a differentiable node: the forward IS the trunk's
    backward chain.  Its backward, given h = d r1 / d (d D / d images), is the forward-mode (tangent) pass of the
    linearised trunk -- the same convolutions and blurs as the forward, with the LeakyReLU regions of the forward pass as
    fixed masks -- which yields  d / d g_top = J h  (continues into the head's double backward)  and, layer by layer, the
    weight gradient of term (i):  wgrad(tangent at the layer input, gradient signal at its pre-activation output)."""

    @staticmethod
    def forward(ctx, g_top, n_wb, geo, img_shape, kernel, *rest):
        wb, saved = rest[:n_wb], rest[n_wb:]
        keep = {}
        _, g_img = _trunk_backward_chain(None, kernel, wb, saved, geo, g_top, False, True, None, keep, img_shape[1])
        nb = len(geo)
        flat = [keep[-1]]
        for bi in range(nb):
            flat += list(keep[bi])
        ctx.save_for_backward(kernel, *wb, *saved, *flat)
        ctx.cfg = (n_wb, geo, len(saved))
        return g_img

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, h):
        n_wb, geo, n_saved = ctx.cfg
        t = ctx.saved_tensors
        kernel = t[0]
        wb, saved, flat = t[1:1 + n_wb], t[1 + n_wb:1 + n_wb + n_saved], t[1 + n_wb + n_saved:]
        nb = len(geo)
        r0, gm0 = saved[0], flat[0]
        h = h.contiguous()
        need_w = any(ctx.needs_input_grad[5:5 + n_wb])
        gws_out = [None] * n_wb
        K0 = r0.shape[3]
        if need_w:
            gw0 = A._packed_buffer(wb[0].shape, K0, wb[0].device)
            ops.rgb_conv_wgrad(h, gm0, 1, 2.0, 0.0, gw0)
            gws_out[0] = gw0
        U = ops.rgb_conv_fwd(h, wb[0], None, K0, 1, 2.0, 0.0, 1.0, 1.0)             # tangent at FromRGB's pre-activation
        U = ops.fused_bias_act(U, None, r0, 3, 1, _SLOPE, _GAIN)                     # ... at its output
        for bi in range(nb):
            x, o, ob, y2, sb = saved[1 + 5 * bi: 6 + 5 * bi]
            wp1, b1, wp2, b2, wps = wb[2 + 5 * bi: 7 + 5 * bi]
            ci, co, (p0, p1), (q0, q1) = geo[bi]
            g, gm, g_o = flat[1 + 3 * bi: 4 + 3 * bi]
            if need_w:
                gw1 = A._packed_buffer(wp1.shape, ci, wp1.device)
                ops.conv2d_wgrad(U, g_o, 3, 3, 1, 1, out=gw1)
                gws_out[2 + 5 * bi] = gw1
            t1 = ops.conv2d_fwd(U, wp1, None, ci, 3, 3, 1, 1)
            t1 = ops.fused_bias_act(t1, None, o, 3, 1, _SLOPE, _GAIN)
            tb = ops.upfirdn2d(t1, kernel, 1, 1, (p0, p1, p0, p1))
            del t1
            if need_w:
                gw2 = A._packed_buffer(wp2.shape, co, wp2.device)
                ops.conv2d_wgrad(tb, gm, 3, 3, 2, 0, out=gw2)
                gws_out[4 + 5 * bi] = gw2
            t2 = ops.conv2d_fwd(tb, wp2, None, co, 3, 3, 2, 0)
            del tb
            t2 = ops.fused_bias_act(t2, None, y2, 3, 1, _SLOPE, 1.0)
            usb = ops.upfirdn2d(U, kernel, 1, 2, (q0, q1, q0, q1))
            if need_w:
                gws = A._packed_buffer(wps.shape, co, wps.device)
                ops.conv2d_wgrad(usb, g, 1, 1, 1, 0, out=gws)
                gws_out[6 + 5 * bi] = gws
            U = ops.conv2d_fwd(usb, wps, None, co, 1, 1, 1, 0, addend=t2)
            del t2, usb
        return (U, None, None, None, None) + tuple(gws_out) + (None,) * n_saved


class ResidualDiscriminatorP(BaseDiscriminator):
    def __init__(self, size, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), small32=False, mlp_linear=True,
                 d_hidden=512, d_project=128):
        super().__init__()
        if tuple(blur_kernel) != (1, 3, 3, 1) or not mlp_linear:
            raise NotImplementedError('configuration outside get_architecture()')
        if small32:
            channels = {4: 512, 8: 512, 16: 256, 32: 128}
        else:
            channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: int(256 * channel_multiplier),
                        128: int(128 * channel_multiplier), 256: int(64 * channel_multiplier),
                        512: int(32 * channel_multiplier), 1024: int(16 * channel_multiplier)}
        self.size = size
        self.n_features = channels[4] * 4 * 4
        self.d_penul = self.n_features
        self.n_classes, self.d_hidden, self.d_project = 1, d_hidden, d_project

        self.linear = TinyHead(self.n_features, d_hidden, spectral=False)
        self.projection = make_projection(self.n_features, d_hidden, d_project, spectral=False)
        self.projection2 = make_projection(self.n_features, d_hidden, d_project, spectral=False)

        layers = [_conv_layer(3, channels[size], 1)]                 # FromRGB
        log_size = int(math.log(size, 2))
        in_channel = channels[size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            layers.append(_ResBlock(in_channel, out_channel))
            in_channel = out_channel
        self.layers = nn.Sequential(*layers)
        self.last_conv = _conv_layer(in_channel + 1, channels[4], 3)
        self.c_last_in = in_channel
        self.fuse_trunk = True          # (tests switch it off to compare the two graph constructions)
        self.fuse_r1 = True             # images with a gradient (R1, generator step): _TrunkR1Fn instead of the node family
        self._batch_splits = None
        self._pack_comm = None          # engine.OverlappedGradReducer: exchange the packed weight gradients in-backward
        self._cur_meta = None

    def enable_grad_overlap(self, comm):
        """Exchange the weight gradients inside the backward (engine.setup_grad_exchange); returns the parameters that
        are NOT covered -- the biases, which the caller reduces afterwards.  Pass None to disable."""
        self._pack_comm = comm
        return self.overlap_rest() if comm is not None else list(self.parameters())

    def drop_pack_cache(self):
        """Forget the shared weight pack of the current step (engine: before a stream capture)."""
        _PACK_CACHE.pop(self, None)

    def overlap_rest(self):
        packed_ids = {id(m.weight) for m in self.modules() if isinstance(m, (_EqualConvParams, PlainParams))}
        return [p for p in self.parameters() if id(p) not in packed_ids]

    def call_batches(self, batches, **flags):
        """Several discriminator calls with the same flags as ONE pass: every layer of this network acts per sample
        except the minibatch-stddev statistics (discriminator.py:22-33), which stay confined to each original call's
        batch.  Same results as ``[self(b, **flags) for b in batches]`` -- with one weight preparation, one launch per
        layer over the concatenated batch and one weight-gradient reduction instead of one per call.  Returns the list
        of per-call results."""
        sizes = [int(b.shape[0]) for b in batches]
        self._batch_splits = sizes
        try:
            out = self(torch.cat(list(batches), dim=0), **flags)
        finally:
            self._batch_splits = None
        if isinstance(out, tuple):
            logits, aux = out
            parts = torch.split(logits, sizes, dim=0)
            auxs = [{k: torch.split(v, sizes, dim=0)[i] for k, v in aux.items()} for i in range(len(sizes))]
            return [(parts[i], auxs[i]) for i in range(len(sizes))]
        return list(torch.split(out, sizes, dim=0))

    def call_merged(self, x, sizes, **flags):
        """``call_batches`` for calls whose inputs already sit one after the other in ``x`` (sum(sizes) rows): returns the
        MERGED result -- logits / embeddings of all calls in the row order of ``x`` -- so that a caller that wants them
        concatenated anyway (train_stylegan2_contraD.py:95-109 concatenates real views and fakes before the losses) needs
        neither the per-call split nor the concatenation, forward or backward."""
        self._batch_splits = [int(n) for n in sizes]
        try:
            if sum(self._batch_splits) != x.shape[0]:      # (not an assert: a wrong split silently mis-segments the minibatch stddev)
                raise ValueError('call_merged: sizes %r do not add up to the %d rows of x' % (self._batch_splits, x.shape[0]))
            return self(x, **flags)
        finally:
            self._batch_splits = None

    # ---- weight packing plan -------------------------------------------------------------------------
    def _pack(self, fused=True):
        """``fused`` (both graph constructions use it): the skip weights carry the residual merge's 1/sqrt2 -- together
        with conv2's activation gain sqrt2 / sqrt2 = 1 the merge (out + skip) / sqrt2 (discriminator.py:72-74) becomes a
        plain sum, whose backward is the identity on both branches (no scaling pass, forward or backward).
        Two discriminator calls on the SAME weights inside one step (the ContraD call and the R1 call) share one pack:
        keyed on the parameters' version counters, dropped as soon as its backward has run."""
        comm = self._pack_comm if (self._pack_comm is not None and self._pack_comm.active()
                                   and torch.is_grad_enabled()) else None
        plist = list(self.parameters())
        key = (tuple(p._version for p in plist), sum(1 for p in plist if p.requires_grad), torch.is_grad_enabled(),
               id(comm), plist[0].device)
        c = _PACK_CACHE.get(self)
        if c is not None and c[0] == key and not c[1].dead:
            c[1].shared = True          # (its gradients are then exchanged in PackWeightsFn.backward, not by the producers)
            self._cur_meta = c[1]
            return c[2], c[3]
        ws, entries, groups = [], [], []

        def add(w, K, C, T, scale, group=None, col=0):
            if group is None:
                groups.append((T * C, A.ops.round_up(K, 4)))
                group = len(groups) - 1
            ws.append(w)
            entries.append((K, C, T, scale, group, col))
            return group

        idx = {}
        rgb = self.layers[0][0]
        idx['rgb'] = add(rgb.weight, rgb.weight.shape[0], 3, 1, rgb.scale)
        for bi, blk in enumerate(list(self.layers)[1:]):
            for name, seq, ci in (('conv1', blk.conv1, 0), ('conv2', blk.conv2, 1), ('skip', blk.skip, 1)):
                m = seq[ci]
                K, C, k, _ = m.weight.shape
                idx[(bi, name)] = add(m.weight, K, C, k * k,
                                      m.scale / math.sqrt(2.0) if (fused and name == 'skip') else m.scale)
        m = self.last_conv[0]
        cpad = A.ops.round_up(self.c_last_in + 1, 16)
        w_last = F.pad(m.weight, (0, 0, 0, 0, 0, cpad - (self.c_last_in + 1)))     # zero rows for the pad channels
        idx['last'] = add(w_last, m.weight.shape[0], cpad, 9, m.scale)
        dh, dp, T = self.d_hidden, self.d_project, 16
        idx['l1'] = add(self.linear.l1.weight, dh, 512, T, 1.0)
        groups.append((self.n_features, 2 * dh))                                    # projection.0 | projection2.0
        gm = len(groups) - 1
        add(self.projection[0].weight, dh, 512, T, 1.0, gm, 0)
        add(self.projection2[0].weight, dh, 512, T, 1.0, gm, dh)
        idx['p0q0'] = gm
        idx['l2'] = add(self.linear.l2.weight, 1, dh, 1, 1.0)
        idx['p2'] = add(self.projection[2].weight, dp, dh, 1, 1.0)
        idx['q2'] = add(self.projection2[2].weight, dp, dh, 1, 1.0)
        meta = A.PackMeta(entries, groups)
        meta.comm = comm            # PackWeightsFn.backward then exchanges the packed gradients (data parallel)
        owner = weakref.ref(self)
        # once its backward has run the pack must not be referenced any more: a graph kept alive across a hipGraph capture
        # (its AccumulateGrad nodes are tied to the stream it ran on) crashes hipStreamEndCapture
        meta.on_backward = lambda: (_PACK_CACHE.pop(owner(), None) if owner() is not None else None)
        packed = A.PackWeightsFn.apply(meta, *ws)
        self._cur_meta = meta
        _PACK_CACHE[self] = (key, meta, packed, idx)
        return packed, idx

    # ---- forward ---------------------------------------------------------------------------------------
    def _trunk_fused(self, images, wp, idx, rec=None, r1=False):
        wb = [wp[idx['rgb']], self.layers[0][1].bias]
        for bi, blk in enumerate(list(self.layers)[1:]):
            wb += [wp[idx[(bi, 'conv1')]], blk.conv1[1].bias, wp[idx[(bi, 'conv2')]], blk.conv2[2].bias,
                   wp[idx[(bi, 'skip')]]]
        blur = list(self.layers)[1].conv2[0].kernel
        # exchange handles for the backward: group index of every packed weight among ``wb`` (None for the biases)
        gids = [idx['rgb'], None]
        for bi in range(len(self.layers) - 1):
            gids += [idx[(bi, 'conv1')], None, idx[(bi, 'conv2')], None, idx[(bi, 'skip')]]
        self._trunk_xch = (self._cur_meta.comm, self._cur_meta, gids) if (self._cur_meta.comm is not None
                                                                             and not r1) else None
        x = (_TrunkR1Fn if r1 else _TrunkFn).apply(self, images, blur, *wb)
        if rec is not None:
            rec.extend(self._trunk_rec)
        x = minibatch_stddev_batches(x, self._batch_splits)
        x = A.ConvBiasActFn.apply(x, wp[idx['last']], self.last_conv[1].bias,
                                  (self.last_conv[0].weight.shape[0], 3, 3, 1, 1), _SLOPE, _GAIN,
                                  *self._xch(idx['last'], not r1))
        if rec is not None:
            rec.append(x)
        return x

    def _xch(self, group, first_order=True):
        """Exchange handle of a packed weight for a node whose backward is its ONLY gradient producer (the first-order
        call: fused trunk, constant images); () otherwise."""
        m = self._cur_meta
        return ((m.comm, m, group),) if (first_order and m is not None and m.comm is not None) else ()

    def _trunk(self, images, wp, idx, rec=None):
        rgb = self.layers[0]
        K0 = rgb[0].weight.shape[0]
        x = A.RgbConvBiasActFn.apply(images, wp[idx['rgb']], rgb[1].bias, K0, (1, 2.0, -1.0), _SLOPE, _GAIN)
        if rec is not None:
            rec.append(x)
        for bi, blk in enumerate(list(self.layers)[1:]):
            ci, co = blk.cin, blk.cout
            o = A.ConvBiasActFn.apply(x, wp[idx[(bi, 'conv1')]], blk.conv1[1].bias, (ci, 3, 3, 1, 1), _SLOPE, _GAIN)
            if rec is not None:
                rec.append(o)
            p0, p1 = blk.conv2[0].pad
            o = A.UpFirDn2dFn.apply(o, blk.conv2[0].kernel, 1, 1, (p0, p1, p0, p1))
            # gain sqrt2 * (1/sqrt2 of the merge) = 1; the packed skip weights carry the other 1/sqrt2 (_pack)
            o = A.ConvBiasActFn.apply(o, wp[idx[(bi, 'conv2')]], blk.conv2[2].bias, (co, 3, 3, 2, 0), _SLOPE, 1.0)
            if rec is not None:
                rec.append(o)
            p0, p1 = blk.skip[0].pad
            # skip = Blur -> 1x1 stride-2 conv (discriminator.py:60-63 of the reference).  A 1x1 stride-2 conv only ever
            # reads the even pixels of its input, so the blur is evaluated at those pixels only (down = 2: a quarter of
            # the outputs, a quarter of the bytes written and re-read) and the conv runs at stride 1 -- the same sums.
            s = A.UpFirDn2dFn.apply(x, blk.skip[0].kernel, 1, 2, (p0, p1, p0, p1))
            s = A.Conv2dFn.apply(s, wp[idx[(bi, 'skip')]], (co, 1, 1, 1, 0))
            x = A.LinCombFn.apply(o, s, 1.0, 1.0)
        x = minibatch_stddev_batches(x, self._batch_splits)
        x = A.ConvBiasActFn.apply(x, wp[idx['last']], self.last_conv[1].bias,
                                  (self.last_conv[0].weight.shape[0], 3, 3, 1, 1), _SLOPE, _GAIN)
        if rec is not None:
            rec.append(x)
        return x                                                                     # (B,4,4,512) NHWC

    # BaseDiscriminator.forward: call _run(..., want_proj=projection or projection2).  The reference evaluates both
    # projection heads in every call and adds 0 * their means to the logits (base.py:139-141, so that DDP sees every
    # parameter used); here a call that does not ask for them (R1's D(x), the generator step of train_stylegan2.py)
    # skips their GEMMs; the parameters still receive their exact-zero gradients (A.ZeroGradFn / the unpack node).
    _lazy_projections = True

    def _run(self, inputs, sg_linear, finetuning, want_features, want_proj=True):
        if not inputs.is_cuda:
            raise RuntimeError('contrad_amd.ResidualDiscriminatorP runs on the MI355X HIP path only (no CPU fallback)')
        # constant images (every ContraD discriminator call): one fused first-order node; images that need a gradient
        # (R1's create_graph, the generator step): the any-order node family
        needs_img = inputs.requires_grad and torch.is_grad_enabled()
        fused = self.fuse_trunk and not needs_img and len(self.layers) > 1
        # images that need a gradient: _TrunkR1Fn (whose d / d images is differentiable again), round 3; the any-order
        # node family remains as the cross-check (fuse_r1 = False, tests)
        fused_r1 = self.fuse_trunk and self.fuse_r1 and needs_img and len(self.layers) > 1
        wp, idx = self._pack(True)
        images = inputs.contiguous().float()
        rec = [] if getattr(self, '_record_activations', False) else None     # test hook (linear regions used)
        trunk = (lambda *a: self._trunk_fused(*a, r1=True)) if fused_r1 else (self._trunk_fused if fused else self._trunk)
        if finetuning:
            with torch.no_grad():
                x = trunk(images, wp, idx, rec)
            x = x.detach()
        else:
            x = trunk(images, wp, idx, rec)
        B = x.shape[0]
        dh, dp = self.d_hidden, self.d_project
        feat = x.reshape(B, 1, 1, self.n_features)
        feat_d = feat.detach() if sg_linear else feat
        g1 = (1, 1, 1, 1, 0)
        # (first-order call: each head weight has exactly one gradient producer -> it may start the data-parallel
        # exchange of its packed gradient itself, _xch)
        h_l = A.ConvBiasActFn.apply(feat_d, wp[idx['l1']], self.linear.l1.bias, (dh,) + g1[1:], _HEAD_SLOPE, 1.0,
                                    *self._xch(idx['l1'], fused))
        out = A.ConvBiasActFn.apply(h_l, wp[idx['l2']], self.linear.l2.bias, (1,) + g1[1:], 1.0, 1.0,
                                    *self._xch(idx['l2'], fused)).view(B, 1)
        h_pq = proj = proj2 = None
        if want_proj or rec is not None:
            bias_pq = torch.cat([self.projection[0].bias, self.projection2[0].bias])
            h_pq = A.ConvBiasActFn.apply(feat, wp[idx['p0q0']], bias_pq, (2 * dh,) + g1[1:], _HEAD_SLOPE, 1.0,
                                         *self._xch(idx['p0q0'], fused))
            proj = A.ConvBiasActFn.apply(h_pq[..., :dh], wp[idx['p2']], self.projection[2].bias, (dp,) + g1[1:], 1.0,
                                         1.0, *self._xch(idx['p2'], fused)).view(B, dp)
            proj2 = A.ConvBiasActFn.apply(h_pq[..., dh:], wp[idx['q2']], self.projection2[2].bias, (dp,) + g1[1:], 1.0,
                                          1.0, *self._xch(idx['q2'], fused)).view(B, dp)
        else:
            # skipped heads: their packed weights get zeros from the unpack node; the four biases get them here, so that
            # EVERY parameter carries a gradient after any call (base.py:139-141) and the Adam state stays uniform
            out = A.ZeroGradFn.apply(out, self.projection[0].bias, self.projection[2].bias, self.projection2[0].bias,
                                     self.projection2[2].bias)
        if rec is not None:
            if not hasattr(self, '_recorded'):
                self._recorded = []
            self._recorded.append((rec, h_l, h_pq))
        feats = x.permute(0, 3, 1, 2).reshape(B, -1) if want_features else None      # NCHW-flat like the reference
        return out, proj, proj2, feats

    def penultimate(self, inputs):
        return self._run(inputs, False, False, True)[3]
