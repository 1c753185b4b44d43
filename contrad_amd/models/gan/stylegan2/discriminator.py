"""ResidualDiscriminatorP (models/gan/stylegan2/discriminator.py:191-235) on the HIP library.

FromRGB 1x1 (+ input rescale x*2-1, bias, lrelu(0.2)*sqrt2 fused) -> ResBlocks [conv3x3+act; blur(2,2) -> conv3x3
stride 2 + act; skip: blur(1,1) -> conv1x1 stride 2; (out+skip)/sqrt2] -> minibatch-stddev channel -> last_conv ->
8192 features -> the BaseDiscriminator heads (plain nn.Linear, no spectral norm here).  Built from the
any-order-differentiable nodes of contrad_amd.autograd_ops, so ``autograd.grad(..., create_graph=True)`` -- the R1
penalty (train_stylegan2.py:106-113) -- works as it does in the reference.  Activations are NHWC internally, every
EqualConv2d scale 1/sqrt(fan_in) is folded into the packed weights, state-dict names match the reference.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import autograd_ops as A
from ..base import BaseDiscriminator, PlainParams, TinyHead, _Act, make_projection

_SLOPE, _GAIN = 0.2, math.sqrt(2.0)
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


def minibatch_stddev_nhwc(x, stddev_group=4):
    """_minibatch_stddev_layer (discriminator.py:22-33) on NHWC: one extra channel holding, for sample b, the
    mean over (C,H,W) of the std over the group {b mod M, + M, + 2M, ...}, M = B / group.  (Tiny (B,4,4,512)
    tensor; kept in differentiable torch ops so the R1 double backward through sqrt/var is exact.)  The channel
    dimension is padded to a multiple of 4 (zeros) for the float4 loaders of the conv engine."""
    B, H, W, C = x.shape
    group = min(B, stddev_group)
    y = x.reshape(group, B // group, H, W, C)
    std = torch.sqrt(y.var(0, unbiased=False) + 1e-8)
    std = std.mean([1, 2, 3], keepdim=True)                       # (M,1,1,1)
    std = std.repeat(group, H, W, 1)                              # (B,H,W,1)
    pad = (-(C + 1)) % 4
    parts = [x, std]
    if pad:
        parts.append(x.new_zeros(B, H, W, pad))
    return torch.cat(parts, dim=3)


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

    # ---- weight packing plan -------------------------------------------------------------------------
    def _pack(self):
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
                idx[(bi, name)] = add(m.weight, K, C, k * k, m.scale)
        m = self.last_conv[0]
        cpad = A.ops.round_up(self.c_last_in + 1, 4)
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
        packed = A.PackWeightsFn.apply(A.PackMeta(entries, groups), *ws)
        return packed, idx

    # ---- forward ---------------------------------------------------------------------------------------
    def _trunk(self, images, wp, idx, rec=None):
        rgb = self.layers[0]
        K0 = rgb[0].weight.shape[0]
        x = A.RgbConvBiasActFn.apply(images, wp[idx['rgb']], rgb[1].bias, K0, (1, 2.0, -1.0), _SLOPE, _GAIN)
        if rec is not None:
            rec.append(x)
        inv = 1.0 / math.sqrt(2.0)
        for bi, blk in enumerate(list(self.layers)[1:]):
            ci, co = blk.cin, blk.cout
            o = A.ConvBiasActFn.apply(x, wp[idx[(bi, 'conv1')]], blk.conv1[1].bias, (ci, 3, 3, 1, 1), _SLOPE, _GAIN)
            if rec is not None:
                rec.append(o)
            p0, p1 = blk.conv2[0].pad
            o = A.UpFirDn2dFn.apply(o, blk.conv2[0].kernel, 1, 1, (p0, p1, p0, p1))
            o = A.ConvBiasActFn.apply(o, wp[idx[(bi, 'conv2')]], blk.conv2[2].bias, (co, 3, 3, 2, 0), _SLOPE, _GAIN)
            if rec is not None:
                rec.append(o)
            p0, p1 = blk.skip[0].pad
            # skip = Blur -> 1x1 stride-2 conv (discriminator.py:60-63 of the reference).  A 1x1 stride-2 conv only ever
            # reads the even pixels of its input, so the blur is evaluated at those pixels only (down = 2: a quarter of
            # the outputs, a quarter of the bytes written and re-read) and the conv runs at stride 1 -- the same sums.
            s = A.UpFirDn2dFn.apply(x, blk.skip[0].kernel, 1, 2, (p0, p1, p0, p1))
            s = A.Conv2dFn.apply(s, wp[idx[(bi, 'skip')]], (co, 1, 1, 1, 0))
            x = A.LinCombFn.apply(o, s, inv, inv)
        x = minibatch_stddev_nhwc(x)
        x = A.ConvBiasActFn.apply(x, wp[idx['last']], self.last_conv[1].bias,
                                  (self.last_conv[0].weight.shape[0], 3, 3, 1, 1), _SLOPE, _GAIN)
        if rec is not None:
            rec.append(x)
        return x                                                                     # (B,4,4,512) NHWC

    def _run(self, inputs, sg_linear, finetuning, want_features):
        if not inputs.is_cuda:
            raise RuntimeError('contrad_amd.ResidualDiscriminatorP runs on the MI355X HIP path only (no CPU fallback)')
        wp, idx = self._pack()
        images = inputs.contiguous().float()
        rec = [] if getattr(self, '_record_activations', False) else None     # test hook (linear regions used)
        if finetuning:
            with torch.no_grad():
                x = self._trunk(images, wp, idx, rec)
            x = x.detach()
        else:
            x = self._trunk(images, wp, idx, rec)
        B = x.shape[0]
        dh, dp = self.d_hidden, self.d_project
        feat = x.reshape(B, 1, 1, self.n_features)
        feat_d = feat.detach() if sg_linear else feat
        g1 = (1, 1, 1, 1, 0)
        h_l = A.ConvBiasActFn.apply(feat_d, wp[idx['l1']], self.linear.l1.bias, (dh,) + g1[1:], _HEAD_SLOPE, 1.0)
        bias_pq = torch.cat([self.projection[0].bias, self.projection2[0].bias])
        h_pq = A.ConvBiasActFn.apply(feat, wp[idx['p0q0']], bias_pq, (2 * dh,) + g1[1:], _HEAD_SLOPE, 1.0)
        out = A.ConvBiasActFn.apply(h_l, wp[idx['l2']], self.linear.l2.bias, (1,) + g1[1:], 1.0, 1.0).view(B, 1)
        proj = A.ConvBiasActFn.apply(h_pq[..., :dh], wp[idx['p2']], self.projection[2].bias, (dp,) + g1[1:], 1.0,
                                     1.0).view(B, dp)
        proj2 = A.ConvBiasActFn.apply(h_pq[..., dh:], wp[idx['q2']], self.projection2[2].bias, (dp,) + g1[1:], 1.0,
                                      1.0).view(B, dp)
        if rec is not None:
            if not hasattr(self, '_recorded'):
                self._recorded = []
            self._recorded.append((rec, h_l, h_pq))
        feats = x.permute(0, 3, 1, 2).reshape(B, -1) if want_features else None      # NCHW-flat like the reference
        return out, proj, proj2, feats

    def penultimate(self, inputs):
        return self._run(inputs, False, False, True)[3]
