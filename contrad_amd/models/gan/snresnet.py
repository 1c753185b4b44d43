"""D_SNResNet18 on the HIP kernel library -- counterpart of models/gan/snresnet.py:21-89 (scope row N4).

ResNet-18 of BasicBlocks with LeakyReLU(0.1) and spectral norm on every Conv2d / Linear, NO BatchNorm
(snresnet.py:21-41,59-66); ``penultimate`` = x*2-1 -> conv3x3 -> 4 stages x 2 blocks -> 4x4 average pool -> 512
features (snresnet.py:77-89); heads from BaseDiscriminator with d_hidden = 1024 (models/gan/__init__.py:8-12).
Composed from the differentiable HIP nodes of contrad_amd.autograd_ops: all 20 convolutions + 6 linears share ONE
batched spectral-norm launch (SnPackWeightsFn), bias + LeakyReLU sit in the conv epilogues, activations are NHWC.
State-dict names and order match the reference.
"""
import torch
import torch.nn as nn

from ... import autograd_ops as A
from .base import BaseDiscriminator, SNParams, TinyHead, make_projection

_SLOPE = 0.1
_STAGES = [(64, 1), (128, 2), (256, 2), (512, 2)]        # (planes, stride of the first block); [2, 2, 2, 2] blocks


class _BasicBlock(nn.Module):
    def __init__(self, in_planes, planes, stride):
        super().__init__()
        self.conv1 = SNParams((planes, in_planes, 3, 3))
        self.conv2 = SNParams((planes, planes, 3, 3))
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(SNParams((planes, in_planes, 1, 1)))
        self.in_planes, self.planes, self.stride = in_planes, planes, stride


class D_SNResNet18(BaseDiscriminator):
    """Drop-in for D_SNResNet18(mlp_linear=True, d_hidden=1024)."""

    def __init__(self, n_classes=1, disable_sn=False, mlp_linear=True, d_hidden=1024, d_project=128):
        super().__init__()
        if n_classes != 1 or disable_sn or not mlp_linear:
            raise NotImplementedError('only the configuration built by get_architecture("snresnet18") is implemented')
        self.n_features = 512
        self.d_penul = 512
        self.n_classes, self.d_hidden, self.d_project = n_classes, d_hidden, d_project
        self.linear = TinyHead(512, d_hidden, spectral=True)
        self.projection = make_projection(512, d_hidden, d_project, spectral=True)
        self.projection2 = make_projection(512, d_hidden, d_project, spectral=True)
        self.conv1 = SNParams((64, 3, 3, 3))
        in_planes = 64
        for li, (planes, stride) in enumerate(_STAGES, 1):
            blocks = []
            for s in (stride, 1):
                blocks.append(_BasicBlock(in_planes, planes, s))
                in_planes = planes
            setattr(self, 'layer%d' % li, nn.Sequential(*blocks))
        # the reference's reset_parameters (base.py:152-164) re-runs nn.Conv2d / nn.Linear's default init; SNParams'
        # N(0, 0.02) is D_SNDCGAN's.  Use the default (kaiming-uniform) init here, as the reference's constructor does.
        for m in self.modules():
            if isinstance(m, SNParams):
                _default_init(m)

    def reset_parameters(self, root=None):
        root = self if root is None else root
        for m in root.modules():
            if isinstance(m, SNParams):
                _default_init(m)

    def _sn_modules(self):
        mods = [self.conv1]
        for li in range(1, 5):
            for blk in getattr(self, 'layer%d' % li):
                mods += [blk.conv1, blk.conv2] + list(blk.shortcut)
        return mods + [self.linear.l1, self.linear.l2, self.projection[0], self.projection[2], self.projection2[0],
                       self.projection2[2]]

    def _trunk(self, images, wp):
        rec = self._recorded = [] if getattr(self, '_record_activations', False) else None   # test hook: linear regions used
        x = A.RgbConvBiasActFn.apply(images, wp[self.conv1], self.conv1.bias, 64, (3, 2.0, -1.0), _SLOPE, 1.0)
        if rec is not None:
            rec.append(x.detach())
        for li in range(1, 5):
            for blk in getattr(self, 'layer%d' % li):
                p, s = blk.planes, blk.stride
                o = A.ConvBiasActFn.apply(x, wp[blk.conv1], blk.conv1.bias, (p, 3, 3, s, 1), _SLOPE, 1.0)
                if rec is not None:
                    rec.append(o.detach())
                o = A.ConvBiasActFn.apply(o, wp[blk.conv2], blk.conv2.bias, (p, 3, 3, 1, 1), 1.0, 1.0)
                sc = x
                if len(blk.shortcut):
                    m = blk.shortcut[0]
                    sc = A.ConvBiasActFn.apply(x, wp[m], m.bias, (p, 1, 1, s, 0), 1.0, 1.0)
                x = A.ActFn.apply(A.LinCombFn.apply(o, sc, 1.0, 1.0), _SLOPE, 1.0)
                if rec is not None:
                    rec.append(x.detach())
        if x.shape[1] != 4 or x.shape[2] != 4:
            raise NotImplementedError('D_SNResNet18: 32x32 inputs (avg_pool2d(4) over the final 4x4 map, snresnet.py:86)')
        return x.mean((1, 2))                                     # (B, 512); tiny

    def _run(self, inputs, sg_linear, finetuning, want_features):
        if not inputs.is_cuda:
            raise RuntimeError('contrad_amd.D_SNResNet18 runs on the MI355X HIP path only (no CPU fallback)')
        mods = self._sn_modules()
        if finetuning:
            # base.py:114-119: the features are computed in eval mode (no power iteration in the trunk) under no_grad,
            # the heads stay in the caller's mode -> two weight-prep launches
            trunk, heads = mods[:-6], mods[-6:]
            with torch.no_grad():
                pt = A.SnPackWeightsFn.apply(trunk, False, *[m.weight_orig for m in trunk])
            ph = A.SnPackWeightsFn.apply(heads, self.training, *[m.weight_orig for m in heads])
            wp = dict(zip(trunk + heads, tuple(pt) + tuple(ph)))
        else:
            packed = A.SnPackWeightsFn.apply(mods, self.training, *[m.weight_orig for m in mods])
            wp = dict(zip(mods, packed))
        images = inputs.contiguous().float()
        if finetuning:
            with torch.no_grad():
                feat = self._trunk(images, wp)
            feat = feat.detach()
        else:
            feat = self._trunk(images, wp)
        B = feat.shape[0]
        dh, dp = self.d_hidden, self.d_project
        f4 = feat.reshape(B, 1, 1, 512)
        fd = f4.detach() if sg_linear else f4

        def lin(m, t, K, slope):
            return A.ConvBiasActFn.apply(t, wp[m], m.bias, (K, 1, 1, 1, 0), slope, 1.0)

        h_l, h_p, h_p2 = lin(self.linear.l1, fd, dh, _SLOPE), lin(self.projection[0], f4, dh, _SLOPE), \
            lin(self.projection2[0], f4, dh, _SLOPE)
        if getattr(self, '_record_activations', False):
            self._recorded_heads = tuple(t.detach().reshape(B, dh) for t in (h_l, h_p, h_p2))
        out = lin(self.linear.l2, h_l, 1, 1.0).view(B, 1)
        proj = lin(self.projection[2], h_p, dp, 1.0).view(B, dp)
        proj2 = lin(self.projection2[2], h_p2, dp, 1.0).view(B, dp)
        return out, proj, proj2, (feat if want_features else None)

    def penultimate(self, inputs):
        return self._run(inputs, False, False, True)[3]


def _default_init(m):
    """nn.Conv2d / nn.Linear.reset_parameters: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight
    and bias; u, v as torch.nn.utils.spectral_norm draws them."""
    import math
    import torch.nn.functional as F
    fan_in = int(math.prod(m.weight_orig.shape[1:]))
    bound = 1.0 / math.sqrt(fan_in)
    with torch.no_grad():
        m.weight_orig.uniform_(-bound, bound)
        m.bias.uniform_(-bound, bound)
        m.weight_u.copy_(F.normalize(torch.randn(m.weight_u.shape), dim=0, eps=1e-12))
        m.weight_v.copy_(F.normalize(torch.randn(m.weight_v.shape), dim=0, eps=1e-12))
