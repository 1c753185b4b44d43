"""Counterpart of the reference's ``augment`` package for the hot-path pipelines: ``get_augment('simclr')`` and
``get_augment('simclr_hq')`` (augment/__init__.py:13-28,106-122).

The reference composes ~25 PyTorch ops per call (two grid_samples, HSV round trip, blends); here the random
parameters are drawn on the HOST in the reference's exact RNG order (numpy global RNG for geometry / op order /
blur sigma, torch CPU generator for masks and colour factors -- SURVEY.md 8a row A7), shipped as one small
(B, 16) tensor, and a single fused HIP kernel does crop+flip+jitter+gray (plus a separable blur for *_hq).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..hostio import upload
from ..config import configurable, get_bindings

__all__ = ['get_augment', 'SimCLRAugment', 'simclr', 'simclr_hq', 'simclr_hq_cutout']


def _jitter_range(value, center=1.0, clip_first_on_zero=True):
    """Scalar branch of ColorJitterLayer._check_input (augment/color_jitter.py:25-42)."""
    if value < 0:
        raise ValueError('jitter strength must be non negative')
    lo, hi = center - value, center + value
    if clip_first_on_zero:
        lo = max(lo, 0)
    if lo == hi == center:
        return None
    return [lo, hi]


class _SimCLRFn(torch.autograd.Function):
    """Differentiable (w.r.t. the images) fused augmentation for the generator step: gradient flows through the
    bilinear crop/flip gather, the contrast stage, -- as an identity, like RandomHSVFunction.backward
    (augment/color_jitter.py:97-104) -- through the HSV jitter, and through the masked Gaussian blur of simclr_hq
    (its reflect-padding transpose)."""

    @staticmethod
    def forward(ctx, x, Pd, contrast_first, has_contrast, blur, cutout):
        out = ops.simclr_augment(x, Pd, contrast_first, has_contrast)
        if blur is not None:
            radius, g = blur
            out = ops.gaussian_blur_masked(out, Pd, g, radius)
            ctx.save_for_backward(x, Pd, g)
        else:
            ctx.save_for_backward(x, Pd)
        if cutout is not None:
            ops.cutout_masked_(out, Pd, cutout)
        ctx.cfg = (contrast_first, has_contrast, None if blur is None else blur[0], cutout)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        cf, hc, radius, cutout = ctx.cfg
        if cutout is not None:
            g = ops.cutout_masked_(g.contiguous().clone(), ctx.saved_tensors[1], cutout)
        if radius is not None:
            x, Pd, gk = ctx.saved_tensors
            g = ops.gaussian_blur_masked_bwd(g, Pd, gk, radius)
        else:
            x, Pd = ctx.saved_tensors
        return ops.simclr_augment_bwd(x, Pd, g, cf, hc), None, None, None, None, None


class SimCLRAugment(nn.Module):
    """RandomResizeCrop -> HorizontalFlip -> RandomApply(ColorJitter, 0.8) -> RandomApply(Gray, 0.2)
    [-> RandomApply(GaussianBlur, 0.5)] [-> RandomApply(CutOut, 0.5)] as one module; maps NCHW float [0,1] to the same
    shape."""

    def __init__(self, scale, ratio=(3. / 4., 4. / 3.), brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1,
                 p_jitter=0.8, p_gray=0.2, p_blur=None, sigma_range=None, p_cutout=None, cutout_length=None):
        super().__init__()
        self.scale, self.ratio = tuple(scale), tuple(ratio)
        self.r_v = _jitter_range(brightness)
        self.r_c = _jitter_range(contrast)
        self.r_s = _jitter_range(saturation)
        self.r_h = _jitter_range(hue, center=0, clip_first_on_zero=False)
        if self.r_h is not None and not (-0.5 <= self.r_h[0] <= self.r_h[1] <= 0.5):
            raise ValueError('hue values should be between (-0.5, 0.5)')
        self.p_jitter, self.p_gray, self.p_blur = p_jitter, p_gray, p_blur
        self.sigma_range = sigma_range
        self.p_cutout, self.cutout_length = p_cutout, cutout_length
        if p_cutout is not None and (cutout_length is None or cutout_length % 2 == 0):
            raise ValueError("Currently CutOut only accepts odd lengths: length % 2 == 1")        # spatial.py:156-157

    supports_out = True          # forward(x, out=...): the last stage writes into the caller's buffer (no-grad path)

    # ---- host-side sampling, reference draw order ----
    def sample(self, B, dim2, dim3):
        """Returns (params (B,12) CPU float tensor, contrast_first, sigma or None)."""
        P = torch.zeros(B, ops.AUG_NPARAM)
        width, height = dim2, dim3          # the reference's naming of shape[2], shape[3] (spatial.py:113)
        area = height * width
        target_area = np.random.uniform(*self.scale, B * 10) * area
        log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        aspect_ratio = np.exp(np.random.uniform(*log_ratio, B * 10))
        w = np.round(np.sqrt(target_area * aspect_ratio))
        h = np.round(np.sqrt(target_area / aspect_ratio))
        keep = (0 < w) * (w <= width) * (0 < h) * (h <= height)
        w, h = w[keep], h[keep]
        if len(w) > B:
            pick = np.random.choice(len(w), B, replace=False)
            w, h = w[pick], h[pick]
        n = len(w)
        bias_w = np.random.randint(w - width, width - w + 1) / width
        bias_h = np.random.randint(h - height, height - h + 1) / height
        P[:, 0] = 1.0
        P[:, 1] = 1.0
        P[:n, 0] = torch.tensor(w / width)
        P[:n, 1] = torch.tensor(h / height)
        P[:n, 2] = torch.tensor(bias_w)
        P[:n, 3] = torch.tensor(bias_h)
        P[:, 4] = torch.bernoulli(torch.ones(B) * 0.5) * 2 - 1
        P[:, 5] = torch.bernoulli(torch.full((B,), self.p_jitter))
        contrast_first = bool(np.random.rand() > 0.5)

        def draw_contrast():
            P[:, 6] = torch.empty(B, 1, 1, 1).uniform_(*self.r_c).view(B) if self.r_c else 1.0

        def draw_hsv():
            f_h, f_s, f_v = torch.zeros(B, 1, 1), torch.ones(B, 1, 1), torch.ones(B, 1, 1)
            if self.r_h:
                f_h.uniform_(*self.r_h)
            if self.r_s:
                f_s.uniform_(*self.r_s)
            if self.r_v:
                f_v.uniform_(*self.r_v)
            P[:, 7], P[:, 8], P[:, 9] = f_h.view(B), f_s.view(B), f_v.view(B)

        if contrast_first:
            draw_contrast(); draw_hsv()
        else:
            draw_hsv(); draw_contrast()
        P[:, 10] = torch.bernoulli(torch.full((B,), self.p_gray))
        sigma = None
        if self.p_blur is not None:
            P[:, 11] = torch.bernoulli(torch.full((B,), self.p_blur))
            sigma = float(np.random.uniform(*self.sigma_range))
        P[:, 15] = float(contrast_first)    # also carried in the block: a captured hipGraph reads it from there
        if self.p_cutout is not None:        # RandomApply mask, then CutOut's two randint draws (spatial.py:166-170)
            P[:, 12] = torch.bernoulli(torch.full((B,), self.p_cutout))
            P[:, 13] = torch.randint(dim2, (B, 1)).view(B).float()
            P[:, 14] = torch.randint(dim3, (B, 1)).view(B).float()
        return P, contrast_first, sigma

    @staticmethod
    def blur_kernel(H, sigma):
        """GaussianBlur's kernel size (augment/__init__.py:69-71) and kornia's normalised 1-D Gaussian."""
        radius = int((H // 10) / 2)
        xs = torch.arange(2 * radius + 1, dtype=torch.float32) - radius
        g = torch.exp(-xs.pow(2) / (2 * sigma ** 2))
        return radius, g / g.sum()

    def apply(self, inputs, P, contrast_first, sigma=None, out=None):
        """Deterministic device part."""
        Pd = upload(P, inputs.device)      # (B,12) parameter block (asynchronous pinned upload, contrad_amd/hostio.py)
        if inputs.requires_grad and torch.is_grad_enabled():
            blur = None
            if sigma is not None:
                radius, g = self.blur_kernel(inputs.shape[2], sigma)
                if radius > 0:
                    blur = (radius, g.to(inputs.device))
            if out is not None:
                raise RuntimeError('augmentation: out= is a forward-only option')
            return _SimCLRFn.apply(inputs.contiguous().float(), Pd, contrast_first, self.r_c is not None, blur,
                                   self.cutout_length if self.p_cutout is not None else None)
        x = inputs.detach().contiguous().float()
        radius = 0
        if sigma is not None:
            radius, g = self.blur_kernel(x.shape[2], sigma)
        # ``out``: the LAST stage writes there (a caller's slice of a larger batch buffer: no concatenation afterwards)
        res = ops.simclr_augment(x, Pd, contrast_first, self.r_c is not None, out=None if radius > 0 else out)
        if radius > 0:
            res = ops.gaussian_blur_masked(res, Pd, g.to(x.device), radius, out=out)
        if self.p_cutout is not None:
            ops.cutout_masked_(res, Pd, self.cutout_length)
        return res

    def forward(self, inputs, out=None):
        if not inputs.is_cuda:
            raise RuntimeError('contrad_amd augmentation runs on the MI355X HIP path only (no CPU fallback)')
        B, _, d2, d3 = inputs.shape
        P, contrast_first, sigma = self.sample(B, d2, d3)
        return self.apply(inputs, P, contrast_first, sigma, out=out)


def _kwargs_from_bindings():
    cj = get_bindings('ColorJitterLayer')
    rc = get_bindings('RandomResizeCropLayer')
    return dict(scale=rc['scale'], ratio=rc.get('ratio', (3. / 4., 4. / 3.)),
                brightness=cj['brightness'], contrast=cj['contrast'], saturation=cj['saturation'], hue=cj['hue'])


def simclr():
    return SimCLRAugment(**_kwargs_from_bindings())


def simclr_hq():
    gb = get_bindings('GaussianBlur')
    return SimCLRAugment(p_blur=0.5, sigma_range=gb['sigma_range'], **_kwargs_from_bindings())


def simclr_hq_cutout():
    gb = get_bindings('GaussianBlur')
    return SimCLRAugment(p_blur=0.5, sigma_range=gb['sigma_range'], p_cutout=0.5,
                         cutout_length=get_bindings('CutOut')['length'], **_kwargs_from_bindings())


@configurable('augment')
def get_augment(mode='none', **kwargs):
    """Same entry point as augment.get_augment (augment/__init__.py:13-28) for the modes on the hot path."""
    mapping = {'simclr': simclr, 'simclr_hq': simclr_hq, 'simclr_hq_cutout': simclr_hq_cutout}
    if mode not in mapping:
        raise NotImplementedError("augmentation mode '%s' is outside the ContraD hot path (SURVEY.md 2 row 6)" % mode)
    return mapping[mode]()
