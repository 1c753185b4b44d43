"""Dev tool: the HBM-bound kernels of the headline config (SNDCGAN, 3N = 1536 images at 32x32) and of the 512^2
StyleGAN2 config in isolation: time, algorithmic bytes, achieved GB/s against 8 TB/s (6.3 TB/s achievable)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd import ops
from contrad_amd import autograd_ops as A

dev = torch.device('cuda')


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def report(name, t, nbytes):
    print('%-44s %8.1f us  %8.1f MB  %6.2f TB/s  (%4.1f %% of 8.0)' % (name, t * 1e6, nbytes / 1e6, nbytes / t / 1e12,
                                                                     100 * nbytes / t / 8e12), flush=True)


def c10():
    B, H = 1536, 32
    img = torch.rand(B, 3, H, H, device=dev)
    w = torch.randn(64, 3, 3, 3) * 0.05
    wp = ops.pack_weight(w).to(dev)
    bias = torch.zeros(64, device=dev)
    y = torch.empty(B, H, H, 64, device=dev)
    report('rgb_conv_fwd<3,3>  (D first conv)', timeit(lambda: ops.rgb_conv_fwd(img, wp, bias, 64, 3, 2.0, -1.0, 0.1, 1.0, out=y)),
           img.numel() * 4 + y.numel() * 4)
    gy = torch.randn_like(y)
    dwp = torch.zeros_like(wp); db = torch.empty(64, device=dev)
    report('rgb_conv_wgrad<3,3>', timeit(lambda: ops.rgb_conv_wgrad(img, gy, 3, 2.0, -1.0, dwp, db)), img.numel() * 4 + gy.numel() * 4)
    # generator's last ConvT 64 -> 3 + tanh at N = 512
    x = torch.randn(512, H, H, 64, device=dev)
    wt = ops.pack_weight((torch.randn(64, 3, 3, 3) * 0.05)).to(dev)         # rows (tap, c=3), cols k=64
    wt = torch.randn(27, 64, device=dev) * 0.05
    out = torch.empty(512, 3, H, H, device=dev)
    b3 = torch.zeros(3, device=dev)
    report('rgb_conv_dgrad k3 + tanh (G last layer)', timeit(lambda: ops.rgb_conv_dgrad(x, wt, b3, 3, 3, act=1, out_scale=0.5, out_shift=0.5, out=out)),
           x.numel() * 4 + out.numel() * 4)
    P = torch.zeros(B, ops.AUG_NPARAM, device=dev); P[:, 0] = 0.8; P[:, 1] = 0.7; P[:, 4] = 1; P[:, 5] = 1; P[:, 6] = 1.1
    P[:, 8] = 1.0; P[:, 9] = 1.0
    o = torch.empty_like(img)
    report('simclr_small (augment 1536 x 32x32)', timeit(lambda: ops.simclr_augment(img, P, True, True, out=o)), 2 * img.numel() * 4)
    # Adam over D's 18.6 M parameters
    n = 18568961
    p, g, m, v = [torch.randn(n, device=dev) for _ in range(4)]
    v.abs_()
    report('adam (18.6 M params)', timeit(lambda: ops.adam_step([p], [g], [m], [v], 3, 2e-4, 0.5, 0.999)), 28.0 * n)


def sg2():
    B, H, C = 48, 512, 32
    k = A.make_blur_kernel().to(dev)
    x = torch.randn(B, H, H, C, device=dev)
    y = ops.upfirdn2d(x, k, 1, 1, (2, 2, 2, 2))
    report('upfirdn 4x4 u1d1 pad2 (48x512^2x32)', timeit(lambda: ops.upfirdn2d(x, k, 1, 1, (2, 2, 2, 2)), 5, 2), (x.numel() + y.numel()) * 4)
    report('upfirdn 4x4 u1d1 bwd + act (dual)', timeit(lambda: ops.upfirdn2d_fused(y, k, 1, 1, (1, 1, 1, 1), act_ref=x, slope=0.2, gain=1.4, want_out=False, want_out2=True), 5, 2),
           (y.numel() + 2 * x.numel()) * 4)
    s = ops.upfirdn2d(x, k, 1, 2, (1, 1, 1, 1))
    report('upfirdn 4x4 u1d2 (skip blur+decimate)', timeit(lambda: ops.upfirdn2d(x, k, 1, 2, (1, 1, 1, 1)), 5, 2), (x.numel() + s.numel()) * 4)
    add = torch.randn_like(x)
    report('upfirdn 4x4 u2d1 + addend + dual out', timeit(lambda: ops.upfirdn2d_fused(s, k, 2, 1, (2, 1, 2, 1), addend=add, act_ref=x, slope=0.2, gain=1.0, want_out=True, want_out2=True), 5, 2),
           (s.numel() + 4 * x.numel()) * 4)
    img = torch.rand(B, 3, H, H, device=dev)
    wp = ops.pack_weight(torch.randn(32, 3, 1, 1)).to(dev)
    yy = torch.empty(B, H, H, 32, device=dev)
    report('rgb_conv_fwd<1,3> FromRGB 512^2', timeit(lambda: ops.rgb_conv_fwd(img, wp, torch.zeros(32, device=dev), 32, 1, 2.0, -1.0, 0.2, 1.4, out=yy), 5, 2),
           (img.numel() + yy.numel()) * 4)
    dwp = torch.zeros_like(wp); db = torch.empty(32, device=dev)
    report('rgb_conv_wgrad<1,3> 512^2', timeit(lambda: ops.rgb_conv_wgrad(img, yy, 1, 2.0, -1.0, dwp, db), 5, 2), (img.numel() + yy.numel()) * 4)
    z = torch.randn_like(yy)
    report('lincomb', timeit(lambda: ops.lincomb(yy, z, 1.0, 1.0), 5, 2), 3 * yy.numel() * 4)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('c10', 'all'):
        c10()
    if which in ('sg2', 'all'):
        sg2()
