"""Round 5, call s: why does the 3x3 stride-1 FWD conv of ResidualDiscriminatorP(512) run 8 - 15 % slower inside the step
(rocprofv3 rows: 1.86 ms at 128 channels / 128x128, 48 images) than alone (tools/bench_conv.py: 1.62 ms)?  Times the conv
alone through events placed directly around it, in several contexts: random / activation-like input, with the in-step
epilogue (bias, slope 0.2, gain sqrt2), and preceded in every iteration by the kernel that produces its input in the step
(the previous block's 1x1 skip conv with the residual addend) or by an HBM-bound pass over a tensor of the same size."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from contrad_amd import ops

dev = torch.device('cuda')
B = 48


def timed(conv, before=None, iters=12, warm=3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for _ in range(warm):
        if before:
            before()
        conv()
    torch.cuda.synchronize()
    for e0, e1 in ev:
        if before:
            before()
        e0.record()
        conv()
        e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2], ts[0], ts[-1]


for (H, C) in ((128, 128), (256, 64), (64, 256)):
    K = C
    flops = 2.0 * B * H * H * K * C * 9
    x = torch.randn(B, H, H, C, device=dev)
    xa = torch.nn.functional.leaky_relu(torch.randn(B, H, H, C, device=dev), 0.2) * math.sqrt(2.0)
    wp = torch.randn(9 * C, K, device=dev) * 0.05
    bias = torch.randn(K, device=dev) * 0.1
    y = torch.empty(B, H, H, K, device=dev)
    gy = torch.randn(B, H, H, K, device=dev)
    dx = torch.empty(B, H, H, C, device=dev)
    # the step's producer of x: 1x1 skip conv of the previous block (C/2 -> C at 2H... here same-size stand-in: C -> C at H) + addend
    sb = torch.randn(B, H, H, C // 2, device=dev)
    wps = torch.randn(C // 2, C, device=dev) * 0.05
    y2 = torch.randn(B, H, H, C, device=dev)
    xprod = torch.empty(B, H, H, C, device=dev)
    big = torch.randn(B, H, H, C, device=dev)
    big2 = torch.empty_like(big)

    def fwd_plain():
        ops.conv2d_fwd(x, wp, None, K, 3, 3, 1, 1, 0.1, 1.0, out=y)

    def fwd_step():
        ops.conv2d_fwd(x, wp, bias, K, 3, 3, 1, 1, 0.2, math.sqrt(2.0), out=y)

    def fwd_act():
        ops.conv2d_fwd(xa, wp, bias, K, 3, 3, 1, 1, 0.2, math.sqrt(2.0), out=y)

    def fwd_prod():
        ops.conv2d_fwd(xprod, wp, bias, K, 3, 3, 1, 1, 0.2, math.sqrt(2.0), out=y)

    def producer():
        ops.conv2d_fwd(sb, wps, None, C, 1, 1, 1, 0, addend=y2, out=xprod)

    def hbm_pass():
        torch.mul(big, 1.5, out=big2)

    def dgrad():
        ops.conv2d_dgrad(gy, wp, (B, H, H, C), 3, 3, 1, 1, out=dx)

    rows = [('fwd alone, random x, no bias (bench_conv)', fwd_plain, None),
            ('fwd alone, step epilogue (bias, 0.2, sqrt2)', fwd_step, None),
            ('fwd alone, activation-like x', fwd_act, None),
            ('fwd after its producer (1x1 skip conv + addend)', fwd_prod, producer),
            ('fwd after an HBM pass of the same size', fwd_step, hbm_pass),
            ('dgrad alone', dgrad, None),
            ('dgrad after an HBM pass of the same size', dgrad, hbm_pass)]
    for name, fn, before in rows:
        med, lo, hi = timed(fn, before)
        print('H%-3d C%-3d  %-50s median %.3f ms (%.3f .. %.3f)  %.1f TF/s' % (H, C, name, med, lo, hi, flops / med / 1e9),
              flush=True)
