"""Round 5, call t (follows fwd_context.py): does the in-step penalty of the FWD conv come from touching memory it has not
touched recently?  Cycles the conv over NB distinct (input, output) buffer pairs -- the input written by its in-step
producer right before -- instead of one fixed pair; NB x 0.8 GB walks far beyond any cache / TLB reach, like the forward
pass of a step does."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from contrad_amd import ops

dev = torch.device('cuda')
B, H, C = 48, 128, 128
K = C
flops = 2.0 * B * H * H * K * C * 9
wp = torch.randn(9 * C, K, device=dev) * 0.05
bias = torch.randn(K, device=dev) * 0.1
sb = torch.randn(B, H, H, C // 2, device=dev)
wps = torch.randn(C // 2, C, device=dev) * 0.05
y2 = torch.randn(B, H, H, C, device=dev)
gy = torch.randn(B, H, H, K, device=dev)


def run(NB, mode, produce):
    xs = [torch.randn(B, H, H, C, device=dev) for _ in range(NB)]
    ys = [torch.empty(B, H, H, K, device=dev) for _ in range(NB)]
    iters = max(16, 2 * NB)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for it in range(-NB, iters):
        i = it % NB
        if produce:
            ops.conv2d_fwd(sb, wps, None, C, 1, 1, 1, 0, addend=y2, out=xs[i])
        if it >= 0:
            ev[it][0].record()
        if mode == 'fwd':
            ops.conv2d_fwd(xs[i], wp, bias, K, 3, 3, 1, 1, 0.2, math.sqrt(2.0), out=ys[i])
        else:
            ops.conv2d_dgrad(xs[i], wp, (B, H, H, C), 3, 3, 1, 1, out=ys[i])
        if it >= 0:
            ev[it][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    med = ts[len(ts) // 2]
    print('%-5s NB=%-2d producer=%d  median %.3f ms (%.3f .. %.3f)  %.1f TF/s' % (mode, NB, produce, med, ts[0], ts[-1],
                                                                              flops / med / 1e9), flush=True)


for mode in ('fwd', 'dgrad'):
    for NB in (1, 4, 24):
        for produce in (0, 1):
            run(NB, mode, produce)
