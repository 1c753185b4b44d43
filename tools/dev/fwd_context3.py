"""Round 5, call u (follows fwd_context2.py): does the FWD conv's time depend on WHERE its output lies relative to its input
(HBM channel / bank aliasing between the read and the write stream)?  x and y are carved out of one buffer at a swept
distance; 128 channels at 128 x 128, 48 images (403 MB each)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from contrad_amd import ops

dev = torch.device('cuda')
B, H, C = 48, 128, 128
K = C
n = B * H * H * C
flops = 2.0 * B * H * H * K * C * 9
wp = torch.randn(9 * C, K, device=dev) * 0.05
bias = torch.randn(K, device=dev) * 0.1
pool = torch.empty(4 * n + (1 << 26), device=dev)
print('pool base %#x' % pool.data_ptr())
x = pool[:n].view(B, H, H, C)
x.normal_()


def t(fn, iters=12):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for gap_bytes in (0, 256, 1024, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 32 << 20, (1 << 30) - 4 * n, (1 << 30) - 4 * n + 8192,
                  (1 << 29) - 4 * n if (1 << 29) > 4 * n else 12345 * 256):
    if gap_bytes < 0:
        continue
    off = n + gap_bytes // 4
    if off + n > pool.numel():
        continue
    y = pool[off:off + n].view(B, H, H, K)
    ms = t(lambda: ops.conv2d_fwd(x, wp, bias, K, 3, 3, 1, 1, 0.2, math.sqrt(2.0), out=y))
    print('y - x = 403 MB + %10d B  (y %% 1 GB = %#x)   fwd %.3f ms  %.1f TF/s' % (gap_bytes, y.data_ptr() % (1 << 30), ms,
                                                                              flops / ms / 1e9), flush=True)
# the in-step pattern of the backward: dgrad reading gy, act_ref and writing dx at the same three distances
gy = pool[:n].view(B, H, H, K)
for gap_bytes in (0, 4096, 2 << 20):
    off = n + gap_bytes // 4
    dx = pool[off:off + n].view(B, H, H, C)
    ms = t(lambda: ops.conv2d_dgrad(gy, wp, (B, H, H, C), 3, 3, 1, 1, out=dx))
    print('dx - gy = 403 MB + %10d B   dgrad %.3f ms  %.1f TF/s' % (gap_bytes, ms, flops / ms / 1e9), flush=True)
