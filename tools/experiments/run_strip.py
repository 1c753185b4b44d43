"""dev (NOT collected by pytest; needs conv_strip.hip linked into the library -- the shipped library does not export
contrad_conv3x3_c32_strip): the 32 -> 32 strip kernel against the implicit-GEMM engine (parity, then time at the 512 x 512 shape)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from contrad_amd import ops
from contrad_amd.ops import _p, _stream, lib, make_desc

dev = torch.device('cuda')


def strip(mode, d, inp, wp, bias, act, out, slope, gain):
    lib().call('contrad_conv3x3_c32_strip', ctypes.byref(d), mode, _p(inp), _p(wp), _p(bias), _p(act), _p(out), slope, gain,
               _stream())


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def main():
    g = torch.Generator().manual_seed(0)
    for (N, H, W) in ((2, 64, 64), (3, 128, 96)):
        x = torch.randn(N, H, W, 32, generator=g).to(dev)
        wp = ops.pack_weight(torch.randn(32, 32, 3, 3, generator=g) * 0.1).to(dev)
        bias = torch.randn(32, generator=g).to(dev)
        gy = torch.randn(N, H, W, 32, generator=g).to(dev)
        d = make_desc(N, H, W, 32, 32, 3, 3, 1, 1, 32, 32, wp.stride(0))
        assert lib().raw('contrad_conv3x3_c32_strip_ok')(ctypes.byref(d)) == 1
        os.environ['CONTRAD_CONV_STRIP'] = '0'
        y_ref = ops.conv2d_fwd(x, wp, bias, 32, 3, 3, 1, 1, 0.2, 1.4)
        dx_ref = ops.conv2d_dgrad(gy, wp, (N, H, W, 32), 3, 3, 1, 1, act_ref=x, slope=0.2, gain=1.4)
        dx_ref2 = ops.conv2d_dgrad(gy, wp, (N, H, W, 32), 3, 3, 1, 1)
        y = torch.empty_like(y_ref); dx = torch.empty_like(dx_ref); dx2 = torch.empty_like(dx_ref)
        strip(0, d, x, wp, bias, None, y, 0.2, 1.4)
        strip(1, d, gy, wp, None, x, dx, 0.2, 1.4)
        strip(1, d, gy, wp, None, None, dx2, 1.0, 1.0)
        print((N, H, W), 'fwd', rel(y, y_ref), 'dgrad+act', rel(dx, dx_ref), 'dgrad', rel(dx2, dx_ref2))

    N, H, W = 48, 512, 512
    x = torch.randn(N, H, W, 32, device=dev); wp = torch.randn(288, 32, device=dev) * 0.05; bias = torch.zeros(32, device=dev)
    y = torch.empty_like(x)
    d = make_desc(N, H, W, 32, 32, 3, 3, 1, 1, 32, 32, 32)
    flops = 2.0 * N * H * W * 32 * 32 * 9
    for name, fn in (('strip fwd', lambda: strip(0, d, x, wp, bias, None, y, 0.2, 1.4)),
                     ('strip dgrad', lambda: strip(1, d, x, wp, None, x, y, 0.2, 1.4)),
                     ('lean fwd', lambda: ops.conv2d_fwd(x, wp, bias, 32, 3, 3, 1, 1, 0.2, 1.4, out=y)),
                     ('lean dgrad', lambda: ops.conv2d_dgrad(x, wp, (N, H, W, 32), 3, 3, 1, 1, act_ref=x, slope=0.2, gain=1.4, out=y))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print('%-12s %.3f ms  %.1f TF/s' % (name, t, flops / t / 1e9))


if __name__ == '__main__':
    main()
