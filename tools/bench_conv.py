"""Per-layer timing of the conv engine at the BASELINE shape (SNDCGAN, 3N = 1536 images).  Dev tool."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from contrad_amd import ops

LAYERS = [(32, 64, 128, 4, 2, 1), (16, 128, 128, 3, 1, 1), (16, 128, 256, 4, 2, 1), (8, 256, 256, 3, 1, 1),
          (8, 256, 512, 4, 2, 1), (4, 512, 512, 3, 1, 1), (1, 8192, 1536, 1, 1, 0)]


# StyleGAN2 discriminator at 512x512 (3N = 48 images): conv1 of each ResBlock + the blurred stride-2 conv2
LAYERS_SG2 = [(512, 32, 32, 3, 1, 1), (513, 32, 64, 3, 2, 0), (256, 64, 64, 3, 1, 1), (257, 64, 128, 3, 2, 0),
              (128, 128, 128, 3, 1, 1), (64, 256, 256, 3, 1, 1), (32, 512, 512, 3, 1, 1)]
if os.environ.get('CONV_SET') == 'sg2':
    LAYERS = LAYERS_SG2
if os.environ.get('CONV_SET') == 'sg2full':       # every conv of ResidualDiscriminatorP(512, channel_multiplier=1)
    ch = {512: 32, 256: 64, 128: 128, 64: 256, 32: 512, 16: 512, 8: 512, 4: 512}
    LAYERS = []
    for R in (512, 256, 128, 64, 32, 16, 8):
        LAYERS += [(R, ch[R], ch[R], 3, 1, 1), (R + 1, ch[R], ch[R // 2], 3, 2, 0), (R // 2, ch[R], ch[R // 2], 1, 1, 0)]

if os.environ.get('CONV_CUSTOM'):     # "H,C,K,k,s,p;H,C,K,k,s,p;..."
    LAYERS = [tuple(int(v) for v in item.split(',')) for item in os.environ['CONV_CUSTOM'].split(';')]

ITERS = int(os.environ.get('CONV_ITERS', '10'))
WARM = int(os.environ.get('CONV_WARM', '3'))


def timeit(fn, iters=None):
    iters = iters or ITERS
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main(B=int(os.environ.get('CONV_BATCH', '1536'))):
    dev = torch.device('cuda')
    tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
    totf = 0.0
    only = os.environ.get('CONV_LAYERS')
    layers = LAYERS if not only else [LAYERS[int(i)] for i in only.split(',')]
    for (H, C, K, k, s, p) in layers:
        pad = int(os.environ.get('CONV_LDPAD', '0'))     # dev: channel pitch = channels + pad (an image is then not 2^k bytes)
        x = torch.randn(B, H, H, C + pad, device=dev)[..., :C]
        wp = torch.randn(k * k * C, K, device=dev) * 0.05
        Ho = ops.out_size(H, k, s, p)
        gy = torch.randn(B, Ho, Ho, K + pad, device=dev)[..., :K]
        y = torch.empty(B, Ho, Ho, K + pad, device=dev)[..., :K]
        dx = torch.empty(B, H, H, C + pad, device=dev)[..., :C]
        dw = torch.empty_like(wp)
        flops = 2.0 * B * Ho * Ho * K * C * k * k
        modes = os.environ.get('CONV_MODES', 'fwd,dgrad,wgrad').split(',')      # dev: time a subset (the others print inf)
        t_f = timeit(lambda: ops.conv2d_fwd(x, wp, None, K, k, k, s, p, 0.1, 1.0, out=y)) if 'fwd' in modes else float('inf')
        t_d = timeit(lambda: ops.conv2d_dgrad(gy, wp, tuple(x.shape), k, k, s, p, out=dx)) if 'dgrad' in modes else float('inf')
        t_w = timeit(lambda: ops.conv2d_wgrad(x, gy, k, k, s, p, out=dw)) if 'wgrad' in modes else float('inf')
        print('H%-3d C%-4d K%-4d k%d s%d  %6.1f GF | fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF'
              % (H, C, K, k, s, flops / 1e9, t_f, flops / t_f / 1e9, t_d, flops / t_d / 1e9, t_w, flops / t_w / 1e9),
              flush=True)
        tot['fwd'] += t_f; tot['dgrad'] += t_d; tot['wgrad'] += t_w; totf += flops
    for k_, v in tot.items():
        print('%-6s total %7.3f ms  %6.1f TF/s' % (k_, v, totf / v / 1e9))


if __name__ == '__main__':
    main()
