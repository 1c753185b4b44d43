"""GPU parity: implicit-GEMM conv engine (fwd / dgrad / wgrad) through the C ABI vs PyTorch-CPU fp32
(the reference's arithmetic for F.conv2d / F.linear / ConvTranspose2d lives in PyTorch).
Tolerance: 1e-3 relative to the tensor's max magnitude (north_star), typically ~1e-6 observed."""
import os

import pytest
import torch
import torch.nn.functional as F

from contrad_amd import ops

pytestmark = pytest.mark.gpu
TOL = 1e-3

# (N, H, W, C, K, k, stride, pad)
CASES = [
    (6, 32, 32, 64, 128, 4, 2, 1),     # SNDCGAN main.2
    (6, 16, 16, 128, 128, 3, 1, 1),    # main.4
    (6, 16, 16, 128, 256, 4, 2, 1),    # main.6
    (6, 8, 8, 256, 256, 3, 1, 1),      # main.8
    (6, 8, 8, 256, 512, 4, 2, 1),      # main.10
    (6, 4, 4, 512, 512, 3, 1, 1),      # main.12
    (5, 33, 33, 32, 64, 3, 2, 0),      # StyleGAN2 blurred 3x3 stride-2 (odd input)
    (5, 17, 17, 64, 128, 1, 2, 0),     # StyleGAN2 skip: blurred 1x1 stride-2
    (3, 9, 7, 8, 12, 3, 1, 1),         # ragged everything, K % 4 == 0
    (2, 5, 5, 3, 64, 3, 1, 1),         # Cin = 3 (scalar gather fallback)
    (7, 1, 1, 512, 1, 1, 1, 0),        # linear 512 -> 1 (Cout = 1)
    (70, 1, 1, 8192, 384, 1, 1, 0),    # wide linear (heads)
    (3, 6, 6, 516, 32, 3, 1, 1),       # Cin = 513 padded to 516 (last_conv)
    # lean-loop geometry (igemm_lean.h): K-tiles inside one tap, WGRAD 16-position patches
    (4, 64, 64, 32, 64, 3, 1, 1),      # wide rows: patch = 16 consecutive wo, Wo = 64
    (2, 64, 64, 16, 32, 3, 2, 1),      # Cin = 16 (one K-tile per tap), stride 2, Ho = 32
    (32, 1, 1, 256, 128, 1, 1, 0),     # linear: patch = 16 images
    (16, 2, 2, 64, 64, 3, 1, 1),       # 2x2 maps: patch = 4 images, every tap but the centre is padding somewhere
    (8, 4, 4, 64, 96, 4, 2, 1),        # Cout = 96: ragged column tile
    (3, 16, 16, 48, 80, 3, 1, 1),      # Kg = 432: ragged WGRAD row tile
    (2, 20, 12, 32, 48, 3, 1, 1),      # Wo = 12: lean FWD / DGRAD with a ragged M tile, general WGRAD
    (70, 8, 8, 64, 64, 3, 1, 1),       # M-tiles that start in the middle of an image (descriptor rebasing)
    (6, 16, 16, 72, 136, 3, 1, 1),     # Cin % 16 != 0: the general float4 kernel at the big tile
    (2, 32, 32, 32, 32, 3, 1, 1),      # 32 -> 32 channels (StyleGAN2 at 512x512): the 128x32 tile in all three modes
    (3, 16, 16, 64, 16, 3, 1, 1),      # Cout = 16: half-empty 32-column tile; DGRAD contraction of one K-tile per tap
    (2, 16, 16, 16, 32, 3, 2, 1),      # Cin = 16 -> DGRAD writes a 16-column tile
    (192, 1, 1, 8192, 1536, 1, 1, 0),  # the merged head GEMM at a per-rank batch of 64: split-K forward
    (12, 4, 4, 512, 512, 3, 1, 1),     # deep 3x3 layer at a small batch: split-K forward starting mid-chunk
    (4, 128, 128, 32, 32, 3, 1, 1),    # 32 -> 32 at >= 64 k positions: weight-stationary FWD / DGRAD (conv_c32.h), accumulator-stationary WGRAD (wgrad_c32.h)
    (130, 4, 4, 32, 64, 3, 1, 1),      # pixel-major tiles (>= one tile of images on a small map): ragged last image block
    (130, 8, 8, 16, 64, 4, 2, 1),      # ... with the 4x4 stride-2 kernel (4x4 output map); its data gradient: pixel-major inside the parity classes
    (130, 5, 7, 16, 64, 3, 2, 1),      # ... odd map, 3x3 stride 2: parity classes of different sizes (3x4, 3x3, 2x4, 2x3 pixels)
    (144, 4, 4, 128, 64, 3, 1, 1),     # ... weight gradient on pixel-major positions (K-tiles = 16 images at one pixel, padding ones skipped)
    (144, 8, 8, 128, 96, 4, 2, 1),     # ... the same with the 4x4 stride-2 kernel, ragged column tile
    (1040, 8, 8, 64, 64, 3, 1, 1),     # border classes (FWD + DGRAD): 0.84 of the tap-positions valid -> (image, pixel) tiles inside the 9 border rectangles, an eighth of each class per XCD
    (1040, 12, 20, 16, 64, 4, 2, 1),   # ... 4x4 stride-2 onto a 6 x 10 map (FWD), ragged class tiles
    (130, 8, 8, 32, 64, 3, 1, 1),      # (too few images for the automatic plan: border classes only when forced, see the subprocess test)
    (70, 12, 20, 16, 64, 4, 2, 1),
    (200, 2, 6, 32, 48, 3, 1, 1),      # ... a 2 x 6 map: most taps of most pixels are padding
    (3, 64, 512, 32, 32, 3, 1, 1),     # ... a non-square map: 3 images x 16 x 16 tiles of 4 x 32, more tiles than one per block
    # balanced block order of the strided data gradient (3x3 stride 2: blocks of the 4 / 2 / 2 / 1-tap parity classes walk
    # 1 / 2 / 2 / 4 M-tiles): several groups of 8 M-tile indices, ragged last group, classes of different sizes
    (7, 17, 17, 64, 128, 3, 2, 0),     # 567 / 504 / 504 / 448 rows per class: 9 (8, 8, 7) tiles of 64 -> two groups, the second almost empty
    (40, 9, 9, 32, 48, 3, 2, 0),       # 1000 / 800 / 800 / 640 rows, Cout = 48: ragged K walk (3 K-tiles per tap)
    (3, 33, 33, 128, 144, 3, 2, 0),    # two N-tiles (128 + ragged 16), 128-row tiles
    (20, 11, 15, 16, 32, 3, 2, 1),     # pad 1: the class order is 1 / 2 / 2 / 4 taps, non-square odd map
    (2, 129, 129, 16, 16, 3, 2, 0),    # a long launch: 66 M-tiles per class at BM = 128 -> nine groups
    (16, 65, 65, 64, 64, 3, 2, 0),     # > 1024 blocks in the plain order: the plan itself takes the balanced order here
    # pixel-major tiles in the slot-balanced order of the whole launch (pixel_order_full): FWD / stride-1 DGRAD on a 4x4 map,
    # and the strided DGRAD of the 4x4 stride-2 layer (class-major launch, one table mirrored into the four parity classes)
    (384, 4, 4, 64, 256, 3, 1, 1),     # 3 image blocks x 16 pixels, 2 / 4 N-tiles
    (260, 8, 8, 32, 64, 4, 2, 1),      # strided: ragged last image block, one N-tile
    (256, 8, 8, 256, 64, 4, 2, 1),     # strided: several N-tiles per M-tile
]


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def lrelu_ref(y, slope, gain):
    return F.leaky_relu(y, slope) * gain


@pytest.mark.parametrize('case', CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_fwd_dgrad_wgrad(case):
    N, H, W, C, K, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, k, k, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    slope, gain = 0.2, 2 ** 0.5

    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    y_lin = F.conv2d(xr, wr, b, stride=s, padding=p)
    y_ref = lrelu_ref(y_lin, slope, gain)
    gy = torch.randn(y_ref.shape, generator=g)
    # gradient wrt the pre-activation conv output: feed dgrad/wgrad the same gy on the linear conv
    gx_ref, gw_ref = torch.autograd.grad(y_lin, (xr, wr), gy)

    dev = torch.device('cuda')
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = ops.pack_weight(w).to(dev)
    y = ops.conv2d_fwd(x_nhwc, wp, b.to(dev), K, k, k, s, p, slope, gain)
    assert rel(y.permute(0, 3, 1, 2).cpu(), y_ref.detach()) < TOL
    # residual addend in the epilogue (contrad_conv2d_fwd_add): y = gain * lrelu(conv + bias) + addend, on whichever
    # path the shape takes (lean loop, general float4 / scalar kernel, split-K reduce); the plain result stays bitwise
    addend = torch.randn(y_ref.shape, generator=g)
    ya = ops.conv2d_fwd(x_nhwc, wp, b.to(dev), K, k, k, s, p, slope, gain,
                        addend=addend.permute(0, 2, 3, 1).contiguous().to(dev))
    assert rel(ya.permute(0, 3, 1, 2).cpu(), y_ref.detach() + addend) < TOL
    assert torch.equal((ya - addend.permute(0, 2, 3, 1).to(dev)).sub(y).abs().max() < 1e-5 * y.abs().max(),
                       torch.tensor(True, device=dev))

    gy_nhwc = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    dx = ops.conv2d_dgrad(gy_nhwc, wp, (N, H, W, C), k, k, s, p)
    assert rel(dx.permute(0, 3, 1, 2).cpu(), gx_ref) < TOL
    if os.environ.get('CONTRAD_TEST_EXPECT_DGRAD_PATH'):      # (set by the subprocess test below)
        import ctypes
        from contrad_amd import _lib
        d = ops.make_desc(N, H, W, C, K, k, k, s, p, C, K, wp.stride(0))
        assert _lib.lib().raw('contrad_conv2d_path')(ctypes.byref(d), 1) == int(os.environ['CONTRAD_TEST_EXPECT_DGRAD_PATH'])

    fused_bias = (C % 4 == 0 and K % 4 == 0)
    db = torch.zeros(K, device=dev) if fused_bias else None
    dwp = ops.conv2d_wgrad(x_nhwc, gy_nhwc, k, k, s, p, dbias=db)
    dw = ops.unpack_weight(dwp.cpu(), K, C, k, k)
    assert rel(dw, gw_ref) < TOL
    if fused_bias:      # bias gradient accumulated from the gy tiles the wgrad kernel streams anyway
        assert rel(db.cpu(), gy.sum((0, 2, 3))) < TOL


def _dev_lib():
    """The build with the development switches compiled in (contrad_amd/build.py): the shipped library reads no environment."""
    from contrad_amd import build
    assert os.path.exists(build.DEV_LIB), 'run __graft_entry__.build() first'
    return build.DEV_LIB


def test_strided_pixel_major_dgrad_in_a_fresh_process():
    """The launch plan keeps strided data gradients off the pixel-major tiles (slower on the 4x4 stride-2 layer), but the
    kernel walks them inside the parity classes all the same: forced on (the switch is read once per process), the
    4x4 stride-2 and the odd-map 3x3 stride-2 cases must still match the fp32 reference."""
    import subprocess
    import sys
    env = dict(os.environ, CONTRAD_PIXMAJOR_STRIDED='1', CONTRAD_TEST_EXPECT_DGRAD_PATH='3', CONTRAD_HIP_LIB=_dev_lib())
    env.pop('CONTRAD_TILEMODE', None)          # (a forced tile mode would override the plan under test)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-x', '-k',
                        '130x8x8x16x64x4x2x1'], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '1 passed' in r.stdout


def test_forced_balanced_strided_dgrad_in_a_fresh_process():
    """The balanced block order of the 3x3 stride-2 data gradient (blocks of the light parity classes walk 2 / 4 M-tiles) is
    taken by the plan only for launches of more than 1024 blocks; forced on (dev build), the small ragged cases -- several
    groups, an almost empty last group, classes of different sizes, pad 0 and pad 1 -- must match the fp32 reference too."""
    import subprocess
    import sys
    env = dict(os.environ, CONTRAD_DGRAD_BALANCE='2', CONTRAD_HIP_LIB=_dev_lib())
    env.pop('CONTRAD_TILEMODE', None); env.pop('CONTRAD_TEST_EXPECT_DGRAD_PATH', None)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-x', '-k',
                        '5x33x33x32 or 7x17x17x64 or 40x9x9x32 or 3x33x33x128 or 20x11x15x16 or 2x129x129x16 or 2x64x64x16x32x3x2'],
                       env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '7 passed' in r.stdout


def test_forced_border_classes_in_a_fresh_process():
    """Border classes with class tiles that are ragged or fewer than the 8 XCDs (small batches: the plan keeps them off
    there): forced on for FWD and stride-1 DGRAD, the small cases must still match the fp32 reference."""
    import subprocess
    import sys
    env = dict(os.environ, CONTRAD_TILEMODE='2', CONTRAD_HIP_LIB=_dev_lib())
    env.pop('CONTRAD_TEST_EXPECT_DGRAD_PATH', None)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-m', 'gpu', '-x', '-k',
                        '130x8x8x32 or 70x12x20 or 200x2x6 or 130x4x4 or 130x5x7 or 16x2x2'], env=env, capture_output=True,
                       text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert '6 passed' in r.stdout


@pytest.mark.parametrize('shape', [(4, 8, 8, 64, 64), (4, 128, 128, 32, 32)], ids=['lean', 'conv_c32'])
def test_dgrad_fused_activation_derivative(shape):
    """dgrad epilogue multiplies by lrelu'(act_ref) * gain -- the backward of the producer's activation (on the engine's
    lean loop and on the weight-stationary kernel of the 32 -> 32 channel layers, conv_c32.h)."""
    N, H, W, C, K = shape
    k, s, p = 3, 1, 1
    g = torch.Generator().manual_seed(3)
    a = torch.randn(N, C, H, W, generator=g)             # pre-activation of the producer layer
    w = torch.randn(K, C, k, k, generator=g) * 0.1
    ar = a.clone().requires_grad_()
    x = F.leaky_relu(ar, 0.1)
    y = F.conv2d(x, w, None, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    ga_ref, = torch.autograd.grad(y, ar, gy)
    dev = torch.device('cuda')
    x_nhwc = x.detach().permute(0, 2, 3, 1).contiguous().to(dev)
    da = ops.conv2d_dgrad(gy.permute(0, 2, 3, 1).contiguous().to(dev), ops.pack_weight(w).to(dev),
                          (N, H, W, C), k, k, s, p, act_ref=x_nhwc, slope=0.1, gain=1.0)
    assert rel(da.permute(0, 3, 1, 2).cpu(), ga_ref) < TOL


def test_conv_transpose_is_dgrad():
    """nn.ConvTranspose2d forward (G_SNDCGAN, sndcgan.py:26-38) == dgrad of the matching conv."""
    N, Cin, Cout, k, s, p, H = 4, 64, 32, 4, 2, 1, 8
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) * 0.1     # ConvTranspose2d weight layout
    ref = F.conv_transpose2d(x, w, None, stride=s, padding=p)
    dev = torch.device('cuda')
    # as a conv weight this is (K=Cin, C=Cout, k, k); the transposed conv's output is that conv's input
    wp = ops.pack_weight(w).to(dev)
    out = ops.conv2d_dgrad(x.permute(0, 2, 3, 1).contiguous().to(dev), wp, (N, 2 * H, 2 * H, Cout), k, k, s, p)
    assert rel(out.permute(0, 3, 1, 2).cpu(), ref) < TOL


def test_channel_sliced_views():
    """Leading dimensions: read a channel slice of a wider NHWC buffer, write into a slice of another."""
    N, H, W = 3, 4, 4
    g = torch.Generator().manual_seed(5)
    big = torch.randn(N, H, W, 96, generator=g)
    w = torch.randn(48, 32, 1, 1, generator=g)
    dev = torch.device('cuda')
    bigd = big.to(dev)
    outbuf = torch.zeros(N, H, W, 80, device=dev)
    ops.conv2d_fwd(bigd[..., 32:64], ops.pack_weight(w).to(dev), None, 48, 1, 1, 1, 0, out=outbuf[..., 16:64])
    ref = F.conv2d(big[..., 32:64].permute(0, 3, 1, 2), w).permute(0, 2, 3, 1)
    assert rel(outbuf[..., 16:64].cpu(), ref) < TOL
    assert outbuf[..., :16].abs().max().item() == 0 and outbuf[..., 64:].abs().max().item() == 0


def test_channel_sliced_views_dgrad_wgrad():
    """Leading dimensions on the backward kernels: gy read from a channel slice, dx written into a slice of a wider
    buffer (its neighbours untouched), act_ref with the same leading dimension, x read from a slice for wgrad."""
    N, H, W, C, K = 3, 8, 8, 32, 48
    g = torch.Generator().manual_seed(6)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    gy_big = torch.randn(N, H, W, 80, generator=g)
    x_big = torch.randn(N, H, W, 96, generator=g)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    gy_s = gy_big.to(dev)[..., 16:64]                 # (N,H,W,48) view, ld 80
    x_s = x_big.to(dev)[..., 32:64]                   # (N,H,W,32) view, ld 96
    dxbuf = torch.full((N, H, W, 96), 7.0, device=dev)
    ops.conv2d_dgrad(gy_s, wp, (N, H, W, C), 3, 3, 1, 1, act_ref=x_s, slope=0.2, gain=1.5, out=dxbuf[..., 32:64])
    xr = x_big[..., 32:64].permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(gy_big[..., 16:64].permute(0, 3, 1, 2), w, None, stride=1, padding=1)
    ref = ref * torch.where(xr > 0, torch.tensor(1.5), torch.tensor(1.5 * 0.2))
    assert rel(dxbuf[..., 32:64].permute(0, 3, 1, 2).cpu(), ref) < TOL
    assert torch.all(dxbuf[..., :32] == 7.0) and torch.all(dxbuf[..., 64:] == 7.0)
    db = torch.zeros(K, device=dev)
    dwp = ops.conv2d_wgrad(x_s, gy_s, 3, 3, 1, 1, dbias=db)
    xg = x_big[..., 32:64].permute(0, 3, 1, 2).clone().requires_grad_()
    wr = w.clone().requires_grad_()
    y = F.conv2d(xg, wr, None, stride=1, padding=1)
    gw_ref, = torch.autograd.grad(y, wr, gy_big[..., 16:64].permute(0, 3, 1, 2))
    assert rel(ops.unpack_weight(dwp.cpu(), K, C, 3, 3), gw_ref) < TOL
    assert rel(db.cpu(), gy_big[..., 16:64].sum((0, 1, 2))) < TOL


def test_wgrad_is_deterministic_and_linear():
    N, H, W, C, K = 64, 16, 16, 128, 128
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(N, H, W, C, device=dev, generator=g)
    gy = torch.randn(N, H, W, K, device=dev, generator=g)
    a = ops.conv2d_wgrad(x, gy, 3, 3, 1, 1).clone()
    b = ops.conv2d_wgrad(x, gy, 3, 3, 1, 1).clone()
    assert torch.equal(a, b)                                   # fixed-order split reduction
    c = ops.conv2d_wgrad(x, gy * 2.0, 3, 3, 1, 1)
    assert torch.equal(c, a * 2.0)                             # exact power-of-two linearity


def test_full_size_sndcgan_layer_linearity():
    """BASELINE size (3N = 1536 images, main.4 128->128 3x3 at 16x16): size-independent properties."""
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(1536, 16, 16, 128, device=dev, generator=g)
    w = torch.randn(128, 128, 3, 3, device=dev, generator=g) * 0.05
    wp = ops.pack_weight(w)
    y = ops.conv2d_fwd(x, wp, None, 128, 3, 3, 1, 1)
    # batch independence: the first 8 images alone give the same rows (a small batch may be planned with another
    # tile / split-K, i.e. another fp32 summation order, so this is not bitwise), and a repeated call IS bitwise equal
    y8 = ops.conv2d_fwd(x[:8].contiguous(), wp, None, 128, 3, 3, 1, 1)
    # (the full batch runs on Winograd F(4x4, 3x3), the eight images on the direct kernel: 1.3e-5 apart -- round-off of the
    # 6x6 transforms, csrc/wino44.h; the contract is 1e-3)
    assert rel(y[:8], y8) < 1e-4
    assert torch.equal(y, ops.conv2d_fwd(x, wp, None, 128, 3, 3, 1, 1))
    # adjoint identity <conv(x), gy> == <x, dgrad(gy)>
    gy = torch.randn_like(y)
    dx = ops.conv2d_dgrad(gy, wp, tuple(x.shape), 3, 3, 1, 1)
    lhs = (y.double() * gy.double()).sum()
    rhs = (x.double() * dx.double()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 1e-4
    # and == <w, wgrad(x, gy)>
    dwp = ops.conv2d_wgrad(x, gy, 3, 3, 1, 1)
    rhs2 = (wp.double() * dwp.double()).sum()
    assert abs(lhs - rhs2) / abs(lhs) < 1e-4
    # spot-check 4 images against PyTorch CPU
    ref = F.conv2d(x[:4].cpu().permute(0, 3, 1, 2), w.cpu(), padding=1).permute(0, 2, 3, 1)
    assert rel(y[:4].cpu(), ref) < TOL


def test_dgrad_split_k_equals_the_unsplit_kernel_with_fused_epilogue_and_sliced_dx():
    """Stride-1 DGRAD of a small-M / deep-K layer splits the contraction into partial slabs (contrad_conv2d_dgrad_ws);
    the plain entry point never does.  Same sums up to fp32 reassociation, incl. the fused act' epilogue applied by the
    reduce kernel and a dx that is a channel slice of a wider buffer (ldx > C)."""
    import ctypes
    from contrad_amd.ops import _p, _stream, lib, make_desc
    N, H, W, C, K, k, s, p = 12, 4, 4, 512, 512, 3, 1, 1
    g = torch.Generator().manual_seed(11)
    dev = torch.device('cuda')
    gy = torch.randn(N, H, W, K, generator=g).to(dev)
    wp = ops.pack_weight(torch.randn(K, C, k, k, generator=g) * 0.05).to(dev)
    act = torch.randn(N, H, W, C + 64, generator=g).to(dev)
    for sliced in (False, True):
        ldx = C + 64 if sliced else C
        d = make_desc(N, H, W, C, K, k, k, s, p, ldx, K, wp.stride(0))
        nbytes = lib().raw('contrad_conv2d_dgrad_workspace_bytes')(ctypes.byref(d))
        assert nbytes > 0 and nbytes % (N * H * W * ldx * 4) == 0 and nbytes // (N * H * W * ldx * 4) >= 2
        ref_buf = torch.zeros(N, H, W, ldx, device=dev)
        a_ref = act if sliced else act[..., :C].contiguous()
        lib().call('contrad_conv2d_dgrad', ctypes.byref(d), _p(gy), _p(wp), _p(ref_buf), _p(a_ref), 0.2, 1.5, _stream())
        out_buf = torch.zeros(N, H, W, ldx, device=dev)
        dx = ops.conv2d_dgrad(gy, wp, (N, H, W, C), k, k, s, p, act_ref=a_ref[..., :C], slope=0.2, gain=1.5,
                              out=out_buf[..., :C])
        assert rel(dx.cpu(), ref_buf[..., :C].cpu()) < 1e-5
        assert not sliced or out_buf[..., C:].abs().max().item() == 0       # nothing written past the slice
        again = ops.conv2d_dgrad(gy, wp, (N, H, W, C), k, k, s, p, act_ref=a_ref[..., :C], slope=0.2, gain=1.5,
                                 out=torch.zeros(N, H, W, ldx, device=dev)[..., :C])
        assert torch.equal(again, dx)                                        # fixed summation order


def test_nontemporal_store_path_equals_the_plain_one():
    """Outputs of >= 256 MB take non-temporal epilogue stores (igemm.hip NT_STORE_BYTES; conv_c32 kernels too): the same
    layer on a batch whose output is 268 MB and on its two halves (134 MB each: plain stores) -- equal results (1e-6: the
    plan may change with the batch), forward with bias / activation / addend, data gradient with the act' epilogue, for a
    lean-engine shape and for the 32-channel weight-stationary kernel; spot-checked against PyTorch CPU."""
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(4)
    for (N, H, C, K) in ((16, 256, 64, 64), (32, 256, 32, 32)):          # 16 * 256^2 * 64 * 4 B = 268 MB; conv_c32 shape
        x = torch.randn(N, H, H, C, device=dev, generator=g)
        w = torch.randn(K, C, 3, 3, device=dev, generator=g) * 0.05
        b = torch.randn(K, device=dev, generator=g)
        wp = ops.pack_weight(w)
        add = torch.randn(N, H, H, K, device=dev, generator=g)
        assert add.numel() * 4 >= 256 << 20 and add.numel() * 2 < 256 << 20
        y = ops.conv2d_fwd(x, wp, b, K, 3, 3, 1, 1, 0.2, 1.4, addend=add)
        h = N // 2
        for sl in (slice(0, h), slice(h, N)):
            yh = ops.conv2d_fwd(x[sl].contiguous(), wp, b, K, 3, 3, 1, 1, 0.2, 1.4, addend=add[sl].contiguous())
            assert rel(y[sl], yh) < 1e-6
        ref = F.leaky_relu(F.conv2d(x[:1].cpu().permute(0, 3, 1, 2), w.cpu(), b.cpu(), padding=1), 0.2) * 1.4
        assert rel(y[:1].cpu(), ref.permute(0, 2, 3, 1) + add[:1].cpu()) < TOL
        gy = torch.randn(N, H, H, K, device=dev, generator=g)
        act = torch.randn(N, H, H, C, device=dev, generator=g)
        dx = ops.conv2d_dgrad(gy, wp, tuple(x.shape), 3, 3, 1, 1, act_ref=act, slope=0.2, gain=1.4)
        for sl in (slice(0, h), slice(h, N)):
            dh = ops.conv2d_dgrad(gy[sl].contiguous(), wp, (sl.stop - sl.start, H, H, C), 3, 3, 1, 1,
                                  act_ref=act[sl].contiguous(), slope=0.2, gain=1.4)
            assert rel(dx[sl], dh) < 1e-6
        del x, add, y, gy, act, dx
