"""GPU parity: Winograd F(2x2, 3x3) / F(4x4, 3x3) kernels (csrc/wino.h, wino44.h) through the C ABI vs PyTorch-CPU fp32 F.conv2d -- the reference's
own arithmetic for these layers (models/gan/sndcgan.py:91-109, models/gan/stylegan2/layers.py:115-121).

``contrad_conv2d_wino`` forces the kernel on every shape it supports (the automatic plan of conv2d_fwd / conv2d_dgrad only
takes it for launches that fill the chip: checked at the BASELINE shapes below).  Tolerance 1e-3 relative to the tensor's max
magnitude (north_star); observed 2e-7 ... 2e-6."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from contrad_amd import ops
from contrad_amd._lib import lib

pytestmark = pytest.mark.gpu
TOL = 1e-3
TIGHT = 2e-5        # what the kernel actually delivers (fp32 round-off class): a regression guard far below the contract

# (N, H, W, C, K)
CASES = [
    (3, 16, 16, 32, 64),      # one 16 x 16 patch per image: the box is the image, the halo reads the zero pixel
    (2, 32, 32, 16, 128),     # 2 x 2 patches per image: boxes with halo, borders load as zeros; two cout blocks
    (5, 8, 8, 64, 64),        # 4 images per block, ragged last block
    (18, 4, 4, 32, 64),       # 16 images per block, ragged
    (2, 64, 32, 16, 64),      # non-square: 4 x 2 patches
    (1, 4, 8, 16, 64),        # 8 images per block, one present
    (3, 16, 32, 48, 192),     # one patch high, two wide (box 16 x 18); Cin = 48 (6 chunks), three cout blocks
    (40, 16, 16, 16, 64),     # several items per block of the persistent grid (one chunk pair each)
    (9, 8, 16, 128, 64),      # 2 images per block (4 x 8 tiles each)
]


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _inputs(N, H, W, C, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    return x, w, b


@pytest.mark.parametrize('case', CASES)
def test_wino_forward(case):
    N, H, W, C, K = case
    x, w, b = _inputs(N, H, W, C, K, 1)
    add = torch.randn(N, H, W, K, generator=torch.Generator().manual_seed(2))
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    ref_act = F.leaky_relu(ref, 0.2) * 1.3
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev), slope=0.2, gain=1.3)
    assert rel(y.cpu(), ref_act) < TIGHT
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=None, ref=add.to(dev), slope=1.0, gain=1.0)
    assert rel(y.cpu(), F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=1).permute(0, 2, 3, 1) + add) < TIGHT


@pytest.mark.parametrize('case', CASES)
def test_wino_data_gradient(case):
    N, H, W, K, C = case          # (roles swapped: gy has K channels, dx has C; the kernel needs K % 16 == 0, C % 64 == 0)
    g = torch.Generator().manual_seed(3)
    gy = torch.randn(N, H, W, K, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    act = torch.randn(N, H, W, C, generator=g)
    ref = F.conv_transpose2d(gy.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K)
    assert rel(dx.cpu(), ref) < TIGHT
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K, ref=act.to(dev), slope=0.2, gain=1.5)
    assert rel(dx.cpu(), ref * torch.where(act > 0, 1.5, 0.3)) < TIGHT


# 4x4 stride-2 pad-1 layers on F(2x2, 2x2) over the four phases (csrc/wino22.h): (N, H, W, C, K), H / 2 and W / 2 in {4, 8, 16}
S2CASES = [
    (3, 32, 32, 8, 64),        # 16 x 16 output grid: 2 images per block, ragged; one chunk per phase
    (9, 16, 16, 16, 128),      # 8 x 8 grid: 8 images per block, ragged; two cout blocks
    (40, 8, 8, 32, 64),        # 4 x 4 grid: 32 images per block, ragged
    (2, 32, 16, 24, 64),       # non-square 16 x 8 grid (4 images per block); Cin = 24: three chunks per phase
    (70, 8, 8, 16, 192),       # several items per block, three cout blocks
]


@pytest.mark.parametrize('case', S2CASES)
def test_wino22_forward(case):
    N, H, W, C, K = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, H, W, C, generator=g)
    w = torch.randn(K, C, 4, 4, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    add = torch.randn(N, H // 2, W // 2, K, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=2, padding=1).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev), slope=0.1, gain=1.2, k4s2=True)
    assert rel(y.cpu(), F.leaky_relu(ref, 0.1) * 1.2) < TIGHT
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, ref=add.to(dev), k4s2=True)
    assert rel(y.cpu(), F.conv2d(x.permute(0, 3, 1, 2), w, None, stride=2, padding=1).permute(0, 2, 3, 1) + add) < TIGHT


@pytest.mark.parametrize('case', S2CASES)
def test_wino22_data_gradient(case):
    N, H, W, K, C = case           # (roles swapped: gy has K channels -- a multiple of 16 --, dx has C -- a multiple of 64)
    if K % 16:
        K = 16
    g = torch.Generator().manual_seed(22)
    gy = torch.randn(N, H // 2, W // 2, K, generator=g)
    w = torch.randn(K, C, 4, 4, generator=g) * 0.1
    act = torch.randn(N, H, W, C, generator=g)
    ref = F.conv_transpose2d(gy.permute(0, 3, 1, 2), w, stride=2, padding=1).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K, k4s2=True)
    assert rel(dx.cpu(), ref) < TIGHT
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K, ref=act.to(dev), slope=0.1, gain=1.0, k4s2=True)
    assert rel(dx.cpu(), ref * torch.where(act > 0, 1.0, 0.1)) < TIGHT


# weight gradient of the 4x4 stride-2 layers: (N, H, W, C, K), C = 64 or a multiple of 128, K a multiple of 64
S2WCASES = [
    (3, 32, 32, 64, 64),       # 16 x 16 grid: chunks of 2 x 4 tiles, boxes 5 x 9; C = 64: two phases per row block
    (5, 16, 16, 128, 128),     # 8 x 8 grid (full-width chunks); C = 128: one phase per row block, two k-blocks
    (19, 8, 8, 256, 64),       # 4 x 4 grid: 2 images per chunk, ragged last chunk; two row blocks per phase
    (2, 32, 16, 64, 64),       # non-square 16 x 8 grid
    (130, 16, 16, 64, 64),     # many chunks, several splits with odd chunk counts
]


@pytest.mark.parametrize('case', S2WCASES)
def test_wino22_weight_gradient(case):
    N, H, W, C, K = case
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, H, W, C, generator=g)
    gy = torch.randn(N, H // 2, W // 2, K, generator=g)
    w0 = torch.zeros(K, C, 4, 4, requires_grad=True)
    b0 = torch.zeros(K, requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2), w0, b0, stride=2, padding=1).backward(gy.permute(0, 3, 1, 2))
    dev = torch.device('cuda')
    dbias = torch.empty(K, device=dev)
    dwp = ops.conv2d_wino_wgrad(x.to(dev), gy.to(dev), dbias=dbias)
    assert rel(ops.unpack_weight(dwp, K, C, 4, 4).cpu(), w0.grad) < TIGHT
    assert rel(dbias.cpu(), b0.grad) < TIGHT
    assert torch.equal(dwp, ops.conv2d_wino_wgrad(x.to(dev), gy.to(dev)))


# (N, H, W, C, K): C and K multiples of 64
WCASES = [
    (3, 16, 16, 64, 64),       # chunks of 2 x 4 tiles: 4 x 2 chunks per image, boxes of 6 x 10 pixels with halo
    (2, 32, 32, 64, 128),      # boxes with halo on both axes, two k-blocks
    (5, 8, 8, 128, 64),        # full-width chunks (box 6 x 8), two c-blocks; odd chunk count per split possible
    (19, 4, 4, 64, 64),        # 2 images per chunk, ragged last chunk
    (2, 64, 16, 64, 64),       # non-square
    (1, 4, 8, 64, 64),         # one chunk per image
    (40, 16, 16, 64, 192),     # many chunks, several splits
]


@pytest.mark.parametrize('case', WCASES)
def test_wino_weight_gradient(case):
    N, H, W, C, K = case
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, H, W, C, generator=g)
    gy = torch.randn(N, H, W, K, generator=g)
    xr = x.permute(0, 3, 1, 2).contiguous().requires_grad_(False)
    w0 = torch.zeros(K, C, 3, 3, requires_grad=True)
    b0 = torch.zeros(K, requires_grad=True)
    F.conv2d(xr, w0, b0, padding=1).backward(gy.permute(0, 3, 1, 2))
    dev = torch.device('cuda')
    dbias = torch.empty(K, device=dev)
    dwp = ops.conv2d_wino_wgrad(x.to(dev), gy.to(dev), dbias=dbias)
    dw = ops.unpack_weight(dwp, K, C, 3, 3).cpu()
    assert rel(dw, w0.grad) < TIGHT
    assert rel(dbias.cpu(), b0.grad) < TIGHT
    dwp2 = ops.conv2d_wino_wgrad(x.to(dev), gy.to(dev))
    assert torch.equal(dwp, dwp2)            # deterministic (fixed-order reduce), with or without the bias gradient


def test_wino_channel_sliced_views():
    """Input and output are channel slices of wider buffers (leading dimension != channels), as the engine's callers pass them."""
    N, H, W, C, K = 3, 16, 16, 32, 64
    x, w, b = _inputs(N, H, W, C, K, 5)
    dev = torch.device('cuda')
    xb = torch.full((N, H, W, C + 8), 7.0, device=dev)
    xb[..., 4:4 + C] = x.to(dev)
    yb = torch.full((N, H, W, K + 12), -3.0, device=dev)
    ops.conv2d_wino(0, xb[..., 4:4 + C], ops.pack_weight(w).to(dev), C, K, bias=b.to(dev), slope=0.1, gain=1.0, out=yb[..., 8:8 + K])
    ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1), 0.1).permute(0, 2, 3, 1)
    assert rel(yb[..., 8:8 + K].cpu(), ref) < TIGHT
    assert (yb[..., :8] == -3.0).all() and (yb[..., 8 + K:] == -3.0).all()      # nothing outside the slice is written


def test_strided_and_gradient_kernels_on_channel_sliced_views():
    """The same for the 4x4 stride-2 kernel (forward and data gradient) and for both weight-gradient kernels: every operand a
    channel slice of a wider buffer."""
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(31)
    N, H, C, K = 5, 16, 64, 64
    x = torch.randn(N, H, H, C, generator=g)
    w4 = torch.randn(K, C, 4, 4, generator=g) * 0.1
    xb = torch.full((N, H, H, C + 12), 5.0, device=dev); xb[..., 8:8 + C] = x.to(dev)
    yb = torch.full((N, H // 2, H // 2, K + 4), -2.0, device=dev)
    ops.conv2d_wino(0, xb[..., 8:8 + C], ops.pack_weight(w4).to(dev), C, K, out=yb[..., 4:4 + K], k4s2=True)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w4, None, stride=2, padding=1).permute(0, 2, 3, 1)
    assert rel(yb[..., 4:4 + K].cpu(), ref) < TIGHT and (yb[..., :4] == -2.0).all()
    gy = torch.randn(N, H // 2, H // 2, K, generator=g)
    gb = torch.full((N, H // 2, H // 2, K + 8), 3.0, device=dev); gb[..., :K] = gy.to(dev)
    dxb = torch.full((N, H, H, C + 4), -1.0, device=dev)
    ops.conv2d_wino(1, gb[..., :K], ops.pack_weight(w4).to(dev), C, K, out=dxb[..., :C], k4s2=True)
    refd = F.conv_transpose2d(gy.permute(0, 3, 1, 2), w4, stride=2, padding=1).permute(0, 2, 3, 1)
    assert rel(dxb[..., :C].cpu(), refd) < TIGHT and (dxb[..., C:] == -1.0).all()
    # weight gradients: 4x4 stride 2 (gy half the size) and 3x3 stride 1 (gy as large as x)
    for k, st, gyt in ((4, 2, gy), (3, 1, torch.randn(N, H, H, K, generator=g))):
        w0 = torch.zeros(K, C, k, k, requires_grad=True)
        F.conv2d(x.permute(0, 3, 1, 2), w0, None, stride=st, padding=1).backward(gyt.permute(0, 3, 1, 2))
        gb2 = torch.full(tuple(gyt.shape[:3]) + (K + 8,), 3.0, device=dev); gb2[..., 4:4 + K] = gyt.to(dev)
        dwp = ops.conv2d_wino_wgrad(xb[..., 8:8 + C], gb2[..., 4:4 + K])
        assert rel(ops.unpack_weight(dwp, K, C, k, k).cpu(), w0.grad) < TIGHT


def test_wino_is_deterministic_and_batch_independent():
    N, H, W, C, K = 24, 8, 8, 32, 64
    x, w, b = _inputs(N, H, W, C, K, 7)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    y1 = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev))
    y2 = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev))
    assert torch.equal(y1, y2)
    y3 = ops.conv2d_wino(0, x[5:9].contiguous().to(dev), wp, C, K, bias=b.to(dev))
    assert torch.equal(y1[5:9], y3)          # an image's result does not depend on its block or its neighbours


def test_wino_rejects_what_it_cannot_run():
    dev = torch.device('cuda')
    for (N, H, W, C, K) in [(2, 16, 16, 24, 64), (2, 16, 16, 32, 48), (2, 12, 16, 32, 64), (2, 2, 2, 32, 64)]:
        x = torch.zeros(N, H, W, C, device=dev)
        wp = torch.zeros(9 * C, ops.round_up(K, 4), device=dev)
        with pytest.raises(RuntimeError):
            ops.conv2d_wino(0, x, wp, C, K)


# F(4x4, 3x3) (csrc/wino44.h): maps >= 8x8, input channels % 32, output channels % 64.  Round-off one order above F(2x2, 3x3)
# (interpolation points 0, +-1, +-2: observed 6e-7 ... 5e-6 of the tensor's max), two orders below the contract.
TIGHT44 = 1e-4
CASES44 = [
    (3, 16, 16, 32, 64),      # two images per item (box = image + halo of zeros), ragged last item
    (2, 32, 32, 32, 128),     # 4 x 8 tiles: one patch wide, two high; two cout blocks
    (5, 8, 8, 64, 64),        # eight images per item, ragged: the half-waves of a pass sit in different images
    (2, 64, 32, 32, 64),      # non-square: four patches high
    (3, 16, 32, 64, 192),     # one patch per image, three cout blocks
    (40, 16, 16, 32, 64),     # several items per block of the persistent grid
    (17, 8, 8, 32, 192),
    (1, 128, 64, 32, 64),     # 8 x 2 patches: boxes with real halos on every side
    (9, 16, 16, 128, 64),     # 16 chunks: eight pairs
]


@pytest.mark.parametrize('case', CASES44)
def test_wino44_forward(case):
    N, H, W, C, K = case
    x, w, b = _inputs(N, H, W, C, K, 21)
    add = torch.randn(N, H, W, K, generator=torch.Generator().manual_seed(22))
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev), slope=0.2, gain=1.3, f44=True)
    assert rel(y.cpu(), F.leaky_relu(ref, 0.2) * 1.3) < TIGHT44
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=None, ref=add.to(dev), slope=1.0, gain=1.0, f44=True)
    assert rel(y.cpu(), F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=1).permute(0, 2, 3, 1) + add) < TIGHT44
    # against F(2x2, 3x3) where both run: the two transforms agree to round-off
    if C % 16 == 0 and K % 64 == 0:
        y2 = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=None, ref=add.to(dev), slope=1.0, gain=1.0)
        assert rel(y, y2) < TIGHT44


@pytest.mark.parametrize('case', CASES44)
def test_wino44_data_gradient(case):
    N, H, W, K, C = case          # (roles swapped: gy has K channels, dx has C; the kernel needs K % 32 == 0, C % 64 == 0)
    g = torch.Generator().manual_seed(23)
    gy = torch.randn(N, H, W, K, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    act = torch.randn(N, H, W, C, generator=g)
    ref = F.conv_transpose2d(gy.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K, f44=True)
    assert rel(dx.cpu(), ref) < TIGHT44
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K, ref=act.to(dev), slope=0.2, gain=1.5, f44=True)
    assert rel(dx.cpu(), ref * torch.where(act > 0, 1.5, 0.3)) < TIGHT44


# F(4x4, 3x3) with 32-wide cout blocks (csrc/wino44n.h: StyleGAN2_512's 32 -> 32 channel layers): (N, H, W, C, K), K an odd multiple of 32
CASES44N = [
    (3, 16, 16, 32, 32),      # two images per item, ragged; four chunks (two pairs)
    (9, 8, 8, 32, 32),        # eight images per item, ragged
    (2, 32, 64, 32, 96),      # patches of 16 x 32 pixels, 2 x 2 per image; three cout blocks
    (1, 128, 64, 64, 32),     # 8 x 2 patches: boxes with real halos on every side; eight chunks
    (40, 16, 16, 32, 32),     # several items per block of the persistent grid
    (40, 4, 4, 32, 64),       # 4x4 maps (this variant whatever the cout count): a tile is an image, 32 per item, ragged; shared halos
    (70, 4, 4, 64, 96),       # three items per cout block, three cout blocks
]


@pytest.mark.parametrize('case', CASES44N)
def test_wino44n_forward(case):
    N, H, W, C, K = case
    x, w, b = _inputs(N, H, W, C, K, 41)
    add = torch.randn(N, H, W, K, generator=torch.Generator().manual_seed(42))
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev), slope=0.2, gain=1.3, f44=True)
    assert rel(y.cpu(), F.leaky_relu(ref, 0.2) * 1.3) < TIGHT44
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=None, ref=add.to(dev), slope=1.0, gain=1.0, f44=True)
    assert rel(y.cpu(), F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=1).permute(0, 2, 3, 1) + add) < TIGHT44
    y2 = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=None, ref=add.to(dev), slope=1.0, gain=1.0, f44=True)
    assert torch.equal(y, y2)


@pytest.mark.parametrize('case', CASES44N)
def test_wino44n_data_gradient(case):
    N, H, W, K, C = case          # (roles swapped: gy has K channels -- a multiple of 32 --, dx has C -- an odd multiple of 32)
    g = torch.Generator().manual_seed(43)
    gy = torch.randn(N, H, W, K, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    act = torch.randn(N, H, W, C, generator=g)
    ref = F.conv_transpose2d(gy.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K, f44=True)
    assert rel(dx.cpu(), ref) < TIGHT44
    dx = ops.conv2d_wino(1, gy.to(dev), wp, C, K, ref=act.to(dev), slope=0.2, gain=1.5, f44=True)
    assert rel(dx.cpu(), ref * torch.where(act > 0, 1.5, 0.3)) < TIGHT44


def test_wino44_channel_sliced_views_determinism_and_rejects():
    N, H, W, C, K = 6, 16, 16, 32, 64
    x, w, b = _inputs(N, H, W, C, K, 25)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    xb = torch.randn(N, H, W, C + 12, device=dev)
    xb[..., 4:4 + C] = x.to(dev)
    yb = torch.full((N, H, W, K + 16), 7.0, device=dev)
    ops.conv2d_wino(0, xb[..., 4:4 + C], wp, C, K, bias=b.to(dev), slope=0.1, gain=1.0, out=yb[..., 8:8 + K], f44=True)
    ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1), 0.1).permute(0, 2, 3, 1)
    assert rel(yb[..., 8:8 + K].cpu(), ref) < TIGHT44
    assert (yb[..., :8] == 7.0).all() and (yb[..., 8 + K:] == 7.0).all()
    y1 = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev), f44=True)
    y2 = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev), f44=True)
    assert torch.equal(y1, y2)
    y3 = ops.conv2d_wino(0, x[2:4].contiguous().to(dev), wp, C, K, bias=b.to(dev), f44=True)     # (an item = two images)
    assert torch.equal(y1[2:4], y3)
    for (h, c, k) in ((2, 32, 64), (16, 16, 64), (16, 32, 48), (12, 32, 64)):
        with pytest.raises(RuntimeError):
            ops.conv2d_wino(0, torch.zeros(2, h, h, c, device=dev), torch.zeros(9 * c, k, device=dev), c, k, f44=True)


# 3x3 stride-2 pad-0 layers on odd maps (StyleGAN2's blurred conv2, models/gan/stylegan2/layers.py:174-198) on F(2x2, 2x2)
# over the four phases with the zero planes skipped (csrc/wino23.h): (N, G, C, K), input (2G + 1)^2, output G^2
S3CASES = [
    (3, 16, 16, 64),         # 16 x 16 grid: two images per item, ragged; one chunk pair per phase
    (9, 8, 32, 128),         # 8 x 8 grid: 8 images per item, ragged; two cout blocks
    (40, 4, 16, 64),         # 4 x 4 grid: 32 images per item, ragged
    (2, 32, 16, 64),         # 32 x 32 grid: patches of 16 x 32 output pixels, 2 x 1 per image
    (1, 64, 32, 192),        # 4 x 2 patches per image, three cout blocks
    (70, 8, 48, 64),         # several items per block; Cin = 48: six chunks per phase
]


@pytest.mark.parametrize('case', S3CASES)
def test_wino23_forward(case):
    N, G, C, K = case
    H = 2 * G + 1
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, H, H, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    add = torch.randn(N, G, G, K, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=2).permute(0, 2, 3, 1)
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, bias=b.to(dev), slope=0.2, gain=1.4, k3s2=True)
    assert rel(y.cpu(), F.leaky_relu(ref, 0.2) * 1.4) < TIGHT
    y = ops.conv2d_wino(0, x.to(dev), wp, C, K, ref=add.to(dev), k3s2=True)
    assert rel(y.cpu(), F.conv2d(x.permute(0, 3, 1, 2), w, None, stride=2).permute(0, 2, 3, 1) + add) < TIGHT


def test_wino23_forward_on_a_non_square_map():
    """33 x 129 -> 16 x 64: one patch high, two wide (the models' maps are square; the kernel's patch walk is not)."""
    N, C, K = 3, 16, 64
    g = torch.Generator().manual_seed(33)
    x = torch.randn(N, 33, 129, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    dev = torch.device('cuda')
    y = ops.conv2d_wino(0, x.to(dev), ops.pack_weight(w).to(dev), C, K, bias=b.to(dev), slope=0.2, gain=1.0, k3s2=True)
    ref = F.leaky_relu(F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=2), 0.2).permute(0, 2, 3, 1)
    assert tuple(y.shape) == (N, 16, 64, K)
    assert rel(y.cpu(), ref) < TIGHT


def test_wino23_channel_sliced_views_determinism_and_rejects():
    N, G, C, K = 5, 16, 32, 64
    H = 2 * G + 1
    g = torch.Generator().manual_seed(32)
    x = torch.randn(N, H, H, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.1
    dev = torch.device('cuda')
    wp = ops.pack_weight(w).to(dev)
    xb = torch.randn(N, H, H, C + 12, device=dev)
    xb[..., 4:4 + C] = x.to(dev)
    yb = torch.full((N, G, G, K + 16), 7.0, device=dev)
    ops.conv2d_wino(0, xb[..., 4:4 + C], wp, C, K, out=yb[..., 8:8 + K], k3s2=True)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, None, stride=2).permute(0, 2, 3, 1)
    assert rel(yb[..., 8:8 + K].cpu(), ref) < TIGHT
    assert (yb[..., :8] == 7.0).all() and (yb[..., 8 + K:] == 7.0).all()
    y1 = ops.conv2d_wino(0, x.to(dev), wp, C, K, k3s2=True)
    y2 = ops.conv2d_wino(0, x.to(dev), wp, C, K, k3s2=True)
    assert torch.equal(y1, y2)
    y3 = ops.conv2d_wino(0, x[2:4].contiguous().to(dev), wp, C, K, k3s2=True)      # (an item = two images)
    assert torch.equal(y1[2:4], y3)
    with pytest.raises(RuntimeError):          # forward only
        ops.conv2d_wino(1, torch.zeros(2, G, G, K, device=dev), wp, C, K, k3s2=True)
    for (h, c, k) in ((32, 32, 64), (33, 8, 64), (33, 32, 32), (25, 32, 64), (5, 32, 64)):
        with pytest.raises(RuntimeError):
            ops.conv2d_wino(0, torch.zeros(2, h, h, c, device=dev), torch.zeros(9 * c, k, device=dev), c, k, k3s2=True)


# the 3x3 stride-1 layers of the BASELINE workloads: (N, H, C) -- SNDCGAN at 3N = 1536, StyleGAN2_512 at 3N = 48
PLANNED = [(1536, 16, 128), (1536, 8, 256), (1536, 4, 512), (48, 128, 128), (48, 64, 256), (48, 32, 512), (48, 256, 64)]


@pytest.mark.parametrize('shape', PLANNED)
def test_the_plan_takes_winograd_at_the_baseline_shapes_and_matches_torch(shape):
    """conv2d_fwd / conv2d_dgrad (the calls the models make) at full size: the path query says Winograd, and the first and
    last images match PyTorch-CPU."""
    N, H, C = shape
    K = C
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, H, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    b = torch.randn(K, generator=g)
    wp = ops.pack_weight(w).to(dev)
    d = ops.make_desc(N, H, H, C, K, 3, 3, 1, 1, C, K, wp.stride(0))
    want = 11 if H == 4 else 9      # (F(4x4, 3x3) on all of them; the 4x4 maps on its variant with 32-wide cout blocks: full rounds)
    assert lib().raw('contrad_conv2d_path')(ctypes.byref(d), 0) == want
    assert lib().raw('contrad_conv2d_path')(ctypes.byref(d), 1) == want
    assert abs(lib().raw('contrad_conv2d_executed_fraction')(ctypes.byref(d), 0) - 0.25) < 1e-12
    xd = x.to(dev)
    y = ops.conv2d_fwd(xd, wp, b.to(dev), K, 3, 3, 1, 1, slope=0.1, gain=1.0)
    sel = [0, 1, N - 2, N - 1] if H <= 64 else [0, N - 1]
    xs = x[sel]
    ref = F.leaky_relu(F.conv2d(xs.permute(0, 3, 1, 2), w, b, padding=1), 0.1).permute(0, 2, 3, 1)
    assert rel(y[sel].cpu(), ref) < TIGHT44
    dx = ops.conv2d_dgrad(y, wp, (N, H, H, C), 3, 3, 1, 1, act_ref=xd, slope=0.1, gain=1.0)
    refd = F.conv_transpose2d(y[sel].cpu().permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) * torch.where(xs > 0, 1.0, 0.1)
    assert rel(dx[sel].cpu(), refd) < TIGHT44
    # weight gradient at full size: Winograd by the plan; against the direct kernels' result on HALF the images twice (linearity:
    # dW(all) = dW(first half) + dW(second half), each half small enough for ... the same plan) and against PyTorch-CPU on a
    # sub-batch through the forced entry point
    assert lib().raw('contrad_conv2d_path')(ctypes.byref(d), 2) == 7
    dbias = torch.empty(K, device=dev)
    dwp = ops.conv2d_wgrad(xd, y, 3, 3, 1, 1, dbias=dbias)
    nb = 8 if H <= 64 else 2
    acc = torch.zeros_like(dwp)
    accb = torch.zeros_like(dbias)
    step = N // 4
    for i in range(0, N, step):              # four quarter-batches on whichever kernel the plan picks for them
        db = torch.empty(K, device=dev)
        acc += ops.conv2d_wgrad(xd[i:i + step], y[i:i + step], 3, 3, 1, 1, dbias=db)
        accb += db
    assert rel(dwp, acc) < 1e-4 and rel(dbias, accb) < 1e-4
    xs2, ys2 = x[:nb], y[:nb].cpu()
    w0 = torch.zeros(K, C, 3, 3, requires_grad=True)
    F.conv2d(xs2.permute(0, 3, 1, 2), w0, None, padding=1).backward(ys2.permute(0, 3, 1, 2))
    dws = ops.unpack_weight(ops.conv2d_wino_wgrad(xd[:nb], y[:nb]), K, C, 3, 3).cpu()
    assert rel(dws, w0.grad) < TIGHT


def test_the_plan_takes_winograd_for_the_32_channel_layers_at_512():
    """StyleGAN2_512's 32 -> 32 channel 3x3 layers (3N = 48 images at 512^2): forward and data gradient on F(4x4, 3x3) with 32-wide
    cout blocks (csrc/wino44n.h), the weight gradient on its dedicated direct kernel; computed here on 4 images."""
    N, H, C, K = 4, 512, 32, 32
    dev = torch.device('cuda')
    wp0 = torch.zeros(9 * C, K)
    for n in (48, 16, N):
        d = ops.make_desc(n, H, H, C, K, 3, 3, 1, 1, C, K, wp0.stride(0))
        assert lib().raw('contrad_conv2d_path')(ctypes.byref(d), 0) == 11
        assert lib().raw('contrad_conv2d_path')(ctypes.byref(d), 1) == 11
        assert 0 <= lib().raw('contrad_conv2d_path')(ctypes.byref(d), 2) < 7
    g = torch.Generator().manual_seed(13)
    x = torch.randn(N, H, H, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    b = torch.randn(K, generator=g)
    wp = ops.pack_weight(w).to(dev)
    xd = x.to(dev)
    y = ops.conv2d_fwd(xd, wp, b.to(dev), K, 3, 3, 1, 1, slope=0.2, gain=1.0)
    sel = [0, N - 1]
    ref = F.leaky_relu(F.conv2d(x[sel].permute(0, 3, 1, 2), w, b, padding=1), 0.2).permute(0, 2, 3, 1)
    assert rel(y[sel].cpu(), ref) < TIGHT44
    dx = ops.conv2d_dgrad(y, wp, (N, H, H, C), 3, 3, 1, 1, act_ref=xd, slope=0.2, gain=1.0)
    refd = F.conv_transpose2d(y[sel].cpu().permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1) * torch.where(x[sel] > 0, 1.0, 0.2)
    assert rel(dx[sel].cpu(), refd) < TIGHT44


# the blurred 3x3 stride-2 layers of StyleGAN2_512 at 3N = 48 that the plan gives to csrc/wino23.h: (N, G, C, K)
PLANNED23 = [(48, 256, 32, 64), (48, 128, 64, 128), (48, 64, 128, 256), (48, 32, 256, 512)]


@pytest.mark.parametrize('shape', PLANNED23)
def test_the_plan_takes_the_strided_3x3_winograd_forward_at_the_baseline_shapes(shape):
    """conv2d_fwd (the call ResBlock.conv2 makes, with the skip path's addend) at full size: path 10, 25 / 36 of the dense
    products, first and last image against PyTorch-CPU; data and weight gradient of the same layer stay on the direct engine."""
    N, G, C, K = shape
    H = 2 * G + 1
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(12)
    x = torch.randn(N, H, H, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    b = torch.randn(K, generator=g)
    add = torch.randn(N, G, G, K, generator=g)
    wp = ops.pack_weight(w).to(dev)
    d = ops.make_desc(N, H, H, C, K, 3, 3, 2, 0, C, K, wp.stride(0))
    assert lib().raw('contrad_conv2d_path')(ctypes.byref(d), 0) == 10
    assert abs(lib().raw('contrad_conv2d_executed_fraction')(ctypes.byref(d), 0) - 25.0 / 36.0) < 1e-12
    assert 0 <= lib().raw('contrad_conv2d_path')(ctypes.byref(d), 1) < 7          # (direct engine)
    assert 0 <= lib().raw('contrad_conv2d_path')(ctypes.byref(d), 2) < 7
    y = ops.conv2d_fwd(x.to(dev), wp, b.to(dev), K, 3, 3, 2, 0, slope=0.2, gain=1.0, addend=add.to(dev))
    sel = [0, N - 1]
    ref = F.leaky_relu(F.conv2d(x[sel].permute(0, 3, 1, 2), w, b, stride=2), 0.2).permute(0, 2, 3, 1) + add[sel]
    assert rel(y[sel].cpu(), ref) < TIGHT
