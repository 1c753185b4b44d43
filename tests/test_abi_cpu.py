"""C-ABI checks that need no GPU: the library loads, and exports every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

from contrad_amd import _lib


@pytest.fixture(scope='module')
def built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.lib()


def test_header_parses():
    protos = _lib.parse_header()
    assert 'contrad_conv2d_fwd' in protos and 'contrad_contrast_fwd' in protos
    src = open(_lib.HEADER_PATH).read()
    declared = set(re.findall(r'\b(contrad_\w+)\s*\(', re.sub(r'/\*.*?\*/', '', src, flags=re.S)))
    assert declared == set(protos), declared ^ set(protos)


def test_library_exports_every_declared_symbol(built):
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in built.protos:
        assert hasattr(dll, name), name
    assert built.raw('contrad_abi_version')() >= 1


def test_argument_errors_are_reported_without_gpu(built):
    # NULL descriptor -> -EINVAL, raised as RuntimeError by the host wrapper (TORCH_CHECK analogue)
    with pytest.raises(RuntimeError):
        built.call('contrad_conv2d_fwd', None, None, None, None, None, 1.0, 1.0, None, 0, None)
    d = _lib.ConvDesc(1, 8, 8, 4, 4, 8, 8, 4, 4, 3, 3, 3, 1, 4)   # stride 3 unsupported
    assert built.raw('contrad_conv2d_wgrad_workspace_bytes')(ctypes.byref(d)) < 0
    d = _lib.ConvDesc(2, 8, 8, 4, 4, 8, 8, 8, 8, 3, 3, 1, 1, 8)
    assert built.raw('contrad_conv2d_wgrad_workspace_bytes')(ctypes.byref(d)) >= 3 * 3 * 4 * 8 * 4


def _desc(N, H, C, K, k, s, p):
    Ho = (H + 2 * p - k) // s + 1
    return _lib.ConvDesc(N, H, H, C, C, Ho, Ho, K, K, k, k, s, p, (K + 3) // 4 * 4)


# SNDCGAN discriminator trunk at the BASELINE batch (3N = 1536 images): (H, Cin, Cout, k, stride, pad)
_SNDCGAN = [(32, 64, 128, 4, 2, 1), (16, 128, 128, 3, 1, 1), (16, 128, 256, 4, 2, 1), (8, 256, 256, 3, 1, 1),
            (8, 256, 512, 4, 2, 1), (4, 512, 512, 3, 1, 1)]


def test_launch_plans_are_host_logic(built):
    """Kernel family, tile and split-K decisions are pure host code behind the C ABI: every igemm layer of the headline
    configuration must land on the lean loop at the big tile, odd shapes on the general kernel, and the merged head GEMM
    of a small per-rank batch on the split-K forward (with a workspace to match)."""
    path, tile = built.raw('contrad_conv2d_path'), built.raw('contrad_conv2d_tile')
    bm, bn = ctypes.c_int(0), ctypes.c_int(0)
    for (H, C, K, k, s, p) in _SNDCGAN:
        d = _desc(1536, H, C, K, k, s, p)
        for mode in (0, 1, 2):
            # lean loop everywhere.  Padding tap-positions are skipped (path 3): forward and stride-1 data gradient of every
            # layer (pixel-major tiles on the 4 x 4 maps, border classes on the larger ones); of the strided data gradients
            # the 4x4 stride-2 layer onto the 8 x 8 dx map (0.77 of its tap-positions valid: pixel-major inside the parity
            # classes, slot-balanced class-major order, round 4) -- the larger maps stay image-major; the weight gradient
            # where at most 0.85 of the tap-positions are valid
            # Round 6: forward and data gradient of the 3x3 stride-1 layers run on the Winograd kernel (path 7, csrc/wino.h; one
            # persistent 512-thread block per CU; the workspace holds the transformed filter)
            # ... and the 4x4 stride-2 layers on F(2x2, 2x2) over the four input phases (path 8, csrc/wino22.h) -- except the
            # forward of 256 -> 512 channels: 384 items of 128 tiles x 64 channels are a round and a half of the 256 CUs, where
            # the direct kernel (pixel-major tiles, padding taps skipped) is faster
            Ho = (H + 2 * p - k) // s + 1
            # ... and, later in round 6, forward and data gradient of the 3x3 layers on 16x16 / 8x8 maps on F(4x4, 3x3) (path 9,
            # csrc/wino44.h: 1536 / 768 items of 512 pixels x 64 channels); the 4x4 maps on its variant with 32-wide cout blocks
            # (path 11, csrc/wino44n.h: 768 items of 32 images x 32 channels = three rounds where 64-wide blocks give 1.5)
            if k == 3:
                want = (9 if H >= 8 else 11) if mode != 2 else 7   # (the weight gradient: wino_wgrad_kernel, F(3x3, 2x2), one block per CU over split tile ranges)
            elif not (mode == 0 and C == 256):
                want = 8
            else:
                want = 3
            assert path(ctypes.byref(d), mode) == want, (H, C, K, mode)
            if want in (7, 8, 9, 11):
                assert built.raw('contrad_conv2d_grid_blocks')(ctypes.byref(d), mode, 1) == 256
                frac = 0.25 if want in (9, 11) else 4.0 / 9.0 if want == 7 else (9.0 / 16.0 if mode != 2 else 1.0)
                if want in (9, 11):
                    assert built.raw('contrad_conv2d_wino44_ok')(ctypes.byref(d), mode) == 1
                if mode != 2 or want == 7:
                    assert abs(built.raw('contrad_conv2d_executed_fraction')(ctypes.byref(d), mode) - frac) < 1e-12
                assert built.raw('contrad_conv2d_wino_ok')(ctypes.byref(d), mode) == 1
                continue
            assert tile(ctypes.byref(d), mode, ctypes.byref(bm), ctypes.byref(bn)) == 0
            want_bn = 64 if (mode == 1 and C == 64) else 128        # dgrad's columns are the input channels
            assert (bm.value, bn.value) == (128, want_bn), (H, C, K, mode, bm.value, bn.value)
        want_ws = 16 * C * K * 4 if k == 3 else 36 * C * K * 4       # the transformed filter: 16 planes / 4 phases x 9 planes
        plan_ws = 36 * C * K * 4 if k == 3 else want_ws  # (F(4x4, 3x3): 36 planes)
        assert built.raw('contrad_conv2d_fwd_workspace_bytes')(ctypes.byref(d)) == (plan_ws if not (k == 4 and C == 256) else 0)
        assert built.raw('contrad_conv2d_dgrad_workspace_bytes')(ctypes.byref(d)) == plan_ws
        assert built.raw('contrad_conv2d_wino_workspace_bytes')(ctypes.byref(d), 0) == want_ws
        if k == 3:
            assert built.raw('contrad_conv2d_wino44_workspace_bytes')(ctypes.byref(d)) == 36 * C * K * 4
            assert built.raw('contrad_conv2d_wino44_ok')(ctypes.byref(d), 0) == 1      # (4x4 maps too since csrc/wino44n.h)
        # weight gradient: 256 / (row blocks x 64-wide k blocks) split slabs of the packed gradient + the bias partials
        splits = 256 // ((C // 64) * (K // 64)) if k == 3 else 256 // ((4 * C // 128) * (K // 64))
        assert built.raw('contrad_conv2d_wgrad_workspace_bytes')(ctypes.byref(d)) == splits * (k * k * C + 1) * K * 4
        assert built.raw('contrad_conv2d_wino_workspace_bytes')(ctypes.byref(d), 2) == splits * (k * k * C + 1) * K * 4
        assert built.raw('contrad_conv2d_grid_blocks')(ctypes.byref(d), 2, 1) == 256
    # Cin = 3 / Cout = 1 / 513 channels: general kernel, scalar or float4 gathers
    assert path(ctypes.byref(_desc(8, 32, 3, 64, 3, 1, 1)), 0) == 0
    d1 = _desc(8, 1, 512, 1, 1, 1, 0)                      # the 512 -> 1 logit: its own one-wave-per-row kernel forward,
    assert path(ctypes.byref(d1), 0) == 5                  # the general kernel for its two gradients
    assert path(ctypes.byref(d1), 1) == 0 and path(ctypes.byref(d1), 2) == 0
    assert built.raw('contrad_conv2d_grid_blocks')(ctypes.byref(d1), 0, 1) == 2
    assert path(ctypes.byref(_desc(8, 4, 516, 512, 3, 1, 1)), 0) == 1
    # 32-column GEMM (StyleGAN2 at 512x512): the 128x32 tile
    d = _desc(48, 64, 32, 32, 3, 1, 1)
    assert tile(ctypes.byref(d), 0, ctypes.byref(bm), ctypes.byref(bn)) == 0 and (bm.value, bn.value) == (128, 32)
    # merged head layer at 3N = 192 rows: split-K forward with partial slabs of M x N floats each
    d = _desc(192, 1, 8192, 1536, 1, 1, 0)
    nbytes = built.raw('contrad_conv2d_fwd_workspace_bytes')(ctypes.byref(d))
    assert nbytes > 0 and nbytes % (192 * 1536 * 4) == 0 and 2 <= nbytes // (192 * 1536 * 4) <= 16
    # deep 3x3 layer at the 4x4 level, 3N = 192 images: split-K data gradient (slabs in dx's layout); never at 1536
    d = _desc(192, 4, 512, 512, 3, 1, 1)
    nbytes = built.raw('contrad_conv2d_dgrad_workspace_bytes')(ctypes.byref(d))
    assert nbytes > 0 and nbytes % (192 * 16 * 512 * 4) == 0 and 2 <= nbytes // (192 * 16 * 512 * 4) <= 16
    assert built.raw('contrad_conv2d_dgrad_workspace_bytes')(ctypes.byref(_desc(1536, 4, 516, 512, 3, 1, 1))) == 0   # (not Winograd: Cin)
    # Winograd is planned only for launches of about an item (64 tiles x 64 output channels) per CU or more with a last round that
    # is not mostly empty; it needs input channels % 16, output channels % 64 and power-of-two maps
    wok = built.raw('contrad_conv2d_wino_ok')
    assert path(ctypes.byref(_desc(192, 4, 512, 512, 3, 1, 1)), 0) != 7 and wok(ctypes.byref(_desc(192, 4, 512, 512, 3, 1, 1)), 0) == 1
    assert path(ctypes.byref(_desc(96, 16, 128, 128, 3, 1, 1)), 0) == 7         # 192 items: most of a round (F(4x4, 3x3) wants 230 of its own)
    assert path(ctypes.byref(_desc(192, 16, 128, 128, 3, 1, 1)), 0) == 11       # 384 items of 32 tiles x 32 couts (192 of 64 would not fill the chip)
    assert path(ctypes.byref(_desc(160, 16, 128, 128, 3, 1, 1)), 0) != 7        # 320 items: the second round a quarter full
    assert wok(ctypes.byref(_desc(8, 16, 24, 64, 3, 1, 1)), 0) == 0 and wok(ctypes.byref(_desc(8, 16, 32, 48, 3, 1, 1)), 0) == 0
    assert wok(ctypes.byref(_desc(8, 16, 32, 48, 3, 1, 1)), 1) == 0 and wok(ctypes.byref(_desc(8, 16, 64, 48, 3, 1, 1)), 1) == 1
    assert wok(ctypes.byref(_desc(8, 12, 32, 64, 3, 1, 1)), 0) == 0 and wok(ctypes.byref(_desc(8, 16, 32, 64, 3, 2, 1)), 0) == 0
    # F(4x4, 3x3): a full round of its 512-pixel items (from 230), input channels % 32, maps >= 8x8
    w44 = built.raw('contrad_conv2d_wino44_ok')
    assert path(ctypes.byref(_desc(256, 16, 128, 128, 3, 1, 1)), 0) == 9 and path(ctypes.byref(_desc(256, 16, 128, 128, 3, 1, 1)), 1) == 9   # 256 items
    assert path(ctypes.byref(_desc(16, 64, 256, 256, 3, 1, 1)), 0) == 9         # StyleGAN2_512, 64x64 maps of 16 images: 512 items
    assert path(ctypes.byref(_desc(16, 16, 512, 512, 3, 1, 1)), 0) != 9         # 64 items
    assert w44(ctypes.byref(_desc(8, 16, 48, 64, 3, 1, 1)), 0) == 0 and w44(ctypes.byref(_desc(8, 2, 64, 64, 3, 1, 1)), 0) == 0
    # ... its variant with 32-wide cout blocks (path 11, csrc/wino44n.h): output channels that are not whole 64-wide blocks, and the 4x4 maps
    assert w44(ctypes.byref(_desc(8, 4, 64, 64, 3, 1, 1)), 0) == 1 and w44(ctypes.byref(_desc(8, 16, 32, 96, 3, 1, 1)), 0) == 1
    assert w44(ctypes.byref(_desc(8, 16, 32, 48, 3, 1, 1)), 0) == 0
    assert path(ctypes.byref(_desc(48, 512, 32, 32, 3, 1, 1)), 0) == 11 and path(ctypes.byref(_desc(16, 512, 32, 32, 3, 1, 1)), 1) == 11
    assert path(ctypes.byref(_desc(1536, 4, 512, 512, 3, 1, 1)), 0) == 11 and path(ctypes.byref(_desc(1536, 8, 256, 256, 3, 1, 1)), 0) == 9
    assert abs(built.raw('contrad_conv2d_executed_fraction')(ctypes.byref(_desc(48, 512, 32, 32, 3, 1, 1)), 0) - 0.25) < 1e-12
    assert w44(ctypes.byref(_desc(8, 8, 64, 64, 3, 1, 1)), 0) == 1 and w44(ctypes.byref(_desc(8, 8, 64, 64, 3, 1, 1)), 2) == 0
    assert built.raw('contrad_conv2d_dgrad_workspace_bytes')(ctypes.byref(_desc(192, 8, 260, 512, 4, 2, 1))) == 0   # strided (not F(2x2,2x2): Cin)
    # contrastive column splits: ~256 blocks
    assert built.raw('contrad_contrast_workspace_bytes')(1024, 128) == 16 * 1024 * 128 * 4


def _axis_runs(out_ext, in_ext, k, s, p, mode):
    """Runs of consecutive output (mode 0) / dx (mode 1, stride 1) coordinates with the same set of non-padding taps."""
    runs, prev = [], None
    for o in range(out_ext):
        m = tuple(t for t in range(k) if 0 <= (o * s - p + t if mode == 0 else o + p - t) < in_ext)
        if m != prev:
            runs.append([o, 0, len(m)])
            prev = m
        runs[-1][1] += 1
    return runs


def test_padding_skipping_tile_plans_without_gpu(built):
    """The tiles that skip padding taps are planned on the host: restate the plan in Python and
    compare what the C ABI reports -- kernel family 3, the share of the nominal multiply-adds that is issued, and the
    workgroup count, which for border classes is 8 x (the largest per-XCD share of the class tiles) x column tiles."""
    path, tile = built.raw('contrad_conv2d_path'), built.raw('contrad_conv2d_tile')
    frac, blocks = built.raw('contrad_conv2d_executed_fraction'), built.raw('contrad_conv2d_grid_blocks')
    bm, bn = ctypes.c_int(0), ctypes.c_int(0)
    N = 1536
    seen = set()
    # (the 3x3 stride-1 layers with channel counts no Winograd kernel takes -- they need output channels % 32 --: the
    # direct kernels and their padding-skipping tiles serve them as they served 128 / 256 / 512 channels until round 5)
    nowino = [(H, C - 16, K - 16, k, s, p) for (H, C, K, k, s, p) in _SNDCGAN]
    for (H, C, K, k, s, p) in nowino:
        d = _desc(N, H, C, K, k, s, p)
        Ho = (H + 2 * p - k) // s + 1
        for mode in (0, 1):
            if mode == 1 and s != 1:
                if H == 8:      # dx map 8 x 8 from the 4 x 4 gy: per axis 14 of 16 tap-positions valid -> pixel-major, class-major
                    assert path(ctypes.byref(d), mode) == 3 and abs(frac(ctypes.byref(d), mode) - (14.0 / 16.0) ** 2) < 1e-12
                    assert tile(ctypes.byref(d), mode, ctypes.byref(bm), ctypes.byref(bn)) == 0
                    assert blocks(ctypes.byref(d), mode, 1) == -(-N // bm.value) * 16 * 4 * -(-C // bn.value)
                    seen.add('strided pixel-major')
                else:
                    assert frac(ctypes.byref(d), mode) == 1.0       # larger maps: image-major, nothing skipped
                continue
            ext, inn = (Ho, H) if mode == 0 else (H, Ho)
            runs = _axis_runs(ext, inn, k, s, p, mode)
            classes = sorted(((a[2] * b[2], a[1] * b[1]) for a in runs for b in runs), key=lambda c: -c[0])
            valid = sum(t * n for t, n in classes) / float(ext * ext * (k * k if mode == 0 or s == 1 else 1))
            assert path(ctypes.byref(d), mode) == 3
            assert abs(frac(ctypes.byref(d), mode) - valid) < 1e-12, (H, k, mode)
            assert tile(ctypes.byref(d), mode, ctypes.byref(bm), ctypes.byref(bn)) == 0
            tiles_n = -(-(K if mode == 0 else C) // bn.value)
            if valid <= 0.80:                                        # pixel-major: image blocks x pixels
                want = -(-N // bm.value) * ext * ext * tiles_n
                seen.add('pixel-major')
            else:                                                    # border classes, an eighth of each class per XCD
                per_class = [-(-N * npix // bm.value) for _, npix in classes]
                most = max(sum(((x + 1) * n >> 3) - (x * n >> 3) for n in per_class) for x in range(8))
                want = 8 * most * tiles_n
                assert sum(npix for _, npix in classes) == ext * ext and len(classes) == 9
                seen.add('border classes')
            assert blocks(ctypes.byref(d), mode, 1) == want, (H, k, mode, want)
    assert seen == {'pixel-major', 'border classes', 'strided pixel-major'}
    # too few images for a tile of the smallest class on every XCD: image-major tiles, everything issued
    d = _desc(192, 8, 256, 224, 3, 1, 1)        # (224 output channels: not a Winograd shape)
    assert path(ctypes.byref(d), 0) == 2 and frac(ctypes.byref(d), 0) == 1.0
    assert path(ctypes.byref(d), 2) == 3 and abs(frac(ctypes.byref(d), 2) - 484.0 / 576.0) < 1e-12   # (WGRAD: pixel-major positions)
    # one rank of the 8-GPU headline config (3N = 192 images): Winograd from 150 items of a single round (0.59 of the CUs)
    assert path(ctypes.byref(_desc(192, 8, 256, 256, 3, 1, 1)), 0) == 7 and path(ctypes.byref(_desc(192, 8, 256, 256, 3, 1, 1)), 2) == 7


def test_shipped_library_reads_no_environment(built):
    """SURVEY.md 8b (no hidden state): the development switches of the conv engine (tile modes, forced tiles, split-K
    on / off ...) are compiled into libcontrad_hip_dev.so only; the shipped library neither imports getenv nor carries
    any of their names."""
    import subprocess
    from contrad_amd import build
    def names(path):
        return subprocess.run(['strings', path], capture_output=True, text=True, check=True).stdout
    def imports(path):
        return subprocess.run(['nm', '-D', path], capture_output=True, text=True, check=True).stdout
    assert 'CONTRAD_' not in names(build.LIB) and 'getenv' not in imports(build.LIB)
    assert 'CONTRAD_TILEMODE' in names(build.DEV_LIB) and 'getenv' in imports(build.DEV_LIB)
    # the tuner's plan override (tools/tune_plans.py) is an entry point of the development library only
    assert 'contrad_dev_plan_override' in imports(build.DEV_LIB) and 'contrad_dev_plan_override' not in imports(build.LIB)


def test_plan_override_changes_the_reported_plan():
    """tools/tune_plans.py forces tile and split count through contrad_dev_plan_override (dev library): the plan queries
    must report what was forced (they share the plan functions with the launchers), and 0, 0, 0 must restore the model."""
    from contrad_amd import build, ops
    dev = ctypes.CDLL(build.DEV_LIB)
    dev.contrad_dev_plan_override.restype, dev.contrad_dev_plan_override.argtypes = None, [ctypes.c_int] * 3
    d = ops.make_desc(192, 8, 8, 256, 256, 3, 3, 1, 1, 256, 256, 256)
    bm, bn = ctypes.c_int(0), ctypes.c_int(0)
    plans = {}
    for forced in ((0, 0, 0), (64, 64, 1), (128, 64, 1), (0, 0, 0)):
        dev.contrad_dev_plan_override(*forced)
        per_mode = []
        for mode in (0, 1, 2):
            assert dev.contrad_conv2d_tile(ctypes.byref(d), mode, ctypes.byref(bm), ctypes.byref(bn)) == 0
            per_mode.append((bm.value, bn.value))
        plans.setdefault(forced, []).append(per_mode)
        if forced[0]:
            assert all(t == forced[:2] for t in per_mode), (forced, per_mode)
    assert plans[(0, 0, 0)][0] == plans[(0, 0, 0)][1]          # the model's plan is back


def test_slot_balanced_tile_order_without_gpu(built):
    """Round 4's block order of the pixel-major launches is host logic ("How blocks reach CUs"): the
    table must be a permutation of the launch's M-tiles, and under the dispatch model it was derived from -- an XCD owns a
    contiguous run of blocks (xcd_remap), dealt round-robin over its 32 CUs -- every CU of the headline launches gets the same
    work to within one tap-unit (3x3 on a 4x4 map at 1 536 images: 18 or 19 of them; heaviest-first gave 24 vs 16)."""
    order = built.raw('contrad_conv2d_tile_order')
    tile = built.raw('contrad_conv2d_tile')
    bm, bn = ctypes.c_int(0), ctypes.c_int(0)

    def taps_fwd(H, k, s, p, Ho):
        ax = [sum(0 <= o * s - p + t < H for t in range(k)) for o in range(Ho)]
        return [a * b for a in ax for b in ax]

    def taps_dgrad(H, k, p, Ho):            # stride 1
        ax = [sum(0 <= h + p - t < Ho for t in range(k)) for h in range(H)]
        return [a * b for a in ax for b in ax]

    def cu_loads(table, taps, npix, tiles_n, runs, mirror=None):
        nb = len(table) * tiles_n
        q, r = divmod(nb, runs)
        loads = []
        for x in range(runs):
            b0 = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
            b1 = b0 + q + (1 if x < r else 0)
            cu = [0] * 32
            for b in range(b0, b1):
                cu[(b - b0) % 32] += taps[table[b // tiles_n] % npix]
            loads += cu
        return loads

    buf = (ctypes.c_ubyte * 256)()
    N = 1536
    # 3x3 pad 1, 512 -> 512 on a 4x4 map: forward and data gradient
    for mode in (0, 1):
        d = _desc(N, 4, 512, 512, 3, 1, 1)
        n = order(ctypes.byref(d), mode, buf, 256)
        assert tile(ctypes.byref(d), mode, ctypes.byref(bm), ctypes.byref(bn)) == 0
        assert n == -(-N // bm.value) * 16 and sorted(buf[:n]) == list(range(n))
        taps = taps_fwd(4, 3, 1, 1, 4) if mode == 0 else taps_dgrad(4, 3, 1, 4)
        loads = cu_loads(list(buf[:n]), taps, 16, 512 // bn.value, 8)
        assert max(loads) - min(loads) <= 1 and max(loads) == 19, (mode, min(loads), max(loads))
    # 4x4 stride 2 pad 1, 256 -> 512, 8x8 -> 4x4: forward (one table) ...
    d = _desc(N, 8, 256, 512, 4, 2, 1)
    n = order(ctypes.byref(d), 0, buf, 256)
    assert tile(ctypes.byref(d), 0, ctypes.byref(bm), ctypes.byref(bn)) == 0
    assert n == -(-N // bm.value) * 16 and sorted(buf[:n]) == list(range(n))
    loads = cu_loads(list(buf[:n]), taps_fwd(8, 4, 2, 1, 4), 16, 512 // bn.value, 8)
    assert max(loads) <= 1.08 * sum(loads) / len(loads)
    # ... and its data gradient: class-major launch, one table for the four parity classes (each = two XCD runs)
    n = order(ctypes.byref(d), 1, buf, 256)
    assert tile(ctypes.byref(d), 1, ctypes.byref(bm), ctypes.byref(bn)) == 0
    assert n == -(-N // bm.value) * 16 and sorted(buf[:n]) == list(range(n))
    ax = [1, 2, 2, 2]                        # class (0,0): valid taps per dx row / column of the class (hq = 0 loses one)
    loads = cu_loads(list(buf[:n]), [a * b for a in ax for b in ax], 16, 256 // bn.value, 2)
    assert max(loads) <= 1.10 * sum(loads) / len(loads)
    # launches without such a table
    assert order(ctypes.byref(_desc(N, 16, 128, 128, 3, 1, 1)), 0, buf, 256) == 0     # border classes
    assert order(ctypes.byref(_desc(N, 32, 64, 128, 4, 2, 1)), 1, buf, 256) == 0      # strided, image-major
