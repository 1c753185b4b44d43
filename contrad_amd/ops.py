"""Thin, allocation-explicit Python wrappers over the C ABI (include/contrad_hip.h).

Every function takes CUDA fp32 torch tensors purely as device-memory handles (``data_ptr()``) and
launches on ``torch.cuda.current_stream()``, so the calls are safe from both the Python main thread
(forward) and the autograd engine thread (backward) -- the threading contract of SURVEY.md 8(b).
Activations are NHWC: a tensor of shape (N, H, W, C) whose last dim is contiguous; the pixel stride
``stride(2)`` is the leading dimension (allows channel-sliced views of wider buffers).
"""
import ctypes

import torch

from ._lib import ConvDesc, lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_handle():
    """Raw hipStream_t of torch's current stream on the current device (the direct C accessor when this torch build has
    it: torch.cuda.current_stream() builds a Stream object per call, ~9 us x 50 launches per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _stream():
    return ctypes.c_void_p(_stream_handle())


def _chk(t, name):
    if t is None:
        return
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError('contrad_hip: %s must be a CUDA float32 tensor (got %s, %s)' % (name, t.device, t.dtype))
    if t.dim() > 0 and t.numel() > 0 and t.stride(-1) != 1:
        raise RuntimeError('contrad_hip: %s must be contiguous in its last dimension' % name)


def _ld(t):
    """Pixel stride of an NHWC (N,H,W,C) view / row stride of a (M,C) matrix."""
    if t.dim() == 4:
        N, H, W, C = t.shape
        ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else (t.stride(0) if N > 1 else C))
        if (W > 1 and t.stride(1) != W * ld) or (H * W > 1 and N > 1 and t.stride(0) != H * W * ld):
            raise RuntimeError('contrad_hip: NHWC view must be dense over (N,H,W)')
        return ld
    if t.dim() == 2:
        return t.stride(0) if t.size(0) > 1 else max(t.size(1), t.stride(0))
    raise RuntimeError('contrad_hip: expected a 2-D or 4-D tensor')


def round_up(x, m):
    return (x + m - 1) // m * m


def out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


# Optional per-launch profiler (bench.py): when set to a list, conv-engine launches are bracketed by events on the
# launch stream and appended as (kernel_name, algorithmic_flops, start_event, end_event, layer shape, workgroups of the main launch).  kernel_name is the device
# kernel's name as rocprofv3 prints it.  PROFILE_ONLY (a kernel name) restricts the bracketing to that kernel: every
# event pair costs ~10 us of stream time, so the timed region of bench.py instruments the dominant kernel only.
# (A WGRAD bracket spans the split-K main kernel and its wgrad_reduce_kernel: one C-ABI call launches both.)
PROFILE = None
PROFILE_ONLY = None
# Optional launch log without events (legal inside a stream capture): when set to a list, every conv-engine call appends
# (kernel_name, shape, workgroups, algorithmic GFLOP, executed share); engine.Graphed*Step appends the marker 'capture'
# right before it captures, so bench.py --shape-table can record the launch ORDER of the captured step (a replay keeps the
# order of its capture, which need not be the order of the eager step the event-bracketed table comes from).
SEQUENCE = None


def conv_kernel_name(mode, d):
    bm, bn = ctypes.c_int(0), ctypes.c_int(0)
    lib().call('contrad_conv2d_tile', ctypes.byref(d), mode, ctypes.byref(bm), ctypes.byref(bn))
    path = lib().raw('contrad_conv2d_path')(ctypes.byref(d), mode)
    if path == 4:
        return 'wgrad_c32_kernel'
    if path == 5:
        return 'fwd_k1_kernel'
    if path == 6:
        return 'conv_c32_kernel<%d>' % mode
    if path == 7:
        return 'wino_wgrad_kernel' if mode == 2 else 'wino_kernel<%d>' % mode
    if path == 8:
        return 'wino22_wgrad_kernel' if mode == 2 else 'wino22_kernel<%d>' % mode
    if path == 9:
        return 'wino44_kernel<%d>' % mode
    if path == 11:
        return 'wino44n_kernel<%d>' % mode
    if path == 10:
        return 'wino23_kernel'
    if path in (2, 3):
        return 'igemm_lean_kernel<%d, %d, %d>' % (mode, bm.value, bn.value)
    return 'igemm_kernel<%d, %d, %d, %s>' % (mode, bm.value, bn.value, 'true' if path == 1 else 'false')


def _conv_call(mode, d, name, *args):
    if SEQUENCE is not None:
        SEQUENCE.append([conv_kernel_name(mode, d), [d.N, d.H, d.W, d.C, d.K, d.KH, d.KW, d.stride, d.pad],
                         int(lib().raw('contrad_conv2d_grid_blocks')(ctypes.byref(d), mode, 1)),
                         round(2.0 * d.N * d.Ho * d.Wo * d.K * d.C * d.KH * d.KW / 1e9, 4),
                         round(float(lib().raw('contrad_conv2d_executed_fraction')(ctypes.byref(d), mode)), 4)])
    if PROFILE is None:
        lib().call(name, *args)
        return
    kname = conv_kernel_name(mode, d)
    if PROFILE_ONLY is not None and kname != PROFILE_ONLY:
        lib().call(name, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    e0.record(st)
    lib().call(name, *args)
    e1.record(st)
    # algorithmic flops; for a strided dgrad this equals the forward count (only contributing taps are multiplied)
    flops = 2.0 * d.N * d.Ho * d.Wo * d.K * d.C * d.KH * d.KW
    shape = (d.N, d.H, d.W, d.C, d.K, d.KH, d.KW, d.stride, d.pad)
    blocks = lib().raw('contrad_conv2d_grid_blocks')(ctypes.byref(d), mode, 1)
    # share of those flops the kernel issues (pixel-major tiles skip the tap-positions that read padding)
    executed = float(lib().raw('contrad_conv2d_executed_fraction')(ctypes.byref(d), mode))
    PROFILE.append((kname, flops, e0, e1, shape, int(blocks), executed))


def make_desc(N, H, W, C, K, KH, KW, stride, pad, ldx, ldy, ldw):
    return ConvDesc(N, H, W, C, ldx, out_size(H, KH, stride, pad), out_size(W, KW, stride, pad), K, ldy,
                    KH, KW, stride, pad, ldw)


def pack_weight(w):
    """(K, C, KH, KW) -> packed GEMM layout [(kh*KW+kw)*C + c][ldw], ldw = round_up(K, 4).  (Host helper for
    tests / one-off packing; the training path packs on device inside the spectral-norm weight prep.)"""
    K, C, KH, KW = w.shape
    ldw = round_up(K, 4)
    wp = w.new_zeros(KH * KW * C, ldw)
    wp[:, :K] = w.permute(2, 3, 1, 0).reshape(KH * KW * C, K)
    return wp


def unpack_weight(wp, K, C, KH, KW):
    return wp[:, :K].reshape(KH, KW, C, K).permute(3, 2, 0, 1).contiguous()


# --------------------------------------------------------------------------------------------------
# convolution engine
# --------------------------------------------------------------------------------------------------
def conv2d_fwd(x, wp, bias, K, KH, KW, stride, pad, slope=1.0, gain=1.0, out=None, addend=None):
    """x: (N,H,W,C) NHWC; wp packed (KH*KW*C, ldw); returns y (N,Ho,Wo,K) = gain * lrelu(conv + bias) [+ addend]."""
    _chk(x, 'x'); _chk(wp, 'wp'); _chk(bias, 'bias'); _chk(addend, 'addend')
    N, H, W, C = x.shape
    Ho, Wo = out_size(H, KH, stride, pad), out_size(W, KW, stride, pad)
    if out is None:
        out = torch.empty((N, Ho, Wo, K), device=x.device, dtype=torch.float32)
    _chk(out, 'out')
    d = make_desc(N, H, W, C, K, KH, KW, stride, pad, _ld(x), _ld(out), wp.stride(0))
    nbytes = lib().raw('contrad_conv2d_fwd_workspace_bytes')(ctypes.byref(d))
    ws = _workspace(nbytes, x.device) if nbytes > 0 else None
    if addend is not None and (tuple(addend.shape) != tuple(out.shape) or _ld(addend) != _ld(out)):
        raise RuntimeError('contrad_hip: addend must match y in shape and leading dimension')
    _conv_call(0, d, 'contrad_conv2d_fwd_add', ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(addend), _p(out),
               float(slope), float(gain), _p(ws), nbytes, _stream())
    return out


def conv2d_dgrad(gy, wp, x_shape, KH, KW, stride, pad, act_ref=None, slope=1.0, gain=1.0, out=None):
    """gy: (N,Ho,Wo,K); returns dx (N,H,W,C) for x_shape=(N,H,W,C); optional fused act' of the producer."""
    _chk(gy, 'gy'); _chk(wp, 'wp'); _chk(act_ref, 'act_ref')
    N, H, W, C = x_shape
    K = gy.shape[3]
    if out is None:
        out = torch.empty((N, H, W, C), device=gy.device, dtype=torch.float32)
    _chk(out, 'out')
    if act_ref is not None and (tuple(act_ref.shape) != tuple(out.shape) or _ld(act_ref) != _ld(out)):
        raise RuntimeError('contrad_hip: act_ref must match dx in shape and leading dimension')
    d = make_desc(N, H, W, C, K, KH, KW, stride, pad, _ld(out), _ld(gy), wp.stride(0))
    if (d.Ho, d.Wo) != (gy.shape[1], gy.shape[2]):
        raise RuntimeError('contrad_hip: gy spatial size does not match the conv geometry')
    nbytes = lib().raw('contrad_conv2d_dgrad_workspace_bytes')(ctypes.byref(d))
    ws = _workspace(nbytes, gy.device) if nbytes > 0 else None
    _conv_call(1, d, 'contrad_conv2d_dgrad_ws', ctypes.byref(d), _p(gy), _p(wp), _p(out), _p(act_ref),
               float(slope), float(gain), _p(ws), nbytes, _stream())
    return out


def conv2d_wino(mode, inp, wp, C, K, bias=None, ref=None, slope=1.0, gain=1.0, out=None, k4s2=False, f44=False, k3s2=False):
    """Winograd kernels on ANY shape ``contrad_conv2d_wino_ok`` accepts (conv2d_fwd / conv2d_dgrad pick them by themselves for
    launches that fill the chip).  3x3 stride 1 pad 1 (F(2x2,3x3), csrc/wino.h; ``f44``: F(4x4,3x3), csrc/wino44.h, maps of
    16x16 and larger) -- mode 0: inp = x (N,H,W,C) -> y (N,H,W,K);
    mode 1: inp = gy (N,H,W,K) -> dx (N,H,W,C) -- or, ``k4s2``, 4x4 stride 2 pad 1 (F(2x2,2x2) on the four phases,
    csrc/wino22.h): mode 0: x (N,H,W,C) -> y (N,H/2,W/2,K); mode 1: gy (N,Ho,Wo,K) -> dx (N,2Ho,2Wo,C) -- or, ``k3s2``, 3x3 stride 2
    pad 0 on an odd map (F(2x2,2x2) on the phases, zero planes skipped, csrc/wino23.h): mode 0 only, x (N,2G+1,2G+1,C) -> y (N,G,G,K)."""
    _chk(inp, 'inp'); _chk(wp, 'wp'); _chk(bias, 'bias'); _chk(ref, 'ref')
    N, Hi, Wi, _ = inp.shape
    co = K if mode == 0 else C
    if k3s2:
        if mode != 0:
            raise RuntimeError('contrad_hip: the 3x3 stride-2 Winograd kernel is forward only')
        H, W = Hi, Wi
        oshape = (N, (H - 1) // 2, (W - 1) // 2, co)
        geo = (3, 3, 2, 0)
    elif k4s2:
        H, W = (Hi, Wi) if mode == 0 else (2 * Hi, 2 * Wi)          # the conv's input map
        oshape = (N, H // 2, W // 2, co) if mode == 0 else (N, H, W, co)
        geo = (4, 4, 2, 1)
    else:
        H, W = Hi, Wi
        oshape = (N, H, W, co)
        geo = (3, 3, 1, 1)
    if out is None:
        out = torch.empty(oshape, device=inp.device, dtype=torch.float32)
    _chk(out, 'out')
    if ref is not None and (tuple(ref.shape) != tuple(out.shape) or _ld(ref) != _ld(out)):
        raise RuntimeError('contrad_hip: ref must match the output in shape and leading dimension')
    d = make_desc(N, H, W, C, K, geo[0], geo[1], geo[2], geo[3], _ld(inp) if mode == 0 else _ld(out),
                  _ld(out) if mode == 0 else _ld(inp), wp.stride(0))
    if f44:
        if k4s2 or lib().raw('contrad_conv2d_wino44_ok')(ctypes.byref(d), mode) != 1:
            raise RuntimeError('contrad_hip: shape not supported by the F(4x4, 3x3) Winograd kernel')
        nbytes = lib().raw('contrad_conv2d_wino44_workspace_bytes')(ctypes.byref(d))
    else:
        if lib().raw('contrad_conv2d_wino_ok')(ctypes.byref(d), mode) != 1:
            raise RuntimeError('contrad_hip: shape not supported by the Winograd kernels')
        nbytes = lib().raw('contrad_conv2d_wino_workspace_bytes')(ctypes.byref(d), mode)
    ws = _workspace(nbytes, inp.device)
    lib().call('contrad_conv2d_wino44' if f44 else 'contrad_conv2d_wino', ctypes.byref(d), mode, _p(inp), _p(wp), _p(bias), _p(ref), _p(out),
               float(slope), float(gain), _p(ws), ctypes.c_longlong(nbytes), _stream())
    return out


def conv2d_wino_wgrad(x, gy, ldw=None, out=None, dbias=None):
    """Weight gradient of a 3x3 stride-1 pad-1 layer (gy as large as x: F(3x3, 2x2), csrc/wino.h) or of a 4x4 stride-2 pad-1
    layer (gy half as large: F(2x2, 2x2) per input phase, csrc/wino22.h) on the Winograd kernels (forced; conv2d_wgrad picks
    them by itself when the launch fills the chip).  Returns dwp packed (KH * KW * C, ldw)."""
    _chk(x, 'x'); _chk(gy, 'gy')
    N, H, W, C = x.shape
    K = gy.shape[3]
    k, st = (3, 1) if gy.shape[1] == H else (4, 2)
    if out is None:
        out = torch.zeros((k * k * C, ldw or round_up(K, 4)), device=x.device, dtype=torch.float32)
    _chk(out, 'out')
    d = make_desc(N, H, W, C, K, k, k, st, 1, _ld(x), _ld(gy), out.stride(0))
    if lib().raw('contrad_conv2d_wino_ok')(ctypes.byref(d), 2) != 1:
        raise RuntimeError('contrad_hip: shape not supported by the Winograd weight-gradient kernel')
    nbytes = lib().raw('contrad_conv2d_wino_workspace_bytes')(ctypes.byref(d), 2)
    ws = _workspace(nbytes, x.device)
    lib().call('contrad_conv2d_wino_wgrad', ctypes.byref(d), _p(x), _p(gy), _p(out), _p(dbias), _p(ws),
               ctypes.c_longlong(ws.numel() * 4), _stream())
    return out


_ws_cache = {}


class private_workspace(object):
    """Scratch buffers allocated inside this scope belong to ``owner_dict`` (held by the caller) instead of the
    process-wide cache.  A captured hipGraph bakes raw scratch pointers into its nodes, and every ``torch.cuda.graph``
    capture runs on the same capture stream: without this, a second captured graph would reuse -- or, when it needs
    more, REPLACE and thereby free -- the buffer the first graph still writes to.  Each Graphed*Step wraps its capture
    in one of these and keeps ``owner_dict`` alive as long as its graph."""

    def __init__(self, owner_dict):
        self.mine = owner_dict

    def __enter__(self):
        global _ws_cache
        self.saved, _ws_cache = _ws_cache, self.mine
        return self

    def __exit__(self, *exc):
        global _ws_cache
        _ws_cache = self.saved
        return False


def _workspace(nbytes, device):
    """Grow-only per-device scratch (reused across calls on the same stream; torch owns the memory)."""
    key = (device.index, _stream_handle())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((max(nbytes, 1) + 3) // 4, device=device, dtype=torch.float32)
        _ws_cache[key] = buf
    return buf


def conv2d_wgrad(x, gy, KH, KW, stride, pad, ldw=None, out=None, dbias=None):
    """Returns dwp packed (KH*KW*C, ldw); optionally also writes the bias gradient sum(gy) into ``dbias`` (K,)."""
    _chk(x, 'x'); _chk(gy, 'gy')
    N, H, W, C = x.shape
    K = gy.shape[3]
    if out is None:
        ldw = ldw or round_up(K, 4)
        out = torch.zeros((KH * KW * C, ldw), device=x.device, dtype=torch.float32)
    _chk(out, 'out')
    d = make_desc(N, H, W, C, K, KH, KW, stride, pad, _ld(x), _ld(gy), out.stride(0))
    nbytes = lib().raw('contrad_conv2d_wgrad_workspace_bytes')(ctypes.byref(d))
    if nbytes < 0:
        raise RuntimeError('contrad_hip: bad conv descriptor (%d)' % nbytes)
    ws = _workspace(nbytes, x.device)
    _conv_call(2, d, 'contrad_conv2d_wgrad', ctypes.byref(d), _p(x), _p(gy), _p(out), _p(dbias), _p(ws),
               ctypes.c_longlong(ws.numel() * 4), _stream())
    return out


# --------------------------------------------------------------------------------------------------
# contrastive losses
# --------------------------------------------------------------------------------------------------
def l2norm_fwd(u):
    _chk(u, 'u')
    R, D = u.shape
    z = torch.empty((R, D), device=u.device, dtype=torch.float32)
    inv = torch.empty((R,), device=u.device, dtype=torch.float32)
    lib().call('contrad_l2norm_fwd', _p(u), _ld(u), _p(z), _p(inv), R, D, 1e-12, _stream())
    return z, inv


def l2norm_bwd(dz, z, inv, out=None, accumulate=False):
    R, D = z.shape
    if out is None:
        out = torch.empty((R, D), device=z.device, dtype=torch.float32)
    lib().call('contrad_l2norm_bwd', _p(dz), _p(z), _p(inv), _p(out), _ld(out), R, D, int(accumulate), _stream())
    return out


def contrast_fwd(z, N, mode, temperature):
    """z (R,D) normalised rows; returns (loss[1], lse[R])."""
    _chk(z, 'z')
    if not z.is_contiguous():
        raise RuntimeError('contrad_hip: z must be contiguous')
    R, D = z.shape
    lse = torch.empty((R,), device=z.device, dtype=torch.float32)
    rowloss = torch.empty((R,), device=z.device, dtype=torch.float32)
    loss = torch.empty((1,), device=z.device, dtype=torch.float32)
    nbytes = lib().raw('contrad_contrast_workspace_bytes')(R, D)
    ws = _workspace(nbytes, z.device)
    lib().call('contrad_contrast_fwd', _p(z), R, D, N, mode, 1.0 / temperature, _p(lse), _p(rowloss), _p(loss),
               _p(ws), nbytes, _stream())
    return loss, lse


def contrast_bwd(z, lse, N, mode, temperature, grad_scale=None):
    R, D = z.shape
    dz = torch.empty((R, D), device=z.device, dtype=torch.float32)
    nbytes = lib().raw('contrad_contrast_workspace_bytes')(R, D)
    ws = _workspace(nbytes, z.device)
    lib().call('contrad_contrast_bwd', _p(z), _p(lse), R, D, N, mode, 1.0 / temperature, _p(grad_scale), _p(dz),
               _p(ws), nbytes, _stream())
    return dz


# --------------------------------------------------------------------------------------------------
# weight preparation (spectral norm / fixed scale + packing), batched
# --------------------------------------------------------------------------------------------------
from ._lib import SnBatch, SnLayer, AdamBatch, SN_MAX_LAYERS, ADAM_MAX_TENSORS, AUG_NPARAM  # noqa: E402


class SnSpec(object):
    """Static description of one layer for the batched weight prep: weight (K, C, KH, KW) [or (K, C) for a
    linear], optional spectral-norm buffers, destination view inside a packed buffer."""
    __slots__ = ('w', 'u', 'v', 'K', 'C', 'T', 'fixed_scale')

    def __init__(self, w, u=None, v=None, fixed_scale=0.0, view_kct=None):
        self.w, self.u, self.v = w, u, v
        if view_kct is not None:
            self.K, self.C, self.T = view_kct
        elif w.dim() == 4:
            self.K, self.C, self.T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
        else:
            self.K, self.C, self.T = w.shape[0], w.shape[1], 1
        assert self.K * self.C * self.T == w.numel()
        self.fixed_scale = float(fixed_scale)


def sn_scratch_floats(specs):
    offs, tot = [], 0
    for s in specs:
        offs.append(tot)
        tot += int(lib().raw('contrad_sn_scratch_floats')(s.K, s.C, s.T))
        tot = round_up(tot, 4)
    return offs, tot


def _sn_batches(specs, wps, ldws, offs, u_snaps=None, v_snaps=None, gwps=None, gws=None):
    """Yield (SnBatch, first_layer_index) chunks of at most SN_MAX_LAYERS layers."""
    for start in range(0, len(specs), SN_MAX_LAYERS):
        b = SnBatch()
        chunk = specs[start:start + SN_MAX_LAYERS]
        b.n = len(chunk)
        for j, s in enumerate(chunk):
            i = start + j
            L = b.layers[j]
            L.w = s.w.data_ptr()
            L.u = s.u.data_ptr() if s.u is not None else None
            L.v = s.v.data_ptr() if s.v is not None else None
            L.u_snap = u_snaps[i].data_ptr() if (u_snaps is not None and u_snaps[i] is not None) else None
            L.v_snap = v_snaps[i].data_ptr() if (v_snaps is not None and v_snaps[i] is not None) else None
            L.wp = wps[i].data_ptr()
            L.gwp = gwps[i].data_ptr() if gwps is not None else None
            L.gw = gws[i].data_ptr() if gws is not None else None
            L.K, L.C, L.T, L.ldw = s.K, s.C, s.T, ldws[i]
            L.fixed_scale = s.fixed_scale
            b.scratch_off[j] = offs[i]
        yield b, start


def sn_weight_prep(specs, wps, ldws, training, scratch, offs, sigma, u_snaps=None, v_snaps=None, eps=1e-12):
    for b, start in _sn_batches(specs, wps, ldws, offs, u_snaps, v_snaps):
        lib().call('contrad_sn_weight_prep', ctypes.byref(b), int(bool(training)), float(eps), _p(scratch),
                   ctypes.c_void_p(sigma.data_ptr() + 4 * start), _stream())


def sn_weight_grad(specs, wps, ldws, gwps, gws, scratch, offs, sigma, u_snaps=None, v_snaps=None):
    for b, start in _sn_batches(specs, wps, ldws, offs, u_snaps, v_snaps, gwps, gws):
        lib().call('contrad_sn_weight_grad', ctypes.byref(b), _p(scratch),
                   ctypes.c_void_p(sigma.data_ptr() + 4 * start), _stream())


# --------------------------------------------------------------------------------------------------
# RGB-end convolutions
# --------------------------------------------------------------------------------------------------
def rgb_conv_fwd(img, wp, bias, K, k, in_scale, in_shift, slope, gain, out=None):
    """img NCHW (N,3,H,W) -> NHWC (N,H,W,K)."""
    _chk(img, 'img'); _chk(wp, 'wp'); _chk(bias, 'bias')
    if not img.is_contiguous():
        raise RuntimeError('contrad_hip: image batch must be contiguous NCHW')
    N, Cin, H, W = img.shape
    if out is None:
        out = torch.empty((N, H, W, K), device=img.device, dtype=torch.float32)
    lib().call('contrad_rgb_conv_fwd', _p(img), _p(wp), _p(bias), _p(out), N, Cin, H, W, K, k, _ld(out),
               wp.stride(0), float(in_scale), float(in_shift), float(slope), float(gain), _stream())
    return out


def rgb_conv_wgrad(img, gy, k, in_scale, in_shift, dwp, dbias=None):
    _chk(img, 'img'); _chk(gy, 'gy'); _chk(dwp, 'dwp')
    N, Cin, H, W = img.shape
    K = gy.shape[3]
    nbytes = lib().raw('contrad_rgb_conv_wgrad_workspace_bytes')(N, Cin, H, W, K, k)
    ws = _workspace(nbytes, img.device)
    lib().call('contrad_rgb_conv_wgrad', _p(img), _p(gy), _p(dwp), _p(dbias), N, Cin, H, W, K, k, _ld(gy),
               dwp.stride(0), float(in_scale), float(in_shift), _p(ws), ctypes.c_longlong(ws.numel() * 4), _stream())
    return dwp


def rgb_conv_dgrad(gy, wp, bias, C, k, act=0, out_scale=1.0, out_shift=0.0, out=None, mod=None, residual=None):
    """gy NHWC (N,H,W,K) -> NCHW (N,C,H,W), C <= 4; optional per-sample channel modulation (N,K) and NCHW residual."""
    _chk(gy, 'gy'); _chk(wp, 'wp'); _chk(bias, 'bias'); _chk(mod, 'mod'); _chk(residual, 'residual')
    N, H, W, K = gy.shape
    if out is None:
        out = torch.empty((N, C, H, W), device=gy.device, dtype=torch.float32)
    if mod is not None and (tuple(mod.shape) != (N, K) or not mod.is_contiguous()):
        raise RuntimeError('contrad_hip: mod must be a contiguous (N, K) tensor')
    if residual is not None and (tuple(residual.shape) != (N, C, H, W) or not residual.is_contiguous()):
        raise RuntimeError('contrad_hip: residual must be a contiguous (N, C, H, W) tensor')
    lib().call('contrad_rgb_conv_dgrad', _p(gy), _p(wp), _p(bias), _p(mod), _p(residual), _p(out), N, C, H, W, K, k,
               _ld(gy), wp.stride(0), int(act), float(out_scale), float(out_shift), _stream())
    return out


# --------------------------------------------------------------------------------------------------
# statistics / BatchNorm / losses / Adam
# --------------------------------------------------------------------------------------------------
def colstats(x2d, with_sq=False, out=None, accumulate=False):
    """x2d: (M, K) row-major view (row stride = ld).  Returns (1 or 2, K)."""
    _chk(x2d, 'x')
    M, K = x2d.shape
    ld = _ld(x2d)
    if out is None:
        out = torch.empty((2 if with_sq else 1, K), device=x2d.device, dtype=torch.float32)
    nbytes = lib().raw('contrad_colstats_workspace_bytes')(ctypes.c_longlong(M), K, int(with_sq))
    ws = _workspace(nbytes, x2d.device)
    lib().call('contrad_colstats', _p(x2d), ctypes.c_longlong(M), K, ld, int(with_sq), _p(out), int(accumulate),
               _p(ws), ctypes.c_longlong(ws.numel() * 4), _stream())
    return out


def as_rows(t):
    """(N,H,W,C) NHWC dense tensor / view -> (N*H*W, C) strided matrix view (no copy)."""
    N, H, W, C = t.shape
    return t.as_strided((N * H * W, C), (_ld(t), 1), t.storage_offset())


def bn_relu_apply(x2d, y2d, stats, count, gamma, beta, eps=1e-5, perm_hw=1):
    M, K = x2d.shape
    lib().call('contrad_bn_relu_apply', _p(x2d), _p(y2d), ctypes.c_longlong(M), K, _ld(x2d), _ld(y2d), _p(stats),
               float(count), _p(gamma), _p(beta), float(eps), int(perm_hw), _stream())
    return y2d


def bn_running_update(stats, count, conv_bias, momentum, running_mean, running_var, num_batches_tracked=None):
    K = running_mean.numel()
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise RuntimeError('contrad_hip: num_batches_tracked must be an int64 tensor on the device')
    lib().call('contrad_bn_running_update', _p(stats), float(count), K, _p(conv_bias), float(momentum),
               _p(running_mean), _p(running_var), _p(num_batches_tracked), _stream())


GAN_LOSS_KINDS = {'nonsat': 0, 'wgan': 1, 'hinge': 2, 'lsgan': 3}


def gan_d_loss(logits, N, kind):
    """logits (3N,1) -> (out3 = [loss, mean d_real, mean d_gen], grad (3N,1))."""
    _chk(logits, 'logits')
    out = torch.empty(3, device=logits.device, dtype=torch.float32)
    grad = torch.empty((3 * N, 1), device=logits.device, dtype=torch.float32)
    lib().call('contrad_gan_d_loss', _p(logits), logits.stride(0), N, GAN_LOSS_KINDS[kind], _p(out), _p(grad), _stream())
    return out, grad


def gan_g_loss(logits, kind):
    _chk(logits, 'logits')
    N = logits.shape[0]
    out = torch.empty(1, device=logits.device, dtype=torch.float32)
    grad = torch.empty((N, 1), device=logits.device, dtype=torch.float32)
    lib().call('contrad_gan_g_loss', _p(logits), logits.stride(0), N, GAN_LOSS_KINDS.get(kind, 1), _p(out), _p(grad),
               _stream())
    return out, grad


def adam_step(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1, beta2, eps=1e-8, grad_scale=1.0):
    n = len(params)
    for start in range(0, n, ADAM_MAX_TENSORS):
        b = AdamBatch()
        m = min(ADAM_MAX_TENSORS, n - start)
        b.n = m
        for j in range(m):
            p, g = params[start + j], grads[start + j]
            if not (p.is_contiguous() and g.is_contiguous()):
                raise RuntimeError('contrad_hip: Adam needs contiguous parameters and gradients')
            t = b.t[j]
            t.p, t.g = p.data_ptr(), g.data_ptr()
            t.m, t.v = exp_avgs[start + j].data_ptr(), exp_avg_sqs[start + j].data_ptr()
            t.numel = p.numel()
        lib().call('contrad_adam_step', ctypes.byref(b), int(step), float(lr), float(beta1), float(beta2),
                   float(eps), float(grad_scale), _stream())


def adam_step_dev(params, grads, exp_avgs, exp_avg_sqs, hyper, beta1, beta2, eps=1e-8):
    """adam_step with {lr / bc1, 1 / sqrt(bc2), grad_scale} taken from the device tensor ``hyper`` (3 floats)."""
    n = len(params)
    for start in range(0, n, ADAM_MAX_TENSORS):
        b = AdamBatch()
        m = min(ADAM_MAX_TENSORS, n - start)
        b.n = m
        for j in range(m):
            p, g = params[start + j], grads[start + j]
            if not (p.is_contiguous() and g.is_contiguous()):
                raise RuntimeError('contrad_hip: Adam needs contiguous parameters and gradients')
            t = b.t[j]
            t.p, t.g = p.data_ptr(), g.data_ptr()
            t.m, t.v = exp_avgs[start + j].data_ptr(), exp_avg_sqs[start + j].data_ptr()
            t.numel = p.numel()
        lib().call('contrad_adam_step_dev', ctypes.byref(b), _p(hyper), float(beta1), float(beta2), float(eps), _stream())


def axpby_(y, x, a, b):
    lib().call('contrad_axpby', _p(y), _p(x), ctypes.c_longlong(y.numel()), float(a), float(b), _stream())
    torch.autograd.graph.increment_version(y)          # raw-pointer write: keep ``_version``-keyed caches honest
    return y


# --------------------------------------------------------------------------------------------------
# augmentation
# --------------------------------------------------------------------------------------------------
def _chk_out(out, like, name):
    """A caller-provided output buffer: same shape, fp32, on the device, dense (the kernels write it with flat indices)."""
    _chk(out, 'out')
    if tuple(out.shape) != tuple(like.shape) or not out.is_contiguous() or out.device != like.device:
        raise RuntimeError('contrad_hip: %s: out must be a contiguous tensor of shape %s on %s' % (name, tuple(like.shape), like.device))


def simclr_augment(x, params, contrast_first, has_contrast, out=None):
    """x NCHW (B,3,H,W); params (B, AUG_NPARAM) on the same device."""
    _chk(x, 'x'); _chk(params, 'params')
    if not x.is_contiguous() or tuple(params.shape) != (x.shape[0], AUG_NPARAM) or not params.is_contiguous():
        raise RuntimeError('contrad_hip: simclr_augment needs contiguous NCHW input and (B,%d) params' % AUG_NPARAM)
    B, C, H, W = x.shape
    if C != 3:
        raise RuntimeError('contrad_hip: simclr_augment is defined for RGB images')
    if out is None:
        out = torch.empty_like(x)
    _chk_out(out, x, 'simclr_augment')
    nbytes = lib().raw('contrad_simclr_workspace_bytes')(B, H, W)
    ws = _workspace(nbytes, x.device)
    lib().call('contrad_simclr_augment', _p(x), _p(out), _p(params), B, H, W, int(contrast_first), int(has_contrast),
               _p(ws), ctypes.c_longlong(ws.numel() * 4), _stream())
    return out


def gaussian_blur_masked(x, params, kernel1d, radius, out=None):
    _chk(x, 'x'); _chk(params, 'params'); _chk(kernel1d, 'kernel1d')
    if not x.is_contiguous():
        raise RuntimeError('contrad_hip: gaussian_blur_masked needs a contiguous NCHW input')
    B, C, H, W = x.shape
    tmp = torch.empty_like(x)
    if out is None:
        out = torch.empty_like(x)
    _chk_out(out, x, 'gaussian_blur_masked')
    lib().call('contrad_gaussian_blur_masked', _p(x), _p(tmp), _p(out), _p(params), _p(kernel1d), B, H, W,
               int(radius), _stream())
    return out


# --------------------------------------------------------------------------------------------------
# the reference's native StyleGAN2 ops
# --------------------------------------------------------------------------------------------------
def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0, 0, 0)):
    """x NHWC (B,H,W,C) contiguous, kernel (kh,kw); pad = (x0, x1, y0, y1).  Returns NHWC."""
    _chk(x, 'x'); _chk(kernel, 'kernel')
    if not x.is_contiguous() or not kernel.is_contiguous():
        raise RuntimeError('contrad_hip: upfirdn2d needs contiguous tensors')
    B, H, W, C = x.shape
    kh, kw = kernel.shape
    px0, px1, py0, py1 = pad
    oh = (H * up + py0 + py1 - kh) // down + 1
    ow = (W * up + px0 + px1 - kw) // down + 1
    out = torch.empty((B, oh, ow, C), device=x.device, dtype=torch.float32)
    lib().call('contrad_upfirdn2d', _p(x), _p(kernel), _p(out), B, H, W, C, kh, kw, up, up, down, down,
               px0, px1, py0, py1, _stream())
    return out


def upfirdn2d_fused(x, kernel, up=1, down=1, pad=(0, 0, 0, 0), addend=None, act_ref=None, slope=1.0, gain=1.0,
                    want_out=True, want_out2=False):
    """upfirdn2d with the fused epilogue: v = upfirdn2d(x) [+ addend]; returns (v or None, v * act'(act_ref) or None)."""
    _chk(x, 'x'); _chk(kernel, 'kernel'); _chk(addend, 'addend'); _chk(act_ref, 'act_ref')
    B, H, W, C = x.shape
    kh, kw = kernel.shape
    px0, px1, py0, py1 = pad
    oh = (H * up + py0 + py1 - kh) // down + 1
    ow = (W * up + px0 + px1 - kw) // down + 1
    shape = (B, oh, ow, C)
    for t, nm in ((addend, 'addend'), (act_ref, 'act_ref')):
        if t is not None and (tuple(t.shape) != shape or not t.is_contiguous()):
            raise RuntimeError('contrad_hip: upfirdn2d_fused %s must be a contiguous tensor of the output shape' % nm)
    if not x.is_contiguous() or not kernel.is_contiguous() or (want_out2 and act_ref is None):
        raise RuntimeError('contrad_hip: upfirdn2d_fused needs contiguous tensors (and act_ref for the second output)')
    out = torch.empty(shape, device=x.device, dtype=torch.float32) if want_out else None
    out2 = torch.empty(shape, device=x.device, dtype=torch.float32) if want_out2 else None
    lib().call('contrad_upfirdn2d_fused', _p(x), _p(kernel), _p(out), B, H, W, C, kh, kw, up, up, down, down,
               px0, px1, py0, py1, _p(addend), _p(act_ref), float(slope), float(gain), _p(out2), _stream())
    return out, out2


def fused_bias_act(x, bias, ref, act, grad, alpha, scale, channels_last_size=None):
    """Elementwise on a contiguous tensor whose LAST dim is the channel (NHWC / (M,K))."""
    _chk(x, 'x'); _chk(bias, 'bias'); _chk(ref, 'ref')
    if not x.is_contiguous() or (ref is not None and not ref.is_contiguous()):
        raise RuntimeError('contrad_hip: fused_bias_act needs contiguous tensors')
    y = torch.empty_like(x)
    C = x.shape[-1]
    lib().call('contrad_fused_bias_act', _p(x), _p(bias), _p(ref), _p(y), ctypes.c_longlong(x.numel()), 1, C,
               int(act), int(grad), float(alpha), float(scale), _stream())
    return y


def lincomb(x, z, a, b):
    if not (x.is_contiguous() and z.is_contiguous()) or x.shape != z.shape:
        raise RuntimeError('contrad_hip: lincomb needs equal-shape contiguous tensors')
    y = torch.empty_like(x)
    lib().call('contrad_lincomb', _p(x), _p(z), _p(y), ctypes.c_longlong(x.numel()), float(a), float(b), _stream())
    return y


def minibatch_stddev(mode, x, gy=None, h=None, cpad=None, splits=None):
    """contrad_minibatch_stddev (stylegan2/discriminator.py:22-33 and its two backward passes) on NHWC x (B,H,W,C).
    mode 0 -> y (B,H,W,Cp); mode 1 (gy (B,H,W,Cp)) -> gx (B,H,W,C); mode 2 (gy, h (B,H,W,C)) -> (gx2, ggy).
    ``splits``: sizes of consecutive sub-batches whose statistics stay separate (several discriminator calls merged into
    one pass, ResidualDiscriminatorP.call_batches): one launch per sub-batch into slices of the same output."""
    _chk(x, 'x')
    B, H, W, C = x.shape
    if not x.is_contiguous():
        raise RuntimeError('contrad_hip: minibatch_stddev needs a dense NHWC input')
    Cp = cpad if gy is None else gy.shape[3]
    out2 = None
    if mode == 0:
        out = torch.empty((B, H, W, Cp), device=x.device, dtype=torch.float32)
    else:
        if not gy.is_contiguous() or tuple(gy.shape[:3]) != (B, H, W):
            raise RuntimeError('contrad_hip: minibatch_stddev needs a dense gy of the output shape')
        out = torch.empty_like(x)
        if mode == 2:
            if not h.is_contiguous() or h.shape != x.shape:
                raise RuntimeError('contrad_hip: minibatch_stddev needs a dense h of the input shape')
            out2 = torch.empty_like(gy)
    splits = list(splits) if splits else [B]
    if sum(splits) != B:
        raise RuntimeError('contrad_hip: minibatch_stddev: splits must sum to the batch size')
    b0 = 0
    for n in splits:
        sl = slice(b0, b0 + n)
        lib().call('contrad_minibatch_stddev', int(mode), _p(x[sl]), _p(gy[sl]) if gy is not None else None,
                   _p(h[sl]) if h is not None else None, _p(out[sl]), _p(out2[sl]) if out2 is not None else None,
                   n, H * W, C, Cp, _stream())
        b0 += n
    return out if mode != 2 else (out, out2)


def sumsq(x, scale):
    """scale * sum(x^2) as a 0-dim tensor, fixed summation order (contrad_sumsq)."""
    _chk(x, 'x')
    x = x if x.is_contiguous() else x.contiguous()
    n = x.numel()
    out = torch.empty((), device=x.device, dtype=torch.float32)
    nbytes = lib().raw('contrad_sumsq_workspace_bytes')(ctypes.c_longlong(n))
    ws = _workspace(nbytes, x.device)
    lib().call('contrad_sumsq', _p(x), ctypes.c_longlong(n), float(scale), _p(out), _p(ws), ctypes.c_longlong(ws.numel() * 4),
               _stream())
    return out


def scale_dev(x, s, c):
    """x * (c * s) with s a 0-dim device tensor (no host read)."""
    _chk(x, 'x'); _chk(s, 's')
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(x)
    lib().call('contrad_scale_dev', _p(x), _p(s), float(c), _p(y), ctypes.c_longlong(x.numel()), _stream())
    return y


def pixelnorm(x):
    _chk(x, 'x')
    x = x.contiguous()
    y = torch.empty_like(x)
    lib().call('contrad_pixelnorm', _p(x), _p(y), x.shape[0], x.shape[1], _stream())
    return y


def nhwc_scale(x, s):
    """x (N,H,W,C) contiguous, s (N,C) -> x * s[:, None, None, :]."""
    _chk(x, 'x'); _chk(s, 's')
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    lib().call('contrad_nhwc_scale', _p(x), _p(s.contiguous()), _p(y), N, ctypes.c_longlong(H * W), C, _stream())
    return y


def modconv_tables(layers):
    """``layers``: list of (w [Cout][Cin][T] contiguous, wp packed view, wsq [Cin][Cout] or None, transposed, scale).
    One launch (per 32 layers) writes every packed shared weight and demodulation table of the generator."""
    from ._lib import ModconvBatch, MODCONV_MAX_LAYERS
    for start in range(0, len(layers), MODCONV_MAX_LAYERS):
        chunk = layers[start:start + MODCONV_MAX_LAYERS]
        b = ModconvBatch()
        b.n = len(chunk)
        for j, (w, wp, wsq, transposed, scale) in enumerate(chunk):
            _chk(w, 'w'); _chk(wp, 'wp'); _chk(wsq, 'wsq')
            if not w.is_contiguous() or wp.stride(1) != 1 or (wsq is not None and not wsq.is_contiguous()):
                raise RuntimeError('contrad_hip: modconv_tables needs a contiguous weight / table and unit-stride packed columns')
            L = b.layers[j]
            L.w, L.wp, L.wsq = w.data_ptr(), wp.data_ptr(), (wsq.data_ptr() if wsq is not None else None)
            L.Cout, L.Cin, L.T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            L.ldw, L.transposed, L.scale = wp.stride(0), int(bool(transposed)), float(scale)
        lib().call('contrad_modconv_tables', ctypes.byref(b), _stream())


def modconv_demod(layers, eps=1e-8):
    """``layers``: list of (style [B][Cin] with contiguous rows, wsq [Cin][K], out [B][K]); one launch writes every
    out = rsqrt(style^2 @ wsq + eps)."""
    from ._lib import DemodBatch, MODCONV_MAX_LAYERS
    for start in range(0, len(layers), MODCONV_MAX_LAYERS):
        chunk = layers[start:start + MODCONV_MAX_LAYERS]
        b = DemodBatch()
        b.n, b.B = len(chunk), chunk[0][0].shape[0]
        for j, (st, wsq, out) in enumerate(chunk):
            _chk(st, 'style'); _chk(wsq, 'wsq'); _chk(out, 'out')
            if (st.shape[0] != b.B or not st.is_contiguous() or not wsq.is_contiguous() or not out.is_contiguous()
                    or st.shape[1] != wsq.shape[0] or tuple(out.shape) != (b.B, wsq.shape[1])):
                raise RuntimeError('contrad_hip: modconv_demod needs contiguous [B][Cin] / [Cin][K] / [B][K] tensors')
            L = b.layers[j]
            L.style, L.wsq, L.out = st.data_ptr(), wsq.data_ptr(), out.data_ptr()
            L.Cin, L.K = wsq.shape[0], wsq.shape[1]
        lib().call('contrad_modconv_demod', ctypes.byref(b), float(eps), _stream())


def modconv_epilogue_(x, demod, noise, noise_w, bias, out=None, post_scale=None):
    """sqrt2 * lrelu_0.2(x * demod + noise_w * noise + bias) [* post_scale[n,k]] on x (N,H,W,K); in place unless ``out``
    is given.  ``post_scale``: the consuming layer's style vector (its nhwc_scale pass folded into this store)."""
    N, H, W, K = x.shape
    out = x if out is None else out
    lib().call('contrad_modconv_epilogue', _p(x), _p(demod), _p(noise), _p(noise_w), _p(bias), _p(post_scale), _p(out), N,
               ctypes.c_longlong(H * W), K, _stream())
    return out


def upfirdn2d_modconv(x, kernel, pad, demod, noise, noise_w, bias, post_scale=None):
    """Blur (4x4 FIR, up = down = 1, pad = (x0, x1, y0, y1)) followed by modconv_epilogue_ in ONE launch: the tail of the
    generator's upsampling StyledConv (generator.py:80-83,97-118).  x (N,H,W,K) NHWC -> (N,H',W',K)."""
    _chk(x, 'x'); _chk(kernel, 'kernel')
    if tuple(kernel.shape) != (4, 4) or not x.is_contiguous():
        raise RuntimeError('contrad_hip: upfirdn2d_modconv needs a dense NHWC input and the 4x4 FIR')
    N, H, W, K = x.shape
    oh, ow = H + pad[2] + pad[3] - 4 + 1, W + pad[0] + pad[1] - 4 + 1
    # the kernel reads demod / post_scale as float4 at [N][K], bias at [K], noise at [N][oh][ow]: wrong sizes read out of bounds
    for t, nm, shape in ((demod, 'demod', (N, K)), (post_scale, 'post_scale', (N, K)), (bias, 'bias', (K,)),
                         (noise, 'noise', (N, 1, oh, ow))):
        if t is None:
            continue
        _chk(t, nm)
        if t.numel() != int(torch.Size(shape).numel()) or not t.is_contiguous():
            raise RuntimeError('contrad_hip: upfirdn2d_modconv: %s must be a contiguous tensor of %s elements' % (nm, shape))
    if K % 4:
        raise RuntimeError('contrad_hip: upfirdn2d_modconv needs a channel count that is a multiple of 4')
    out = torch.empty(N, oh, ow, K, device=x.device, dtype=torch.float32)
    lib().call('contrad_upfirdn2d_modconv', _p(x), _p(kernel), _p(out), N, H, W, K, int(pad[0]), int(pad[1]), int(pad[2]),
               int(pad[3]), _p(demod), _p(noise), _p(noise_w), _p(bias), _p(post_scale), _stream())
    return out


def simclr_augment_bwd(x, params, grad_out, contrast_first, has_contrast):
    B, C, H, W = x.shape
    gin = torch.empty_like(x)
    nbytes = lib().raw('contrad_simclr_augment_bwd_workspace_bytes')(B, H, W)
    ws = _workspace(nbytes, x.device)
    lib().call('contrad_simclr_augment_bwd', _p(x), _p(params), _p(grad_out.contiguous()), _p(gin), B, H, W,
               int(contrast_first), int(has_contrast), _p(ws), ctypes.c_longlong(ws.numel() * 4), _stream())
    return gin


def cutout_masked_(y, params, length):
    """In place on a contiguous NCHW batch: RandomApply(CutOut(length)); also its own backward."""
    _chk(y, 'y'); _chk(params, 'params')
    if not y.is_contiguous():
        raise RuntimeError('contrad_hip: cutout needs a contiguous NCHW batch')
    B, C, H, W = y.shape
    lib().call('contrad_cutout_masked', _p(y), _p(params), B, H, W, int(length), _stream())
    return y


def gaussian_blur_masked_bwd(grad_out, params, kernel1d, radius):
    B, C, H, W = grad_out.shape
    tmp = torch.empty_like(grad_out)
    gin = torch.empty_like(grad_out)
    lib().call('contrad_gaussian_blur_masked_bwd', _p(grad_out.contiguous()), _p(tmp), _p(gin), _p(params),
               _p(kernel1d), B, H, W, int(radius), _stream())
    return gin


def nhwc_dot(a, b, per_channel=True):
    """a (N,H,W,C) contiguous; b (N,H,W,C) [per_channel] or (N,H,W) / (N,1,H,W) -> (N,C) sums over the pixels."""
    _chk(a, 'a'); _chk(b, 'b')
    N, H, W, C = a.shape
    if not (a.is_contiguous() and b.is_contiguous()) or b.numel() != (a.numel() if per_channel else N * H * W):
        raise RuntimeError('contrad_hip: nhwc_dot needs contiguous (N,H,W,C) / (N,H,W) operands')
    out = torch.empty((N, C), device=a.device, dtype=torch.float32)
    nbytes = lib().raw('contrad_nhwc_dot_workspace_bytes')(N, ctypes.c_longlong(H * W), C)
    ws = _workspace(nbytes, a.device)
    lib().call('contrad_nhwc_dot', _p(a), _p(b), _p(out), N, ctypes.c_longlong(H * W), C, int(per_channel), _p(ws),
               ctypes.c_longlong(ws.numel() * 4), _stream())
    return out


def bn_relu_bwd(dy2d, x2d, stats, count, gamma, beta, eps, reduce_fn=None):
    """Returns (dx2d, dgamma, dbeta).  reduce_fn(out2k) -> global count multiplier hook for SyncBN."""
    M, K = x2d.shape
    ld = _ld(x2d)
    if _ld(dy2d) != ld:
        raise RuntimeError('contrad_hip: dy and x must share the leading dimension')
    out = torch.empty((2, K), device=x2d.device, dtype=torch.float32)
    nbytes = lib().raw('contrad_colstats_workspace_bytes')(ctypes.c_longlong(M), K, 1)
    ws = _workspace(nbytes, x2d.device)
    lib().call('contrad_bn_relu_bwd_stats', _p(dy2d), _p(x2d), ctypes.c_longlong(M), K, ld, _p(stats), float(count),
               _p(gamma), _p(beta), float(eps), _p(out), _p(ws), ctypes.c_longlong(ws.numel() * 4), _stream())
    # SyncBN: dx needs the GLOBAL {sum dy, sum dy*xhat}; dgamma / dbeta stay this rank's LOCAL sums (as
    # torch.nn.SyncBatchNorm returns them) -- the gradient exchange averages them like every other parameter
    local = out
    if reduce_fn is not None:
        local = out.clone()
        reduce_fn(out)
    dx = torch.empty_like(x2d)
    if _ld(dx) != ld:
        raise RuntimeError('contrad_hip: bn backward expects dense rows')
    lib().call('contrad_bn_relu_bwd_apply', _p(dy2d), _p(x2d), _p(dx), ctypes.c_longlong(M), K, ld, _p(stats),
               float(count), _p(gamma), _p(beta), float(eps), _p(out), _stream())
    return dx, local[1], local[0]
