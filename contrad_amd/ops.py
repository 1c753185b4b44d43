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


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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
def conv2d_fwd(x, wp, bias, K, KH, KW, stride, pad, slope=1.0, gain=1.0, out=None):
    """x: (N,H,W,C) NHWC; wp packed (KH*KW*C, ldw); returns y (N,Ho,Wo,K)."""
    _chk(x, 'x'); _chk(wp, 'wp'); _chk(bias, 'bias')
    N, H, W, C = x.shape
    Ho, Wo = out_size(H, KH, stride, pad), out_size(W, KW, stride, pad)
    if out is None:
        out = torch.empty((N, Ho, Wo, K), device=x.device, dtype=torch.float32)
    _chk(out, 'out')
    d = make_desc(N, H, W, C, K, KH, KW, stride, pad, _ld(x), _ld(out), wp.stride(0))
    lib().call('contrad_conv2d_fwd', ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(out),
               float(slope), float(gain), _stream())
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
    lib().call('contrad_conv2d_dgrad', ctypes.byref(d), _p(gy), _p(wp), _p(out), _p(act_ref),
               float(slope), float(gain), _stream())
    return out


_ws_cache = {}


def _workspace(nbytes, device):
    """Grow-only per-device scratch (reused across calls on the same stream; torch owns the memory)."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((max(nbytes, 1) + 3) // 4, device=device, dtype=torch.float32)
        _ws_cache[key] = buf
    return buf


def conv2d_wgrad(x, gy, KH, KW, stride, pad, ldw=None, out=None):
    """Returns dwp packed (KH*KW*C, ldw)."""
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
    lib().call('contrad_conv2d_wgrad', ctypes.byref(d), _p(x), _p(gy), _p(out), _p(ws),
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
    lib().call('contrad_contrast_fwd', _p(z), R, D, N, mode, 1.0 / temperature, _p(lse), _p(rowloss), _p(loss),
               _stream())
    return loss, lse


def contrast_bwd(z, lse, N, mode, temperature, grad_scale=None):
    R, D = z.shape
    dz = torch.empty((R, D), device=z.device, dtype=torch.float32)
    lib().call('contrad_contrast_bwd', _p(z), _p(lse), R, D, N, mode, 1.0 / temperature, _p(grad_scale), _p(dz),
               _stream())
    return dz
