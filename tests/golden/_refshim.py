"""Import shims that let the read-only reference at /root/reference be imported on CPU in the build
container (no gin / kornia / tensorboardX / nvcc there).  Used ONLY by make_golden.py -- nothing in
tests/, bench.py or the product imports this at run time, and nothing here is reference code.

  gin          -> a 40-line stand-in: @configurable (in-place __init__/function wrapping that fills
                  missing kwargs from a binding table), REQUIRED, bind().
  kornia       -> get_gaussian_kernel2d / filter2D restated from kornia's documented contract.
  tensorboardX -> no-op SummaryWriter.
  torch.utils.cpp_extension.load -> returns None (CPU tensors take the reference's *_native paths).
"""
import functools
import inspect
import sys
import types

import torch
import torch.nn.functional as F

REFERENCE_ROOT = '/root/reference'

_BINDINGS = {}


def bind(name, **kwargs):
    _BINDINGS.setdefault(name, {}).update(kwargs)


def _fill(name, fn, args, kwargs):
    sig = inspect.signature(fn)
    try:
        bound = sig.bind_partial(*args, **kwargs)
    except TypeError:
        return kwargs
    for k, v in _BINDINGS.get(name, {}).items():
        if k in sig.parameters and k not in bound.arguments:
            kwargs[k] = v
    return kwargs


def _configurable(name_or_fn=None, module=None, whitelist=None, blacklist=None, **_):
    def deco(obj, name=None):
        name = name or obj.__name__
        if inspect.isclass(obj):
            orig = obj.__init__

            @functools.wraps(orig)
            def __init__(self, *a, **kw):
                kw = _fill(name, orig, (self,) + a, kw)
                orig(self, *a, **kw)
            obj.__init__ = __init__
            return obj

        @functools.wraps(obj)
        def wrapper(*a, **kw):
            kw = _fill(name, obj, a, kw)
            return obj(*a, **kw)
        return wrapper

    if callable(name_or_fn):
        return deco(name_or_fn)
    return lambda obj: deco(obj, name_or_fn)


def _gaussian_kernel1d(ksize, sigma):
    x = torch.arange(ksize, dtype=torch.float32) - ksize // 2
    g = torch.exp(-x.pow(2) / (2 * float(sigma) ** 2))
    return g / g.sum()


def _get_gaussian_kernel2d(kernel_size, sigma):
    ky, kx = kernel_size
    sy, sx = sigma
    return torch.outer(_gaussian_kernel1d(ky, sy), _gaussian_kernel1d(kx, sx))


def _filter2D(inp, kernel, border_type='reflect'):
    b, c, h, w = inp.shape
    kh, kw = kernel.shape[-2:]
    k = kernel.view(1, 1, kh, kw).to(inp).repeat(c, 1, 1, 1)
    xp = F.pad(inp, [kw // 2, kw // 2, kh // 2, kh // 2], mode=border_type)
    return F.conv2d(xp, k, groups=c)


def install():
    gin = types.ModuleType('gin')
    gin.configurable = _configurable
    gin.REQUIRED = object()
    gin.bind = bind
    gin.parse_config_files_and_bindings = lambda *a, **k: None
    sys.modules['gin'] = gin

    kornia = types.ModuleType('kornia')
    filters = types.ModuleType('kornia.filters')
    filters.get_gaussian_kernel2d = _get_gaussian_kernel2d
    filters.filter2D = _filter2D
    kornia.filters = filters
    sys.modules['kornia'] = kornia
    sys.modules['kornia.filters'] = filters

    tbx = types.ModuleType('tensorboardX')
    tbx.SummaryWriter = type('SummaryWriter', (), {'__init__': lambda self, *a, **k: None})
    sys.modules['tensorboardX'] = tbx

    import torch.utils.cpp_extension as cpp_ext
    cpp_ext.load = lambda *a, **k: None

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return gin


def bind_cifar_defaults():
    """configs/defaults/augment.gin values."""
    bind('ColorJitterLayer', brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1)
    bind('RandomResizeCropLayer', scale=(0.2, 1.0))
    bind('GaussianBlur', sigma_range=(0.1, 2.0))


def bind_afhq():
    """configs/gan/stylegan2/afhq_dog_style64.gin overrides."""
    bind('ColorJitterLayer', brightness=0.8, contrast=0.8, saturation=0.8, hue=0.2)
    bind('RandomResizeCropLayer', scale=(0.08, 1.0))
    bind('GaussianBlur', sigma_range=(0.1, 2.0))
