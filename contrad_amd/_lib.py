"""ctypes binding of libcontrad_hip.so.  Prototypes are parsed from include/contrad_hip.h, so the header
is the single source of truth for the C ABI.  No fallback: a missing library is a hard error."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CONTRAD_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libcontrad_hip.so')   # (env: dev A/B builds)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'contrad_hip.h')


class ConvDesc(ctypes.Structure):
    """contrad_conv_desc (include/contrad_hip.h)."""
    _fields_ = [(n, ctypes.c_int) for n in
                ('N', 'H', 'W', 'C', 'ldx', 'Ho', 'Wo', 'K', 'ldy', 'KH', 'KW', 'stride', 'pad', 'ldw')]


SN_MAX_LAYERS = 24
ADAM_MAX_TENSORS = 64
ADAM_CHUNK = 16384
AUG_NPARAM = 16


class SnLayer(ctypes.Structure):
    """contrad_sn_layer."""
    _fields_ = [('w', ctypes.c_void_p), ('u', ctypes.c_void_p), ('v', ctypes.c_void_p),
                ('u_snap', ctypes.c_void_p), ('v_snap', ctypes.c_void_p),
                ('wp', ctypes.c_void_p), ('gwp', ctypes.c_void_p), ('gw', ctypes.c_void_p),
                ('K', ctypes.c_int), ('C', ctypes.c_int), ('T', ctypes.c_int), ('ldw', ctypes.c_int),
                ('fixed_scale', ctypes.c_float)]


class SnBatch(ctypes.Structure):
    """contrad_sn_batch."""
    _fields_ = [('n', ctypes.c_int), ('layers', SnLayer * SN_MAX_LAYERS),
                ('scratch_off', ctypes.c_longlong * SN_MAX_LAYERS)]


class AdamTensor(ctypes.Structure):
    _fields_ = [('p', ctypes.c_void_p), ('g', ctypes.c_void_p), ('m', ctypes.c_void_p), ('v', ctypes.c_void_p),
                ('numel', ctypes.c_longlong)]


class AdamBatch(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int), ('t', AdamTensor * ADAM_MAX_TENSORS),
                ('block_start', ctypes.c_int * (ADAM_MAX_TENSORS + 1))]


MODCONV_MAX_LAYERS = 32


class ModconvLayer(ctypes.Structure):
    """contrad_modconv_layer."""
    _fields_ = [('w', ctypes.c_void_p), ('wp', ctypes.c_void_p), ('wsq', ctypes.c_void_p),
                ('Cout', ctypes.c_int), ('Cin', ctypes.c_int), ('T', ctypes.c_int), ('ldw', ctypes.c_int),
                ('transposed', ctypes.c_int), ('scale', ctypes.c_float)]


class ModconvBatch(ctypes.Structure):
    """contrad_modconv_batch."""
    _fields_ = [('n', ctypes.c_int), ('layers', ModconvLayer * MODCONV_MAX_LAYERS)]


class DemodLayer(ctypes.Structure):
    """contrad_demod_layer."""
    _fields_ = [('style', ctypes.c_void_p), ('wsq', ctypes.c_void_p), ('out', ctypes.c_void_p),
                ('Cin', ctypes.c_int), ('K', ctypes.c_int)]


class DemodBatch(ctypes.Structure):
    """contrad_demod_batch."""
    _fields_ = [('n', ctypes.c_int), ('B', ctypes.c_int), ('layers', DemodLayer * MODCONV_MAX_LAYERS)]


_CTYPES = {
    'int': ctypes.c_int,
    'float': ctypes.c_float,
    'double': ctypes.c_double,
    'long long': ctypes.c_longlong,
    'contrad_stream_t': ctypes.c_void_p,
    'void': None,
}


def _map_type(t):
    t = t.strip()
    if t.endswith('*'):
        base = t[:-1].replace('const', '').strip()
        if base == 'contrad_conv_desc':
            return ctypes.POINTER(ConvDesc)
        if base == 'int':
            return ctypes.POINTER(ctypes.c_int)
        return ctypes.c_void_p
    t = t.replace('const', '').strip()
    return _CTYPES[t]


def parse_header(path=HEADER_PATH):
    """Return {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int|long long|void|double)\s+(contrad_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                mm = re.match(r'(.*?)(\w+)$', a)       # strip the parameter name
                argtypes.append(_map_type(mm.group(1)))
        protos[name] = (_CTYPES[ret], argtypes)
    return protos


class _Lib(object):
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'contrad_amd: %s not found -- run `python -c "import __graft_entry__ as g; g.build()"` '
                '(or python contrad_amd/build.py).  There is no CPU / PyTorch fallback.' % LIB_PATH)
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, argtypes) in self.protos.items():
            fn = getattr(self._dll, name)           # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = argtypes
            setattr(self, '_raw_' + name, fn)

    def call(self, name, *args):
        """Invoke an int-returning launcher; non-zero status -> RuntimeError (TORCH_CHECK analogue)."""
        rc = getattr(self, '_raw_' + name)(*args)
        if rc != 0:
            raise RuntimeError('contrad_hip: %s failed with status %d' % (name, rc))

    def raw(self, name):
        return getattr(self, '_raw_' + name)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
