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
