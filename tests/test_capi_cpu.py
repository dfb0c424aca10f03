"""CPU-side checks of the C ABI: the shared library loads and exports every symbol that
include/maskdit_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

from maskdit_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, 'include', 'maskdit_hip.h')).read()
    declared = set(re.findall(r'\b(mdt_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(built, name), f'{name} declared in maskdit_hip.h but not exported'
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)


def test_version_and_error_string(built):
    assert built.mdt_version() >= 1
    assert isinstance(built.mdt_last_error(), bytes)


def test_argument_validation_without_gpu(built):
    """Argument checks run before any launch, so they are testable on a CPU-only host."""
    import ctypes as C
    a = _lib.GemmNTArgs()
    rc = built.mdt_gemm_nt(C.byref(a), None)
    assert rc != 0 and b'null operand' in built.mdt_last_error()
    rc = built.mdt_mask_sort(1, 4, 100, 50, None, None, None, None, None)
    assert rc != 0 and b'power of two' in built.mdt_last_error()
    rc = built.mdt_attn_fwd(1, 1, 1, 2, 100, 2, 72, 0, None)
    assert rc != 0 and b'multiple of 64' in built.mdt_last_error()
