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
    hdr = open(os.path.join(ROOT, 'include', 'maskdit_hip.h')).read()
    assert built.mdt_version() == _lib.ABI_VERSION == int(re.search(r'#define MDT_ABI_VERSION (\d+)', hdr).group(1))
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
    # the fp32-faithful entries (round 6)
    g = _lib.GemmF32Args()
    assert built.mdt_gemm_f32(C.byref(g), None) != 0 and b'null pointer' in built.mdt_last_error()
    g.A, g.B, g.out, g.lda, g.ldb, g.ldo, g.M, g.N, g.K = 16, 16, 16, 8, 8, 8, 4, 4, 6
    assert built.mdt_gemm_f32(C.byref(g), None) != 0 and b'multiple of 4' in built.mdt_last_error()
    g.K, g.epi = 8, 3   # GATE_RES without a residual
    assert built.mdt_gemm_f32(C.byref(g), None) != 0 and b'GATE_RES' in built.mdt_last_error()
    g.epi, g.batch, g.heads = 0, 6, 4
    assert built.mdt_gemm_f32(C.byref(g), None) != 0 and b'multiple of heads' in built.mdt_last_error()
    assert built.mdt_attn_f32_ws_floats(2, 256, 16, 72) == 0            # the fused kernel's domain: no workspace
    assert built.mdt_attn_f32_ws_floats(2, 1024, 16, 72) == 2 * 16 * 1024 * 1024
    assert built.mdt_attn_f32(16, 16, None, 2, 1024, 16, 72, None) != 0 and b'scores_ws' in built.mdt_last_error()
    assert built.mdt_softmax_rows_f32(None, 4, 256, 256, 1.0, None) != 0
    assert built.mdt_ln_modulate_f32(16, 16, 16, 8, 4, 16, 8, 1300, None) != 0 and b'bad shape' in built.mdt_last_error()


def test_experiment_switches_are_not_in_the_product_library(built):
    """the timing-decomposition switches that make kernels skip work (garbage results) and the
    experiment kernels behind them exist only in `make experiments` (libmaskdit_hip_exp.so).  The product library
    refuses the keys and contains neither the E_TRK (class 5) nor the phase-placement (`nt8_sched`) kernels."""
    import subprocess
    for key in (b'nt8_sched', b'nt8_skip_epilogue', b'nt8_trickle', b'attn_dbg'):
        assert built.mdt_set_tuning(key, 1) != 0, key
        assert b'experiments build' in built.mdt_last_error()
    for bad in (1, 2, 4, 7):
        assert built.mdt_set_tuning(b'tn8_dbg', bad) != 0
    assert built.mdt_set_tuning(b'tn8_dbg', 8) == 0 and built.mdt_set_tuning(b'tn8_dbg', 0) == 0  # the A/B bits give correct results
    # the device code objects are embedded in the host library: their kernel names appear as plain strings
    names = subprocess.run(['strings', '-n', '12', _lib.LIB_PATH], capture_output=True, text=True).stdout
    kernels = set(re.findall(r'_Z15gemm_nt8_kernelILi\dELi\dELi(\d)ELi(\d+)EEv8NTParams', names))
    assert kernels, 'no gemm_nt8 kernel names found in the library'
    assert all(cls != '5' for cls, _ in kernels), 'the E_TRK experiment kernel is in the product library'
    assert {int(s) for _, s in kernels} <= {5, 4101, 12293}, f'phase-placement experiment kernels in the product library: {sorted(kernels)}'
    assert 'nt8x_read_stamps' not in names
