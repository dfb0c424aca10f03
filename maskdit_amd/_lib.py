"""ctypes binding of libmaskdit_hip.so (the C ABI declared in include/maskdit_hip.h).

The library is the product: if it is missing or fails to load, everything in this package
that computes fails loudly -- there is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# MASKDIT_HIP_LIB: load ANOTHER build of the same C ABI instead of the in-tree product library -- the experiments build
# (`make -C maskdit_amd/csrc experiments` -> libmaskdit_hip_exp.so, the timing-decomposition switches of tools/*) or a
# historical binary a regression test is checked against (tools/experiments/).  Never set on the product path.
LIB_PATH = os.environ.get('MASKDIT_HIP_LIB') or os.path.join(_HERE, 'libmaskdit_hip.so')
CSRC = os.path.join(_HERE, 'csrc')

vp = C.c_void_p
i32 = C.c_int
i64 = C.c_long
f32 = C.c_float

EPI_BF16, EPI_F32, EPI_GELU, EPI_SILU, EPI_GATE_RES, EPI_DGELU, EPI_DSILU = range(7)


class GemmNTArgs(C.Structure):
    _fields_ = [('A', vp), ('lda', i32), ('B', vp), ('ldb', i32), ('M', i32), ('N', i32), ('K', i32),
                ('bias', vp), ('epi', i32), ('out', vp), ('ldo', i32), ('out2', vp), ('ldo2', i32),
                ('outf', vp), ('ldof', i32), ('res', vp), ('ldres', i32), ('gate', vp), ('gate_ld', i32),
                ('rows_per_sample', i32), ('aux', vp), ('ldaux', i32), ('k_splits', i32), ('colsum', vp)]


class GemmTNArgs(C.Structure):
    _fields_ = [('A', vp), ('lda', i32), ('B', vp), ('ldb', i32), ('M', i32), ('N1', i32), ('N2', i32),
                ('C', vp), ('ldc', i32), ('n1_valid', i32), ('n2_valid', i32), ('splits', i32), ('colsum_a', vp)]


class GemmF32Args(C.Structure):
    _fields_ = [('A', vp), ('lda', i64), ('B', vp), ('ldb', i64), ('b_kmajor', i32), ('M', i32), ('N', i32), ('K', i32),
                ('bias', vp), ('epi', i32), ('out', vp), ('ldo', i64), ('res', vp), ('ldres', i64), ('gate', vp),
                ('gate_ld', i64), ('rows_per_sample', i32), ('batch', i32), ('heads', i32),
                ('a_stride_b', i64), ('a_stride_h', i64), ('b_stride_b', i64), ('b_stride_h', i64),
                ('o_stride_b', i64), ('o_stride_h', i64)]


F32EPI_NONE, F32EPI_GELU, F32EPI_SILU, F32EPI_GATE_RES = range(4)

# name -> argtypes (the trailing stream argument is added to every compute entry)
_PROTOS = {
    'mdt_gemm_nt': [C.POINTER(GemmNTArgs)],
    'mdt_gemm_tn': [C.POINTER(GemmTNArgs)],
    'mdt_attn_fwd': [vp, vp, vp, i32, i32, i32, i32, i32],
    'mdt_attn_bwd': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32],
    'mdt_ln_modulate_fwd': [vp, vp, vp, i32, i32, vp, vp, i32, i32],
    'mdt_ln_modulate_fwd_res': [vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, i32, i32],
    'mdt_ln_modulate_bwd': [vp, vp, vp, vp, i32, i32, vp, i32, vp, vp, i32, i32, i32],
    'mdt_ln_modulate_bwd_gate': [vp, vp, vp, vp, i32, i32, vp, i32, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp],
    'mdt_gate_bwd': [vp, vp, vp, i32, i32, vp, vp, i32, vp, i32, i32],
    'mdt_colsum_bf16': [vp, i32, vp, i32, i32],
    'mdt_mask_sort': [vp, i32, i32, i32, vp, vp, vp, vp],
    'mdt_patch_embed_fwd': [vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32],
    'mdt_patch_embed_bwd': [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32],
    'mdt_timestep_embed': [vp, vp, i32, i32, i32],
    'mdt_cast_f32_bf16': [vp, i32, vp, i32, i32, i32, i32],
    'mdt_add_f32': [vp, vp, vp, i64],
    'mdt_silu_bwd': [vp, vp, vp, i64],
    'mdt_unmask_fwd': [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32],
    'mdt_unmask_bwd': [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32],
    'mdt_final_fwd': [vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32],
    'mdt_final_bwd': [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32],
    'mdt_edm_prep': [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32],
    'mdt_edm_loss_fwd': [vp, vp, vp, vp, vp, f32, vp, vp, i32, i32, i32, i32],
    'mdt_edm_loss_bwd': [vp, vp, vp, vp, vp, vp, f32, vp, i32, i32, i32, i32],
    'mdt_precond_coef': [vp, vp, i32, f32],
    'mdt_scale_rows': [vp, vp, i32, vp, i32, i32],
    'mdt_precond_out': [vp, vp, vp, vp, i32, i32],
    'mdt_sample_moments': [vp, vp, vp, i32, i32, f32],
    'mdt_class_dropout': [vp, vp, f32, i32, i32],
    'mdt_adamw_ema_step': [vp, vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, f32, f32, f32],
    'mdt_ema_update': [vp, vp, i64, f32],
    'mdt_transpose_bf16_batched': [vp, vp, vp, i32, i32],
    'mdt_sampler_prep': [vp, vp, vp, i32, vp, vp, i32, i32, i32, f32],
    'mdt_sampler_euler': [vp, vp, vp, vp, f32, i32, vp, vp, i32, i32, f32],
    'mdt_sampler_heun': [vp, vp, vp, vp, vp, vp, f32, i32, i32, i32, f32],
    'mdt_sampler_advance': [vp],
    'mdt_cfg_combine': [vp, f32, vp, i64],
    'mdt_gn_stats': [vp, vp, i32, i32, i32, i32],
    'mdt_gn_im2col': [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32],
    'mdt_conv3x3_nhwc': [vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, vp, i32],
    'mdt_softmax_rows': [vp, vp, i32, i32, f32],
    'mdt_vae_prologue': [vp, vp, vp, vp, i32, i32, f32],
    'mdt_vae_epilogue': [vp, i32, vp, i32, i32, i32],
    'mdt_lds_poison': [vp],
    'mdt_gemm_f32': [C.POINTER(GemmF32Args)],
    'mdt_softmax_rows_f32': [vp, i64, i32, i32, f32],
    'mdt_attn_f32': [vp, vp, vp, i32, i32, i32, i32],
    'mdt_ln_modulate_f32': [vp, vp, vp, i32, i32, vp, i32, i32],
    'mdt_timestep_embed_f32': [vp, vp, i32, i32, i32],
    'mdt_silu_f32': [vp, vp, i64],
    'mdt_add_rows_f32': [vp, vp, vp, i64, i32, i32],
}
# entries without the trailing stream
_PLAIN = {
    'mdt_graph_begin': [vp],
    'mdt_graph_end': [vp, C.POINTER(vp)],
    'mdt_graph_launch': [vp, vp],
    'mdt_graph_destroy': [vp],
    'mdt_event_create': [C.POINTER(vp)],
    'mdt_event_record': [vp, vp],
    'mdt_event_elapsed_ms': [vp, vp, C.POINTER(f32)],
    'mdt_event_destroy': [vp],
    'mdt_set_tuning': [C.c_char_p, i32],
    'mdt_nt8o_report': [C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), i32],
    'mdt_nt8o_stamps': [C.POINTER(C.c_uint64)],
    'mdt_attn_f32_ws_floats': [i32, i32, i32, i32],
}
EXPORTED = sorted(list(_PROTOS) + list(_PLAIN) + ['mdt_last_error', 'mdt_version'])
ABI_VERSION = 4  # == MDT_ABI_VERSION of include/maskdit_hip.h (tests/test_capi_cpu.py compares the two)


class MaskDiTLibError(RuntimeError):
    pass


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: identifies the BINARY a profile
    was taken on (bench.py refuses a PMC traffic record of another build)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(root, 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'csrc', '*.h')) +
                   glob.glob(os.path.join(root, '..', 'include', '*.h')))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libmaskdit_hip.so (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(['make', '-C', CSRC, '-j', str(min(8, os.cpu_count() or 1))], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise MaskDiTLibError('building libmaskdit_hip.so failed (see compiler output above)')
    return LIB_PATH


_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the shared library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MaskDiTLibError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C {CSRC}`).  maskdit_amd has no non-HIP fallback.')
    L = C.CDLL(LIB_PATH)
    missing = [n for n in EXPORTED if not hasattr(L, n)]
    if missing:
        raise MaskDiTLibError(f'{LIB_PATH} does not export {missing}: it was built from other sources than this package '
                              f'(rebuild with `make -C {CSRC}`, or unset MASKDIT_HIP_LIB)')
    L.mdt_version.restype = i32
    L.mdt_version.argtypes = []
    if L.mdt_version() != ABI_VERSION:
        raise MaskDiTLibError(f'{LIB_PATH} has ABI revision {L.mdt_version()}, this package binds revision {ABI_VERSION} '
                              f'(include/maskdit_hip.h MDT_ABI_VERSION): argument lists differ -- rebuild with `make -C {CSRC}`')
    for name, argt in _PROTOS.items():
        fn = getattr(L, name)
        fn.argtypes = argt + [vp]
        fn.restype = i32
    for name, argt in _PLAIN.items():
        fn = getattr(L, name)
        fn.argtypes = argt
        fn.restype = i64 if name == 'mdt_attn_f32_ws_floats' else i32
    L.mdt_last_error.restype = C.c_char_p
    L.mdt_last_error.argtypes = []
    L.mdt_version.restype = i32
    L.mdt_version.argtypes = []
    # MDT_TUNE="key=value,key=value": process-wide tuning knobs for A/B runs (mdt_set_tuning)
    for item in filter(None, os.environ.get('MDT_TUNE', '').split(',')):
        k, _, v = item.partition('=')
        if L.mdt_set_tuning(k.strip().encode(), int(v)) != 0:
            raise MaskDiTLibError(f'MDT_TUNE: {k.strip()}={v} refused by {LIB_PATH}: {L.mdt_last_error().decode()}')
    _lib = L
    return L


def check(rc: int, what: str = ''):
    if rc != 0:
        raise MaskDiTLibError(f'{what} failed ({rc}): {lib().mdt_last_error().decode()}')


def call(name: str, *args):
    """Checked call of a compute entry; the last positional argument must be the stream."""
    check(getattr(lib(), name)(*args), name)
