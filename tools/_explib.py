"""Imported FIRST by the tools that use timing-decomposition switches (nt8_skip_epilogue, nt8_sched, nt8_trickle,
attn_dbg, tn8_dbg bits 0-2): those make kernels skip work and exist only in the experiments build of the library
(`make -C maskdit_amd/csrc experiments` -> maskdit_amd/libmaskdit_hip_exp.so, -DMDT_EXPERIMENTS).  Points MASKDIT_HIP_LIB
at it (building it on first use: hipcc cross-compiles, ~1 minute) unless the caller already chose a library."""
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_EXP = os.path.join(_ROOT, 'maskdit_amd', 'libmaskdit_hip_exp.so')
if 'MASKDIT_HIP_LIB' not in os.environ:
    if not os.path.exists(_EXP):
        subprocess.run(['make', '-C', os.path.join(_ROOT, 'maskdit_amd', 'csrc'), '-j', '8', 'experiments'], check=True, capture_output=True)
    os.environ['MASKDIT_HIP_LIB'] = _EXP
