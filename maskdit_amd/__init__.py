"""maskdit_amd -- MI355X (gfx950) native engine for the MaskDiT training / sampling hot path.

Only what the hot path needs lives here: `csrc/` (hand-written HIP kernels + the C ABI of
libmaskdit_hip.so, declared in include/maskdit_hip.h) and the host-side mirror of the
reference's Python surface (Precond_models, Losses, edm_sampler, FusedAdam, update_ema).
There is no non-HIP fallback: computing without libmaskdit_hip.so raises.
"""
__version__ = '0.1.0'
