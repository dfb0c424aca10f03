"""LayerNorm-modulate backward + residual-gate backward: two kernels vs the fused kernel.  Run on the GPU box:
    python tools/ln_gate_bench.py [micro_batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import ops  # noqa: E402
from maskdit_amd._lib import call, lib  # noqa: E402

DEV = 'cuda'


def sp():
    return torch.cuda.current_stream().cuda_stream


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    L, D = (128, 1152) if len(sys.argv) < 3 else (256, 512)
    M = B * L
    torch.manual_seed(0)
    x = torch.randn(M, D, device=DEV)
    mod = torch.randn(B, 3 * D, device=DEV) * 0.5
    _, stats = ops.ln_modulate_fwd(x, mod[:, :D], mod[:, 2 * D:], 3 * D, L)
    dxn = torch.randn(M, D, device=DEV).bfloat16()
    dx = torch.randn(M, D, device=DEV)
    y = torch.randn(M, D, device=DEV).bfloat16()
    gate = mod[:, D:2 * D]
    dmod = torch.zeros(B, 3 * D, device=DEV)
    dbias = torch.zeros(D, device=DEV)
    dys = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)

    def separate():
        ops.ln_modulate_bwd(dxn, x, stats, mod[:, 2 * D:], 3 * D, L, dx, True, dmod[:, :D], dmod[:, 2 * D:], 3 * D)
        call('mdt_gate_bwd', dx.data_ptr(), y.data_ptr(), gate.data_ptr(), 3 * D, L, dys.data_ptr(), dmod[:, D:2 * D].data_ptr(),
             3 * D, dbias.data_ptr(), M, D, sp())

    def fused():
        call('mdt_ln_modulate_bwd_gate', dxn.data_ptr(), x.data_ptr(), stats.data_ptr(), mod[:, 2 * D:].data_ptr(), 3 * D, L,
             dx.data_ptr(), 1, dmod[:, :D].data_ptr(), dmod[:, 2 * D:].data_ptr(), 3 * D, M, D, y.data_ptr(), gate.data_ptr(),
             3 * D, dys.data_ptr(), dmod[:, D:2 * D].data_ptr(), 3 * D, dbias.data_ptr(), sp())

    el = M * D
    t = timeit(separate)
    print(f'separate (22 B/el)      {t:8.1f} us  {22 * el / t / 1e6:6.2f} TB/s')
    t = timeit(fused)
    print(f'fused, column-split     {t:8.1f} us  {18 * el / t / 1e6:6.2f} TB/s  (18 B/el)')
    lib().mdt_set_tuning(b'ln_gate_rowwise', 1)
    t = timeit(fused)
    print(f'fused, row per wave     {t:8.1f} us  {18 * el / t / 1e6:6.2f} TB/s  (18 B/el)')
    lib().mdt_set_tuning(b'ln_gate_rowwise', 0)


if __name__ == '__main__':
    main()
