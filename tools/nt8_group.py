"""Tile-order group (nt8_group_m) sweep of gemm_nt8 at M = 131072 (GPU box): full launch us per group size."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import _lib, ops  # noqa: E402


def main():
    L = _lib.lib()
    torch.manual_seed(0)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    m, W_ = 131072, 1152
    groups = [1, 2, 4, 8, 16, 32]
    print('shape                 ' + ' '.join(f'g={g:>2d}' .rjust(9) for g in groups))
    for n, k in ((3 * W_, W_), (W_, 4 * W_), (W_, W_), (W_, 3 * W_), (4 * W_, W_)):
        A = (torch.randn(m, k, device='cuda') * 0.5).to(torch.bfloat16)
        Wt = (torch.randn(n, k, device='cuda') * 0.05).to(torch.bfloat16)
        b = torch.randn(n, device='cuda') * 0.1
        out = torch.empty(m, n, device='cuda', dtype=torch.bfloat16)
        best = {}
        for r in range(3):
            for g in groups:
                L.mdt_set_tuning(b'nt8_group_m', g)
                ops.gemm_nt(A, Wt, bias=b, epi=ops.EPI_BF16, out=out)
                L.mdt_event_record(ev[0], st)
                for _ in range(4):
                    ops.gemm_nt(A, Wt, bias=b, epi=ops.EPI_BF16, out=out)
                L.mdt_event_record(ev[1], st)
                torch.cuda.synchronize()
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                best[g] = min(best.get(g, 1e9), ms.value / 4)
        print(f'({n:4d},{k:4d})          ' + ' '.join(f'{best[g] * 1e3:9.0f}' for g in groups), flush=True)
    L.mdt_set_tuning(b'nt8_group_m', 0)


if __name__ == '__main__':
    main()
