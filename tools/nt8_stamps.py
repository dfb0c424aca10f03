"""Where a persistent gemm_nt8 workgroup's time goes per tile (GPU box): shader-clock stamps of the first and the last
wave of workgroup 0 from the experiment kernels (nt8_sched 517 = product schedule + stamps; 773 = + whole-line stores).
Events per tile: 0 tile start (after the tile-top wait + barrier), 1 K loop done, 2 next tile's LDS-DMA issued, 3 last
epilogue store issued."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: F401,E402  (experiments build of the library: the timing switches are not in the product)
from maskdit_amd import _lib, ops  # noqa: E402


def report(tag, st, nt):
    t = st[1:nt]  # skip the first tile (cold) and the last
    kloop = (t[:, 1] - t[:, 0]).mean()
    dma = (t[:, 2] - t[:, 1]).mean()
    epi = (t[:, 3] - t[:, 2]).mean()
    turn = (st[2:nt + 1, 0] - t[:, 3]).mean()  # last store issued -> next tile running (tile-top wait + barrier)
    per = (st[2:nt + 1, 0] - t[:, 0]).mean()
    print(f'{tag}: per tile {per:8.0f} clk = K loop {kloop:8.0f} + next-tile DMA issue {dma:6.0f} + epilogue issue {epi:7.0f} + '
          f'tile-top wait {turn:7.0f}   ({nt - 1} tiles)', flush=True)


def main():
    L = _lib.lib()
    torch.manual_seed(0)
    scheds = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['517', '773'])]
    for (m, n, k) in ((131072, 4608, 1152), (131072, 3456, 1152), (131072, 1152, 4608)):
        A = (torch.randn(m, k, device='cuda') * 0.5).to(torch.bfloat16)
        Wt = (torch.randn(n, k, device='cuda') * 0.05).to(torch.bfloat16)
        b = torch.randn(n, device='cuda') * 0.1
        out = torch.empty(m, n, device='cuda', dtype=torch.bfloat16)
        nt = min(int((m // 256) * (n // (256 if n % 256 == 0 else 192)) / 256), 64) - 1
        for sched in scheds:
            L.mdt_set_tuning(b'nt8_sched', sched)
            for _ in range(3):
                ops.gemm_nt(A, Wt, bias=b, epi=ops.EPI_BF16, out=out)
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 512)()
            assert L.nt8x_read_stamps(buf) == 0
            both = np.array(buf, dtype=np.uint64).reshape(2, 64, 4).astype(np.int64)
            report(f'({n},{k}) sched {sched} wave 0', both[0], nt)
            report(f'({n},{k}) sched {sched} wave 7', both[1], nt)
            d = both[1][1:nt] - both[0][1:nt]
            print(f'    wave 7 minus wave 0 at the events: {d.mean(0).round().tolist()}')
    L.mdt_set_tuning(b'nt8_sched', 0)


if __name__ == '__main__':
    main()
