"""What a persistent / stream-K gemm_tn8 could save, measured on the existing kernel instead of
estimated: one weight-gradient shape at fixed widths, the contraction length M and the split count varied.  A launch is
`waves` rounds of workgroups (blocks / CUs, rounded up), each block = slots-per-split phases + a fixed cost (ring fill,
dispatch, 96 fp32 atomics per lane, ring drain): time ~ waves * (slots * c + f).  The least-squares (c, f) over the table
gives the fixed cost f in microseconds and as a share of the product launch -- the most stream-K can remove is the
difference between today's (blocks / CU) fixed costs per CU and the ~1.4 a contiguous-span walk would pay.
    python tools/tn8_fixed_cost.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import ops  # noqa: E402


def timed(fn, iters=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


def main():
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    torch.manual_seed(0)
    for n1, n2, name in [(4608, 1152, 'fc1 wgrad'), (3456, 1152, 'qkv wgrad')]:
        tiles = ((max(n1, n2) + 255) // 256) * (min(n1, n2) // 192)
        rows_x, rows_y, rec = [], [], []
        print(f'--- {name} {n1} x {n2}: {tiles} tiles of 256 x 192, {cus} CUs')
        print(f'{"M":>8} {"splits":>6} {"blocks":>6} {"waves":>6} {"slots/blk":>9} {"us":>9} {"TF/s":>7}')
        for M in (16384, 32768, 65536, 131072):
            A = torch.randn(M, n1, device='cuda').bfloat16()
            B = torch.randn(M, n2, device='cuda').bfloat16()
            Cc = torch.zeros(n1, n2, device='cuda')
            slots = M // 32
            for sp in (0, 1, 2, 3, 4, 5, 7, 9, 12, 14, 19):
                if sp and slots // sp < 32:
                    continue
                us = timed(lambda: ops.gemm_tn(A, B, Cc, splits=sp))
                if sp:
                    per = -(-slots // sp)
                    nsp = -(-slots // per)
                    blocks = tiles * nsp
                    waves = -(-blocks // cus)
                    rows_x.append([waves * per, waves])
                    rows_y.append(us)
                    rec.append((M, sp, blocks, waves, per, us))
                    print(f'{M:8d} {sp:6d} {blocks:6d} {waves:6d} {per:9d} {us:9.1f} {2.0 * M * n1 * n2 / us / 1e6:7.0f}')
                else:
                    print(f'{M:8d} {"auto":>6} {"":>6} {"":>6} {"":>9} {us:9.1f} {2.0 * M * n1 * n2 / us / 1e6:7.0f}   <- product dispatch')
        X, y = np.array(rows_x, dtype=np.float64), np.array(rows_y)
        (c, f), res, *_ = np.linalg.lstsq(X, y, rcond=None)
        pred = X @ np.array([c, f])
        print(f'fit: time = waves * (slots * {c * 1000:.1f} ns + {f:.1f} us); rms residual {np.sqrt(np.mean((pred - y) ** 2)):.1f} us; '
              f'fixed cost = {f / c:.0f} slots of work')
        for M in (16384, 131072):
            slots = M // 32
            # today's product dispatch vs an ideal contiguous-span walk: every CU does tiles * slots / cus slots and pays
            # ~ (1 + tiles / cus) fixed costs (one per span start + one per tile boundary inside its span)
            ideal = tiles * slots / cus * c + (1.0 + tiles / cus) * f
            best = min(r[5] for r in rec if r[0] == M)
            print(f'  M = {M}: best measured {best:.1f} us; ideal stream-K walk by this model {ideal:.1f} us ({100 * (1 - ideal / best):.1f} % less)')


if __name__ == '__main__':
    main()
