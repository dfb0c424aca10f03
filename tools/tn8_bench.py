"""Weight-gradient GEMM (mdt_gemm_tn, contraction over rows) per shape: 256x192 ring kernel vs the 256x128 one.
    python tools/tn8_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import ops  # noqa: E402
from maskdit_amd._lib import lib  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    torch.manual_seed(0)
    shapes = [(1152, 3456, 'qkv'), (1152, 1152, 'proj'), (1152, 4608, 'fc1'), (4608, 1152, 'fc2')]
    if M == 131072:
        shapes += [(512, 1536, 'dec qkv (rows x2)'), (512, 2048, 'dec fc1 (rows x2)')]
    print(f'{"shape":>28} {"256x128 us":>11} {"TF/s":>7} {"256x192 us":>11} {"TF/s":>7}')
    for n1, n2, name in shapes:
        rows = M * 2 if name.startswith('dec') else M
        A = torch.randn(rows, n1, device='cuda').bfloat16()
        B = torch.randn(rows, n2, device='cuda').bfloat16()
        Cc = torch.zeros(n1, n2, device='cuda')
        out = []
        for wide in (1, 0):
            lib().mdt_set_tuning(b'tn8_wide', wide)
            for _ in range(3):
                ops.gemm_tn(A, B, Cc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm_tn(A, B, Cc)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            out += [us, 2.0 * rows * n1 * n2 / us / 1e6]
        lib().mdt_set_tuning(b'tn8_wide', 0)
        print(f'{name + f" {rows}x{n1}x{n2}":>28} {out[0]:11.1f} {out[1]:7.0f} {out[2]:11.1f} {out[3]:7.0f}')


if __name__ == '__main__':
    main()
