"""A/B of the weight-gradient kernel's phase forms (run on the GPU box):  python tools/tn8_ab.py [rows]
tn8_dbg bit 3 (8) = the round-2 half-phase-staggered form vs the default FINE form (one memory operation behind each
MFMA, no stagger); the qkv shape is timed WITH its fused column sums.
Checks both against an fp64 reference on a row sample, then interleaved timing rounds (best-of)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import ops  # noqa: E402
from maskdit_amd._lib import lib  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    variants = [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0', '8'])]
    torch.manual_seed(0)
    shapes = [(3456, 1152, 'qkv', True), (1152, 1152, 'proj', False), (1152, 4608, 'fc2', False), (4608, 1152, 'fc1', False),
              (1536, 512, 'dec qkv (rows x2)', False), (512, 2048, 'dec fc2 (rows x2)', False)]
    print(f'{"shape":>34} ' + ' '.join(f'{"dbg=" + str(v) + " us":>12} {"TF/s":>6}' for v in variants))
    for n1, n2, name, cs in shapes:
        rows = M * 2 if name.startswith('dec') else M
        A = (torch.randn(rows, n1, device='cuda') * 0.25).bfloat16()
        B = (torch.randn(rows, n2, device='cuda') * 0.5).bfloat16()
        ref = torch.zeros(n1, n2, device='cuda', dtype=torch.float64)
        for r0 in range(0, rows, 16384):
            ref += A[r0:r0 + 16384].double().t() @ B[r0:r0 + 16384].double()
        best = {}
        msg = []
        for v in variants:
            lib().mdt_set_tuning(b'tn8_dbg', v)
            Cc = torch.zeros(n1, n2, device='cuda')
            csv = torch.zeros(n1, device='cuda') if cs else None
            ops.gemm_tn(A, B, Cc, colsum_a=csv)
            e = ((Cc.double() - ref).abs().max() / ref.abs().max()).item()
            ok = e < 2e-5
            if cs:
                ec = ((csv.double() - A.double().sum(0)).abs().max() / A.double().sum(0).abs().max()).item()
                ok = ok and ec < 2e-5
            msg.append(f'dbg={v}: err {e:.1e}' + ('' if ok else ' WRONG'))
        for rnd in range(3):
            for v in variants:
                lib().mdt_set_tuning(b'tn8_dbg', v)
                Cc = torch.zeros(n1, n2, device='cuda')
                csv = torch.zeros(n1, device='cuda') if cs else None
                ops.gemm_tn(A, B, Cc, colsum_a=csv)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm_tn(A, B, Cc, colsum_a=csv)
                e1.record()
                torch.cuda.synchronize()
                best[v] = min(best.get(v, 1e9), e0.elapsed_time(e1) * 200)
        lib().mdt_set_tuning(b'tn8_dbg', 0)
        print(f'{name + f" {rows}x{n1}x{n2}":>34} ' + ' '.join(f'{best[v]:12.1f} {2.0 * rows * n1 * n2 / best[v] / 1e6:6.0f}' for v in variants) +
              '   ' + '; '.join(msg), flush=True)
        del A, B, ref


if __name__ == '__main__':
    main()
