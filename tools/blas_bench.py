"""What the vendor GEMM (hipBLASLt / rocBLAS through torch) reaches on the plain GEMM shapes of the XL/2 step --
a yardstick for gemm_nt8 / gemm_tn8 (no fused epilogue on either side; bf16 in, bf16 out, fp32 accumulate).
    python tools/blas_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: F401,E402  (experiments build of the library: the timing switches are not in the product)
from maskdit_amd import ops  # noqa: E402
from maskdit_amd._lib import lib  # noqa: E402


def t_us(fn, n=10):
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
    M = 131072
    torch.manual_seed(0)
    print(f'{"NT  out[M,N] = A[M,K] W[N,K]^T":40s} {"torch us":>9} {"TF/s":>6} {"ours us":>9} {"TF/s":>6} {"ours K loop":>11} {"TF/s":>6}')
    for n, k, name in [(3456, 1152, 'qkv fwd'), (1152, 1152, 'proj dgrad'), (1152, 4608, 'fc1 dgrad'), (1152, 3456, 'qkv dgrad'),
                       (4608, 1152, 'fc1 (plain)')]:
        A = (torch.randn(M, k, device='cuda') * 0.5).bfloat16()
        W = (torch.randn(n, k, device='cuda') * 0.05).bfloat16()
        b = torch.randn(n, device='cuda')
        out = torch.empty(M, n, device='cuda', dtype=torch.bfloat16)
        fl = 2.0 * M * n * k
        t0 = t_us(lambda: torch.matmul(A, W.t(), out=out))
        t1 = t_us(lambda: ops.gemm_nt(A, W, bias=b, epi=ops.EPI_BF16, out=out))
        lib().mdt_set_tuning(b'nt8_skip_epilogue', 1)
        t2 = t_us(lambda: ops.gemm_nt(A, W, bias=b, epi=ops.EPI_BF16, out=out))
        lib().mdt_set_tuning(b'nt8_skip_epilogue', 0)
        print(f'{name + f" {M}x{n}x{k}":40s} {t0:9.1f} {fl / t0 / 1e6:6.0f} {t1:9.1f} {fl / t1 / 1e6:6.0f} {t2:11.1f} {fl / t2 / 1e6:6.0f}')
    print(f'{"TN  out[N1,N2] = A[M,N1]^T B[M,N2]":40s} {"torch us":>9} {"TF/s":>6} {"ours us":>9} {"TF/s":>6}   (torch: bf16 out; ours: fp32 accumulate into out)')
    for n1, n2, name in [(1152, 3456, 'qkv wgrad'), (1152, 1152, 'proj wgrad'), (1152, 4608, 'fc1 wgrad'), (4608, 1152, 'fc2 wgrad')]:
        A = torch.randn(M, n1, device='cuda').bfloat16()
        B = torch.randn(M, n2, device='cuda').bfloat16()
        out = torch.empty(n1, n2, device='cuda', dtype=torch.bfloat16)
        Cc = torch.zeros(n1, n2, device='cuda')
        fl = 2.0 * M * n1 * n2
        t0 = t_us(lambda: torch.matmul(A.t(), B, out=out))
        t1 = t_us(lambda: ops.gemm_tn(A, B, Cc))
        print(f'{name + f" {M}x{n1}x{n2}":40s} {t0:9.1f} {fl / t0 / 1e6:6.0f} {t1:9.1f} {fl / t1 / 1e6:6.0f}')


if __name__ == '__main__':
    main()
