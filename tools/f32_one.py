"""One mdt_gemm_f32 shape, a few launches (for rocprofv3 --pmc passes):  python tools/f32_one.py [N K epi iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import ops  # noqa: E402

M = 32768
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4608, 1152)
epi = sys.argv[3] if len(sys.argv) > 3 else 'NONE'
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
torch.manual_seed(0)
A = torch.randn(M, K, device='cuda')
W = torch.randn(N, K, device='cuda') * K ** -0.5
b = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda')
kw = dict(bias=b, epi=getattr(ops, 'F32EPI_' + epi))
if epi == 'GATE_RES':
    kw.update(res=torch.randn(M, N, device='cuda'), gate=torch.randn(M // 256, N, device='cuda'), gate_ld=N, rows_per_sample=256)
for _ in range(iters):
    ops.gemm_f32(A, W, out, M, N, K, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.gemm_f32(A, W, out, M, N, K, **kw)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / iters
print(f'gemm_f32 {M}x{N}x{K} {epi}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.1f} TF/s')
