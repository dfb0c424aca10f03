import sys, torch
sys.path.insert(0, '/root/repo')
from maskdit_amd import ops
def t_us(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 131072
for n1, n2, name in [(3456, 1152, 'qkv wgrad (A = dqkv)'), (512, 1152, 'decoder-layer wgrad (rows: 131072)')]:
    A = torch.randn(M, n1, device='cuda').bfloat16(); B = torch.randn(M, n2, device='cuda').bfloat16()
    Cc = torch.zeros(n1, n2, device='cuda'); cs = torch.zeros(n1, device='cuda')
    t0 = t_us(lambda: ops.gemm_tn(A, B, Cc))
    t1 = t_us(lambda: ops.colsum_bf16(A, cs))
    t2 = t_us(lambda: ops.gemm_tn(A, B, Cc, colsum_a=cs))
    print(f'{name:40s} gemm {t0:7.1f} us + colsum kernel {t1:6.1f} us = {t0 + t1:7.1f};  fused {t2:7.1f} us')
