"""fp32-faithful path (csrc/f32path.hip): mdt_gemm_f32 per XL/2 inference shape + the three-launch attention, TF/s against
the 157.3 TFLOP/s fp32 matrix peak.      python tools/f32_bench.py [rows = 32768]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import ops  # noqa: E402

PEAK = 157.3


def timed(fn, iters=5):
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    torch.manual_seed(0)
    print(f'{"shape":>34} {"epilogue":>9} {"us":>9} {"TF/s":>7} {"of peak":>8}')
    for N, K, epi, name in [(3456, 1152, 'NONE', 'qkv'), (1152, 1152, 'GATE_RES', 'proj'), (4608, 1152, 'GELU', 'fc1'),
                            (1152, 4608, 'GATE_RES', 'fc2'), (1536, 512, 'NONE', 'dec qkv'), (512, 512, 'GATE_RES', 'dec proj'),
                            (2048, 512, 'GELU', 'dec fc1'), (512, 2048, 'GATE_RES', 'dec fc2')]:
        A = torch.randn(M, K, device='cuda')
        W = torch.randn(N, K, device='cuda') * K ** -0.5
        b = torch.randn(N, device='cuda')
        out = torch.empty(M, N, device='cuda')
        kw = dict(bias=b, epi=getattr(ops, 'F32EPI_' + epi))
        if epi == 'GATE_RES':
            kw.update(res=torch.randn(M, N, device='cuda'), gate=torch.randn(M // 256, N, device='cuda'), gate_ld=N, rows_per_sample=256)
        us = timed(lambda: ops.gemm_f32(A, W, out, M, N, K, **kw))
        tf = 2.0 * M * N * K / us / 1e6
        print(f'{name + f" {M}x{N}x{K}":>34} {epi:>9} {us:9.1f} {tf:7.1f} {tf / PEAK:8.3f}')
    for H, hd, name in [(16, 72, 'encoder attention'), (16, 32, 'decoder attention')]:
        B, L = M // 256, 256
        qkv = torch.randn(B * L, 3 * H * hd, device='cuda')
        for three, tag in ((False, 'fused'), (True, '3 launch')):
            us = timed(lambda: ops.attention_f32(qkv, B, L, H, hd, three_launch=three))
            tf = 4.0 * B * H * L * L * hd / us / 1e6
            print(f'{name + f" B{B} L{L} H{H} hd{hd}":>34} {tag:>9} {us:9.1f} {tf:7.1f} {tf / PEAK:8.3f}')
    x = torch.randn(M, 1152, device='cuda')
    mod = torch.randn(M // 256, 3 * 1152, device='cuda')
    us = timed(lambda: ops.ln_modulate_f32(x, mod[:, :1152], mod[:, 2304:], 3456, 256))
    print(f'{"ln_modulate_f32 " + str(M) + "x1152":>34} {"":>9} {us:9.1f} {M * 1152 * 8 / us / 1e6:7.2f} TB/s')


if __name__ == '__main__':
    main()
