"""EXPERIMENT (round 5; needs tools/experiments/nt8_blocked.patch applied to maskdit_amd/csrc and `make experiments`): gemm_nt8 with its operands in the BLOCKED layout [rows / 8][K / 64][8][64] -- every LDS-DMA piece one
contiguous KiB instead of 8 separate 128-byte row segments (tools/micro/lds_dma_bench.hip: 36-55 vs 11-20 B/clk/CU).  GPU box.
    python tools/nt8_blocked_bench.py [--m 131072]
Per shape: row-major (product) / A blocked / B blocked / both; full launch and K loop only (experiments library), TFLOP/s | us,
and a bitwise comparison of every blocked form's outputs with the row-major launch."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _explib  # noqa: F401,E402
from maskdit_amd import _lib, ops  # noqa: E402


def block(x):
    r, k = x.shape
    return x.view(r // 8, 8, k // 64, 64).permute(0, 2, 1, 3).contiguous().view(r, k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=131072)
    ap.add_argument('--iters', type=int, default=5)
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    torch.manual_seed(0)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    M, W_ = args.m, 1152
    shapes = [((M, 3 * W_, W_), 'BF16', 'qkv fwd'), ((M, W_, W_), 'GATE_RES', 'proj fwd'), ((M, 4 * W_, W_), 'GELU', 'fc1 fwd'),
              ((M, W_, 4 * W_), 'GATE_RES', 'fc2 fwd'), ((M, W_, 4 * W_), 'BF16', 'fc1 dgrad'), ((M, W_, 3 * W_), 'BF16', 'qkv dgrad'),
              ((2 * M, 512, 2048), 'GATE_RES', 'dec fc2')]
    forms = [('row-major', 0), ('A blocked', 1), ('B blocked', 2), ('A+B blocked', 3)]
    print(f'{"shape / epilogue":42s} ' + ' '.join(f'{f[0]:>15s}' for f in forms) + '   | K loop only: ' + ' '.join(f'{f[0]:>13s}' for f in forms) + '   (TFLOP/s | us)')
    for (m, n, k), name, tag in shapes:
        A = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        Wt = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        Ab, Wb = block(A), block(Wt)
        kw = dict(bias=torch.randn(n, device=dev) * 0.1, epi=getattr(ops, 'EPI_' + name))
        if name == 'GATE_RES':
            kw.update(res=torch.randn(m, n, device=dev), gate=torch.randn(m // 128, n, device=dev), gate_ld=n, rows_per_sample=128)
        ref, res, same = None, {}, {}
        for rnd in range(3):  # interleaved rounds, best-of: the first launches after the allocations run slow
            for skip in (0, 1):
                for fname, bits in forms:
                    L.mdt_set_tuning(b'nt8_blocked', bits)
                    L.mdt_set_tuning(b'nt8_skip_epilogue', skip)
                    a_, w_ = (Ab if bits & 1 else A), (Wb if bits & 2 else Wt)
                    o = ops.gemm_nt(a_, w_, **kw)
                    if not skip and rnd == 0:
                        outs = [t for t in o if t is not None]
                        if ref is None:
                            ref = [t.clone() for t in outs]
                        same[fname] = all(torch.equal(x.view(torch.int16 if x.dtype == torch.bfloat16 else torch.int32),
                                                      y.view(torch.int16 if y.dtype == torch.bfloat16 else torch.int32)) for x, y in zip(outs, ref))
                    L.mdt_event_record(ev[0], st)
                    for _ in range(args.iters):
                        ops.gemm_nt(a_, w_, **kw)
                    L.mdt_event_record(ev[1], st)
                    torch.cuda.synchronize()
                    ms = C.c_float()
                    L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                    res[(skip, fname)] = min(res.get((skip, fname), 1e9), ms.value / args.iters)
        L.mdt_set_tuning(b'nt8_blocked', 0)
        L.mdt_set_tuning(b'nt8_skip_epilogue', 0)
        f = 2.0 * m * n * k
        cell = lambda t: f'{f / t / 1e9:6.0f} |{t * 1e3:6.0f}'
        print(f'{str((m, n, k)) + " " + name + " " + tag:42s} ' + ' '.join(f'{cell(res[(0, fn)]):>15s}' for fn, _ in forms) + '   |              '
              + ' '.join(f'{cell(res[(1, fn)]):>13s}' for fn, _ in forms) + '   bitwise == row-major: ' + ', '.join(f'{fn} {same[fn]}' for fn, _ in forms[1:]), flush=True)
        del A, Wt, Ab, Wb, kw, ref
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
