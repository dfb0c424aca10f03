"""Phase-placement A/B of the gemm_nt8 K loop (run on the GPU box):  python tools/nt8_sched.py [--m 131072]
Plain-bf16 epilogue, the five plain XL/2 shapes; for every nt8_sched variant (gemm_nt8_impl.h PAIR_BODY): bitwise
comparison with the default kernel, then TFLOP/s | us of the K loop alone (nt8_skip_epilogue) and of the full launch.
Interleaved rounds, best-of."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: F401,E402  (experiments build of the library: the timing switches are not in the product)
from maskdit_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=131072)
    ap.add_argument('--iters', type=int, default=4)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--kloop-only', action='store_true')
    ap.add_argument('--scheds', type=str, default='0,1029')
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    torch.manual_seed(0)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    M, W_ = args.m, 1152
    shapes = [((M, 3 * W_, W_), 'qkv fwd'), ((M, W_, 4 * W_), 'fc1 dgrad'), ((M, W_, W_), 'proj dgrad'),
              ((M, W_, 3 * W_), 'qkv dgrad'), ((M, 4 * W_, W_), '4608x1152 (NF 4)')]
    scheds = [int(v) for v in args.scheds.split(',')]
    print(f'{"shape":34s} ' + ' '.join(f'{"s" + str(v):>22s}' for v in scheds) + '    (K loop TF/s | us || full TF/s | us)')
    for (m, n, k), tag in shapes:
        A = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        Wt = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.randn(n, device=dev) * 0.1
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        kw = dict(bias=b, epi=ops.EPI_BF16, out=out)
        L.mdt_set_tuning(b'nt8_skip_epilogue', 0)
        L.mdt_set_tuning(b'nt8_sched', 0)
        ops.gemm_nt(A, Wt, **kw)
        ref = out.clone()
        bad = []
        for v in scheds:
            L.mdt_set_tuning(b'nt8_sched', v)
            out.zero_()
            ops.gemm_nt(A, Wt, **kw)
            if not torch.equal(out, ref):
                bad.append((v, (out.float() - ref.float()).abs().max().item()))
        best = {}
        for r in range(args.rounds):
            for skip in ((1,) if args.kloop_only else (1, 0)):
                for v in scheds:
                    L.mdt_set_tuning(b'nt8_skip_epilogue', skip)
                    L.mdt_set_tuning(b'nt8_sched', v)
                    ops.gemm_nt(A, Wt, **kw)
                    L.mdt_event_record(ev[0], st)
                    for _ in range(args.iters):
                        ops.gemm_nt(A, Wt, **kw)
                    L.mdt_event_record(ev[1], st)
                    torch.cuda.synchronize()
                    ms = C.c_float()
                    L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                    best[(v, skip)] = min(best.get((v, skip), 1e9), ms.value / args.iters)
        f = 2.0 * m * n * k
        cells = [f'{f / best[(v, 1)] / 1e9:5.0f}|{best[(v, 1)] * 1e3:5.0f} ||{f / best.get((v, 0), 1e9) / 1e9:5.0f}|{best.get((v, 0), 0) * 1e3:5.0f}' for v in scheds]
        print(f'{str((n, k)) + " " + tag:34s} ' + ' '.join(f'{c:>22s}' for c in cells) + (f'   MISMATCH {bad}' if bad else '   bitwise ok'), flush=True)
        del A, Wt, kw, out, ref
        torch.cuda.empty_cache()
    L.mdt_set_tuning(b'nt8_skip_epilogue', 0)
    L.mdt_set_tuning(b'nt8_sched', 0)


if __name__ == '__main__':
    main()
