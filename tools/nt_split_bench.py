"""mdt_gemm_nt's kernel forms at badly quantised tile counts (per-GPU batch 128 of the 8-GPU configuration: M = 16384, where
N = 1152 gives 64 x 6 = 384 tiles of 256 x 192 = 1.5 rounds of the 256 CUs) -- GPU box.
    python tools/nt_split_bench.py [--m 16384]
Per shape / epilogue: auto dispatch, forced 8-wave (variant 2), forced 4-wave (3); us per launch.
Round 5 also measured a COLUMN-SPLIT dispatch with this tool (two launches: the first 768 columns as 64 x 4 = 256 tiles of
256 x 192 = exactly one round, the other 384 as 192 tiles of 256 x 128 -- busiest-CU area 320 instead of 384 columns): bit-
identical, and slower than the 4-wave form everywhere (proj + GATE_RES 72.6 vs 65.3 us, fc2 180 vs 173, proj dgrad 46.8 vs
39.4, qkv dgrad 107 vs 98, fc1 dgrad 138 vs 128: profiles/r5_nt_split_experiment.txt) -- the 4-wave form's odd third
round already runs alone on its CU at ~1.6x the shared rate, which is what the dispatcher's cost model says.  The split
was not kept (git history: round 5)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=16384)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    torch.manual_seed(0)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    M, W_ = args.m, 1152
    shapes = [((M, W_, W_), 'GATE_RES', 'proj fwd'), ((M, W_, 4 * W_), 'GATE_RES', 'fc2 fwd'), ((M, W_, W_), 'BF16', 'proj dgrad'),
              ((M, W_, 3 * W_), 'BF16', 'qkv dgrad'), ((M, W_, 4 * W_), 'BF16', 'fc1 dgrad'), ((M, 3 * W_, W_), 'BF16', 'qkv fwd (no split: N = 3456)'),
              ((M, 4 * W_, W_), 'GELU', 'fc1 fwd (N = 4608)'), ((M, 4 * W_, W_), 'DGELU', 'fc2 dgrad (N = 4608)')]
    print(f'{"shape":44s} {"auto":>9s} {"8-wave":>9s} {"4-wave":>9s}   us per launch')
    for (m, n, k), name, tag in shapes:
        A = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        Wt = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        kw = dict(bias=torch.randn(n, device=dev) * 0.1, epi=getattr(ops, 'EPI_' + name))
        if name == 'GATE_RES':
            kw.update(res=torch.randn(m, n, device=dev), gate=torch.randn(m // 128, n, device=dev), gate_ld=n, rows_per_sample=128)
        elif name == 'DGELU':
            kw.update(bias=None, aux=torch.randn(m, n, device=dev).to(torch.bfloat16))
        res, outs = {}, {}
        for vname, v in (('auto', 0), ('8-wave', 2), ('4-wave', 3)):
            L.mdt_set_tuning(b'gemm_nt_variant', v)
            o = ops.gemm_nt(A, Wt, **kw)
            outs[vname] = [t.clone() for t in o if t is not None]
            best = 1e9
            for r in range(3):
                L.mdt_event_record(ev[0], st)
                for _ in range(args.iters):
                    ops.gemm_nt(A, Wt, **kw)
                L.mdt_event_record(ev[1], st)
                torch.cuda.synchronize()
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                best = min(best, ms.value / args.iters)
            res[vname] = best * 1e3
        L.mdt_set_tuning(b'gemm_nt_variant', 0)
        same_auto = all(torch.equal(x.view(torch.int16 if x.dtype == torch.bfloat16 else torch.int32), y.view(torch.int16 if y.dtype == torch.bfloat16 else torch.int32))
                        for x, y in zip(outs['auto'], outs['8-wave']))
        print(f'{str((m, n, k)) + " " + name + " " + tag:44s} ' + ' '.join(f'{res[v]:9.1f}' for v in ('auto', '8-wave', '4-wave'))
              + f'   auto == 8-wave bitwise: {same_auto}', flush=True)


if __name__ == '__main__':
    main()
