"""Round-2 micro-benchmark of the fused-epilogue NT GEMMs at the BENCHMARKED size (run on the GPU box).
    python tools/nt8_bench.py [--m 131072] [--iters 5]
For every (shape, epilogue) of an XL/2 encoder block + the decoder shapes: TFLOP/s and microseconds of
  full      : the product path (auto dispatch)
  no-epi    : main loop only (mdt_set_tuning nt8_skip_epilogue) -> what the epilogue costs per launch
  nf3       : 192-column tiles preferred (no spill, deeper look-ahead)   [only where N % 192 == 0 and N % 256 == 0]
  4-wave    : 128-row tiles, two workgroups per CU
  240 CUs   : persistent grid capped at 240 workgroups (what DataParallel reserves for RCCL)
  trickle   : TIMING EXPERIMENT (garbage results): the GATE_RES epilogue's bytes issued one 16-byte op per phase INSIDE
              the K loop -- how much of the epilogue could hide under the MFMA work of the same CU
Interleaved rounds, best-of time per variant."""
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
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--decoder', action='store_true')
    ap.add_argument('--groups', action='store_true', help='sweep the tile-order group (nt8_group_m) instead of the kernel forms')
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    torch.manual_seed(0)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    M = args.m
    W_, Lr = 1152, 128
    shapes = [((M, 3 * W_, W_), 'BF16', 'qkv fwd'), ((M, W_, W_), 'GATE_RES', 'proj fwd'), ((M, 4 * W_, W_), 'GELU', 'fc1 fwd'),
              ((M, W_, 4 * W_), 'GATE_RES', 'fc2 fwd'), ((M, 4 * W_, W_), 'DGELU', 'fc2 dgrad'), ((M, W_, 4 * W_), 'BF16', 'fc1 dgrad'),
              ((M, W_, W_), 'BF16', 'proj dgrad'), ((M, W_, 3 * W_), 'BF16', 'qkv dgrad')]
    if args.decoder:
        Md = 2 * M
        shapes += [((Md, 1536, 512), 'BF16', 'dec qkv'), ((Md, 512, 512), 'GATE_RES', 'dec proj'), ((Md, 2048, 512), 'GELU', 'dec fc1'),
                   ((Md, 512, 2048), 'GATE_RES', 'dec fc2'), ((Md, 2048, 512), 'DGELU', 'dec fc2 dgrad'), ((Md, 512, 2048), 'BF16', 'dec fc1 dgrad')]
    variants = [('full', {}), ('no-epi', {'nt8_skip_epilogue': 1}), ('nf3', {'nt8_nf3': 1}), ('4-wave', {'gemm_nt_variant': 3}),
                ('240 CUs', {'nt8_max_cus': 240}), ('trickle', {'nt8_trickle': 1})]
    knobs = ['nt8_skip_epilogue', 'nt8_nf3', 'nt8_stagger', 'gemm_nt_variant', 'nt8_max_cus', 'nt8_trickle', 'nt8_group_m']
    if args.groups:
        variants = [('auto', {})] + [(f'g{g}', {'nt8_group_m': g}) for g in (1, 2, 3, 4, 6, 8, 12, 16)]
    print(f'{"shape / epilogue":40s} ' + ' '.join(f'{v[0]:>16s}' for v in variants) + '    (TFLOP/s | us)')
    for (m, n, k), name, tag in shapes:
        A = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
        Wt = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.randn(n, device=dev) * 0.1
        kw = dict(bias=b, epi=getattr(ops, 'EPI_' + name), out=torch.empty(m, n, device=dev, dtype=torch.bfloat16))
        if name == 'GATE_RES':
            kw.update(res=torch.randn(m, n, device=dev), gate=torch.randn(m // Lr, n, device=dev), gate_ld=n, rows_per_sample=Lr,
                      outf=torch.empty(m, n, device=dev))
        elif name == 'DGELU':
            kw.update(bias=None, aux=torch.randn(m, n, device=dev).to(torch.bfloat16), colsum=torch.zeros(n, device=dev))
        elif name == 'GELU':
            kw.update(out2=torch.empty(m, n, device=dev, dtype=torch.bfloat16))
        best = {}
        for r in range(args.rounds):
            for vname, kn in variants:
                if vname == 'nf3' and not (n % 192 == 0 and n % 256 == 0):
                    continue
                if vname == 'trickle' and not (name == 'GATE_RES' and n % 192 == 0):
                    continue  # the overlap experiment exists for the 256 x 192 GATE_RES tiles only
                for key in knobs:
                    L.mdt_set_tuning(key.encode(), kn.get(key, 0))
                ops.gemm_nt(A, Wt, **kw)
                L.mdt_event_record(ev[0], st)
                for _ in range(args.iters):
                    ops.gemm_nt(A, Wt, **kw)
                L.mdt_event_record(ev[1], st)
                torch.cuda.synchronize()
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                best[vname] = min(best.get(vname, 1e9), ms.value / args.iters)
        f = 2.0 * m * n * k
        cells = []
        for vname, _ in variants:
            cells.append(f'{f / best[vname] / 1e9:7.0f} |{best[vname] * 1e3:7.0f}' if vname in best else f'{"-":>16s}')
        print(f'{str((m, n, k)) + " " + name + " " + tag:40s} ' + ' '.join(f'{c:>16s}' for c in cells), flush=True)
        del A, Wt, kw
        torch.cuda.empty_cache()
    for key in knobs:
        L.mdt_set_tuning(key.encode(), 0)


if __name__ == '__main__':
    main()
