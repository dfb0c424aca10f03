"""A/B micro-benchmark + cross-check of the two NT GEMM kernels (run on the GPU box).
    python tools/gemm_bench.py [--iters 20]
For every shape: variant 1 (128x128 tile) vs variant 2 (256-row phase-pipelined), interleaved
rounds, median time -> TFLOP/s; outputs compared with each other and with torch fp32 matmul."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: F401,E402  (experiments build of the library: the timing switches are not in the product)
from maskdit_amd import _lib, ops  # noqa: E402

SHAPES = [  # (M, N, K, tag)
    (256, 256, 128, 'minimal nf4'), (512, 384, 256, 'small nf3'), (1024, 640, 384, 'small nf2'),
    (32768, 3456, 1152, 'XL qkv'), (32768, 1152, 1152, 'XL proj'), (32768, 4608, 1152, 'XL fc1'),
    (32768, 1152, 4608, 'XL fc2'), (65536, 1536, 512, 'dec qkv'), (65536, 512, 512, 'dec proj'),
    (65536, 2048, 512, 'dec fc1'), (65536, 512, 2048, 'dec fc2'), (8192, 8192, 8192, '8k cube'),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--tn', action='store_true')
    ap.add_argument('--ablate', action='store_true')
    ap.add_argument('--epi', action='store_true')
    ap.add_argument('--group', action='store_true')
    ap.add_argument('--small-m', action='store_true')
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    torch.manual_seed(0)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    print(f'{"shape":34s} {"v1 TF/s":>9s} {"v8 TF/s":>9s} {"ratio":>6s}  v8-v1 maxdiff   v8-ref relerr')
    for M, N, K, tag in SHAPES:
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        b = torch.randn(N, device=dev)
        outs = {}
        times = {1: [], 2: [], 3: []}
        for r in range(args.rounds):
            for v in (1, 2, 3):
                L.mdt_set_tuning(b'gemm_nt_variant', v)
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                ops.gemm_nt(A, W, b, ops.EPI_BF16, out=out)  # warm
                L.mdt_event_record(ev[0], st)
                for _ in range(args.iters):
                    ops.gemm_nt(A, W, b, ops.EPI_BF16, out=out)
                L.mdt_event_record(ev[1], st)
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                times[v].append(ms.value / args.iters)
                outs[v] = out
        tf = {v: 2.0 * M * N * K / (sorted(times[v])[len(times[v]) // 2] * 1e-3) / 1e12 for v in (1, 2, 3)}
        d = max((outs[2].float() - outs[1].float()).abs().max().item(), (outs[3].float() - outs[1].float()).abs().max().item())
        rows = min(M, 2048)
        ref = A[:rows].float() @ W.float().t() + b
        rel = ((outs[2][:rows].float() - ref).abs().max() / ref.abs().max()).item()
        print(f'{str((M, N, K)) + " " + tag:34s} {tf[1]:9.1f} {tf[2]:9.1f} {tf[2] / tf[1]:6.2f}  {d:12.3e}   {rel:10.3e}   4-wave: {tf[3]:7.1f}', flush=True)
    # fused epilogues on the pipelined kernel vs the 128x128 kernel (same arithmetic => same bits expected)
    M, N, K, Lr = 4096, 1152, 1152, 128
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.1
    res = torch.randn(M, N, device=dev)
    gate = torch.randn(M // Lr, 3 * N, device=dev)
    aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
    for name, kw in [('F32', dict(epi=ops.EPI_F32)), ('GELU', dict(epi=ops.EPI_GELU)), ('SILU', dict(epi=ops.EPI_SILU)),
                     ('GATE_RES', dict(epi=ops.EPI_GATE_RES, res=res, gate=gate[:, N:], gate_ld=3 * N, rows_per_sample=Lr)),
                     ('DGELU', dict(epi=ops.EPI_DGELU, aux=aux)), ('DSILU', dict(epi=ops.EPI_DSILU, aux=aux))]:
        got = {}
        for v in (1, 2):
            L.mdt_set_tuning(b'gemm_nt_variant', v)
            got[v] = ops.gemm_nt(A, W, b if 'D' != name[0] else None, **kw)
        worst = 0.0
        for x, y in zip(got[1], got[2]):
            if x is not None:
                worst = max(worst, (x.float() - y.float()).abs().max().item())
        print(f'epilogue {name:9s} v8 vs v1 max abs diff {worst:.3e}')
    L.mdt_set_tuning(b'gemm_nt_variant', 0)


if __name__ == '__main__' and not ({'--tn', '--ablate', '--epi', '--group', '--small-m'} & set(sys.argv)):
    main()


def tn_main(iters=10, rounds=3):
    """Weight-gradient GEMM C[N1,N2] += A[M,N1]^T B[M,N2]: variant 1 (128x128) vs auto (tn8)."""
    L = _lib.lib()
    dev = 'cuda'
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(4096, 512, 512, 'minimal'), (8192, 1152, 384, 'ragged x'), (4096, 384, 1152, 'swap ragged'), (4160, 640, 512, 'odd slots'),
              (32768, 1152, 1152, 'XL proj'), (32768, 3456, 1152, 'XL qkv'), (32768, 4608, 1152, 'XL fc1'),
              (32768, 1152, 4608, 'XL fc2'), (65536, 512, 512, 'dec proj'), (65536, 1536, 512, 'dec qkv'),
              (65536, 2048, 512, 'dec fc1'), (65536, 512, 2048, 'dec fc2')]
    print(f'{"TN shape (M,N1,N2)":34s} {"v1 TF/s":>9s} {"tn8 TF/s":>9s} {"ratio":>6s}  tn8-v1 relerr  tn8-ref relerr')
    for M, N1, N2, tag in shapes:
        A = (torch.rand(M, N1, device=dev) * 2 - 1).to(torch.bfloat16)
        B = (torch.rand(M, N2, device=dev) * 2 - 1).to(torch.bfloat16)
        outs, times = {}, {1: [], 0: []}
        for r in range(rounds):
            for v in (1, 0):
                L.mdt_set_tuning(b'gemm_tn_variant', v)
                Cc = torch.zeros(N1, N2, device=dev)
                ops.gemm_tn(A, B, Cc)
                outs[v] = Cc.clone()
                L.mdt_event_record(ev[0], st)
                for _ in range(iters):
                    ops.gemm_tn(A, B, Cc)
                L.mdt_event_record(ev[1], st)
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                times[v].append(ms.value / iters)
        tf = {v: 2.0 * M * N1 * N2 / (sorted(times[v])[len(times[v]) // 2] * 1e-3) / 1e12 for v in (1, 0)}
        ref = A.float().t() @ B.float()
        sc = ref.abs().max()
        d = ((outs[0] - outs[1]).abs().max() / sc).item()
        rel = ((outs[0] - ref).abs().max() / sc).item()
        print(f'{str((M, N1, N2)) + " " + tag:34s} {tf[1]:9.1f} {tf[0]:9.1f} {tf[0] / tf[1]:6.2f}  {d:12.3e}   {rel:10.3e}', flush=True)
    L.mdt_set_tuning(b'gemm_tn_variant', 0)


if __name__ == '__main__' and '--tn' in sys.argv:
    tn_main()


def ablate_main(iters=10):
    """nt8: full kernel vs main loop only (epilogue skipped) on the XL shapes -> per-tile fixed cost."""
    L = _lib.lib()
    dev = 'cuda'
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    L.mdt_set_tuning(b'gemm_nt_variant', 2)
    print(f'{"shape":30s} {"full TF/s":>10s} {"no-epi TF/s":>12s} {"full us":>9s} {"no-epi us":>10s}')
    for M, N, K in [(32768, 1152, 1152), (32768, 3456, 1152), (32768, 4608, 1152), (32768, 1152, 4608), (32768, 1152, 2304), (32768, 1152, 256)]:
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = {}
        for skip in (0, 1, 0, 1):
            L.mdt_set_tuning(b'nt8_skip_epilogue', skip)
            ops.gemm_nt(A, W, None, ops.EPI_BF16, out=out)
            L.mdt_event_record(ev[0], st)
            for _ in range(iters):
                ops.gemm_nt(A, W, None, ops.EPI_BF16, out=out)
            L.mdt_event_record(ev[1], st)
            ms = C.c_float()
            L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
            res[skip] = min(res.get(skip, 1e9), ms.value / iters)
        f = 2.0 * M * N * K
        print(f'{str((M, N, K)):30s} {f / res[0] / 1e9:10.1f} {f / res[1] / 1e9:12.1f} {res[0] * 1e3:9.1f} {res[1] * 1e3:10.1f}', flush=True)
    L.mdt_set_tuning(b'nt8_skip_epilogue', 0)
    L.mdt_set_tuning(b'gemm_nt_variant', 0)


if __name__ == '__main__' and '--ablate' in sys.argv:
    ablate_main()


def epi_main(iters=10):
    """8-wave (variant 2) vs 4-wave (variant 3) nt8 with the heavy fused epilogues of the training step."""
    L = _lib.lib()
    dev = 'cuda'
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    Lr = 128
    print(f'{"shape / epilogue":44s} {"8-wave TF/s":>12s} {"4-wave TF/s":>12s}')
    for (M, N, K), name in [((32768, 4608, 1152), 'GELU'), ((32768, 4608, 1152), 'DGELU'), ((32768, 1152, 1152), 'GATE_RES'),
                            ((32768, 1152, 4608), 'GATE_RES'), ((32768, 3456, 1152), 'BF16'), ((32768, 1152, 3456), 'BF16'),
                            ((32768, 1152, 4608), 'BF16'), ((65536, 2048, 512), 'GELU'), ((65536, 512, 2048), 'GATE_RES')]:
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device=dev) * 0.1
        kw = dict(bias=b, epi=getattr(ops, 'EPI_' + name))
        if name == 'GATE_RES':
            kw.update(res=torch.randn(M, N, device=dev), gate=torch.randn(M // Lr, N, device=dev), gate_ld=N, rows_per_sample=Lr,
                      out=torch.empty(M, N, device=dev, dtype=torch.bfloat16), outf=torch.empty(M, N, device=dev))
        elif name == 'DGELU':
            kw.update(bias=None, aux=torch.randn(M, N, device=dev).to(torch.bfloat16), out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        elif name == 'GELU':
            kw.update(out=torch.empty(M, N, device=dev, dtype=torch.bfloat16), out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        else:
            kw.update(out=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
        res = {}
        for v in (2, 3, 2, 3):
            L.mdt_set_tuning(b'gemm_nt_variant', v)
            L.mdt_set_tuning(b'nt8_stagger', 0)  # (a staggered start of every other workgroup was tried here: no gain)
            ops.gemm_nt(A, W, **kw)
            L.mdt_event_record(ev[0], st)
            for _ in range(iters):
                ops.gemm_nt(A, W, **kw)
            L.mdt_event_record(ev[1], st)
            ms = C.c_float()
            L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
            res[v] = min(res.get(v, 1e9), ms.value / iters)
        f = 2.0 * M * N * K
        print(f'{str((M, N, K)) + " " + name:44s} {f / res[2] / 1e9:12.1f} {f / res[3] / 1e9:12.1f}', flush=True)
    L.mdt_set_tuning(b'gemm_nt_variant', 0)
    L.mdt_set_tuning(b'nt8_stagger', 0)


if __name__ == '__main__' and '--epi' in sys.argv:
    epi_main()


def group_main(iters=10):
    """nt8 tile-order group height sweep (L2 reuse of the A row-panels vs the B column-panels)."""
    L = _lib.lib()
    dev = 'cuda'
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    gms = (2, 4, 8, 16, 32)
    print(f'{"shape":30s} ' + ' '.join(f'gm={g:<6d}' for g in gms))
    for M, N, K in [(32768, 1152, 1152), (32768, 3456, 1152), (32768, 4608, 1152), (32768, 1152, 4608), (65536, 2048, 512), (65536, 512, 2048)]:
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = {}
        for rep in range(2):
            for gm in gms:
                L.mdt_set_tuning(b'nt8_group_m', gm)
                ops.gemm_nt(A, W, None, ops.EPI_BF16, out=out)
                L.mdt_event_record(ev[0], st)
                for _ in range(iters):
                    ops.gemm_nt(A, W, None, ops.EPI_BF16, out=out)
                L.mdt_event_record(ev[1], st)
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                res[gm] = min(res.get(gm, 1e9), ms.value / iters)
        f = 2.0 * M * N * K
        print(f'{str((M, N, K)):30s} ' + ' '.join(f'{f / res[g] / 1e9:9.1f}' for g in gms), flush=True)
    L.mdt_set_tuning(b'nt8_group_m', 0)


if __name__ == '__main__' and '--group' in sys.argv:
    group_main()


def small_m_main(iters=10):
    """Per-GPU batch 128 (the 8-GPU data-parallel configuration): M = 16384 / 32768 -- tile-count quantisation."""
    L = _lib.lib()
    dev = 'cuda'
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream
    print(f'{"shape":30s} {"v1":>8s} {"8-wave":>8s} {"4-wave":>8s}   (TF/s)')
    for M, N, K in [(16384, 1152, 1152), (16384, 3456, 1152), (16384, 4608, 1152), (16384, 1152, 4608), (16384, 1152, 3456),
                    (32768, 512, 512), (32768, 1536, 512), (32768, 2048, 512), (32768, 512, 2048), (8192, 1152, 1152), (8192, 4608, 1152)]:
        A = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        res = {}
        for rep in range(2):
            for v in (1, 2, 3):
                L.mdt_set_tuning(b'gemm_nt_variant', v)
                ops.gemm_nt(A, W, None, ops.EPI_BF16, out=out)
                L.mdt_event_record(ev[0], st)
                for _ in range(iters):
                    ops.gemm_nt(A, W, None, ops.EPI_BF16, out=out)
                L.mdt_event_record(ev[1], st)
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                res[v] = min(res.get(v, 1e9), ms.value / iters)
        f = 2.0 * M * N * K
        print(f'{str((M, N, K)):30s} ' + ' '.join(f'{f / res[v] / 1e9:8.1f}' for v in (1, 2, 3)), flush=True)
    L.mdt_set_tuning(b'gemm_nt_variant', 0)


if __name__ == '__main__' and '--small-m' in sys.argv:
    small_m_main()
