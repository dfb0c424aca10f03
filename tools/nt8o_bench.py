"""gemm_nt8o (wave-specialised NT GEMM: fused epilogue under the next tile's K loop) against the product gemm_nt8 (GPU box).
    python tools/nt8o_bench.py [--m 131072] [--iters 5] [--rounds 2] [--decoder] [--skip-parity]
Per (shape, epilogue):
  parity : every output of the overlap form bit-compared with the product kernel's (same inputs), abort code of the bounded spins
  timing : product full / product K loop only / overlap form, and for GATE_RES its decomposition (experiments library):
           no-epi-waves (dbg 3)  K loop + ring + y stores, epilogue waves idle
           kloop (35)            K loop + ring only (no y stores either)
           mfma (39)             ... without LDS-DMA (MFMA + fragment reads + counters)
           epi-no-load (9) / epi-no-store (17) / epi-y-only (25)   epilogue waves without their residual loads / stores / both
           no-ystore (33)        everything but the MMA waves' y stores
  stalls : (dbg 1) share of each role's wave-0 lifetime spent waiting on a counter (MMA: operands not landed; loader: ring full;
           epilogue: tile not published)."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: F401,E402
from maskdit_amd import _lib, ops  # noqa: E402

S_NAMES = {0: 'mma_wait_full', 3: 'loader_wait_empty', 5: 'epi_wait_ydone', 6: 'mma_total', 7: 'epi_total', 8: 'loader_total', 9: 'wgs'}


def report(L, reset=1):
    a = C.c_uint32(0)
    st = (C.c_uint64 * 16)()
    assert L.mdt_nt8o_report(C.byref(a), st, reset) == 0
    return a.value, list(st)


def make(m, n, k, name, dev, Lr):
    A = (torch.randn(m, k, device=dev) * 0.5).to(torch.bfloat16)
    Wt = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device=dev) * 0.1
    kw = dict(bias=b, epi=getattr(ops, 'EPI_' + name))
    if name == 'GATE_RES':
        kw.update(res=torch.randn(m, n, device=dev), gate=torch.randn(m // Lr, n, device=dev), gate_ld=n, rows_per_sample=Lr)
    return A, Wt, kw


def outputs(m, n, name, dev):
    o = dict(out=torch.zeros(m, n, device=dev, dtype=torch.bfloat16))
    if name == 'GATE_RES':
        o['outf'] = torch.zeros(m, n, device=dev)
    if name == 'GELU':
        o['out2'] = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=131072)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--decoder', action='store_true')
    ap.add_argument('--skip-parity', action='store_true')
    ap.add_argument('--small-only', action='store_true')
    args = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda'
    torch.manual_seed(0)
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream

    def tune(**kn):
        for key in ('nt8_skip_epilogue', 'nt8_overlap', 'nt8_max_cus'):
            assert L.mdt_set_tuning(key.encode(), kn.get(key, 0)) == 0, key

    # ---- parity at small sizes: multi-tile walks on a few CUs, uneven tile counts, every class, gate per 64-row half
    if not args.skip_parity:
        bad = 0
        for (m, n, k, name, Lr, cus) in [(1024, 256, 256, 'GATE_RES', 128, 0), (2048, 384, 512, 'GATE_RES', 64, 5), (4096, 1152, 1152, 'GATE_RES', 128, 24),
                                         (2048, 512, 256, 'GELU', 128, 3), (2304, 1152, 4608, 'GELU', 128, 0), (1536, 384, 320, 'BF16', 128, 4),
                                         (32768, 1152, 1152, 'GATE_RES', 128, 0), (32768, 4608, 1152, 'GELU', 128, 0), (32768, 3456, 1152, 'BF16', 128, 0),
                                         (32768, 1152, 4608, 'GATE_RES', 128, 0), (65536, 512, 2048, 'GATE_RES', 256, 0)]:
            if m % 256:
                continue
            A, Wt, kw = make(m, n, k, name, dev, Lr)
            ref, got = outputs(m, n, name, dev), outputs(m, n, name, dev)
            tune(nt8_max_cus=cus)
            ops.gemm_nt(A, Wt, **kw, **ref)
            for rep in range(4):
                tune(nt8_max_cus=cus, nt8_overlap=3 if rep < 2 else 7)  # two loader waves, then three
                for t in got.values():
                    t.fill_(7.0)
                ops.gemm_nt(A, Wt, **kw, **got)
                code, _ = report(L)
                diffs = {key: int((ref[key].view(torch.int16 if ref[key].dtype == torch.bfloat16 else torch.int32) !=
                                   got[key].view(torch.int16 if got[key].dtype == torch.bfloat16 else torch.int32)).sum().item()) for key in ref}
                ok = code == 0 and not any(diffs.values())
                bad += not ok
                print(f'parity {(m, n, k)} {name:8s} Lr {Lr:3d} cus {cus:3d} rep {rep}: abort {code} mismatching elements {diffs} {"OK" if ok else "FAIL"}', flush=True)
                if not ok:
                    break
            del A, Wt, kw, ref, got
            torch.cuda.empty_cache()
        tune()
        print(f'parity: {"ALL BIT-IDENTICAL" if bad == 0 else str(bad) + " FAILURES"}', flush=True)
    if args.small_only:
        return

    M = args.m
    W_, Lr = 1152, 128
    shapes = [((M, W_, W_), 'GATE_RES', 'proj fwd'), ((M, W_, 4 * W_), 'GATE_RES', 'fc2 fwd'), ((M, 4 * W_, W_), 'GELU', 'fc1 fwd'),
              ((M, 3 * W_, W_), 'BF16', 'qkv fwd'), ((M, W_, 4 * W_), 'BF16', 'fc1 dgrad'), ((M, W_, W_), 'BF16', 'proj dgrad'),
              ((M, W_, 3 * W_), 'BF16', 'qkv dgrad')]
    if args.decoder:
        Md = 2 * M
        shapes += [((Md, 512, 512), 'GATE_RES', 'dec proj'), ((Md, 512, 2048), 'GATE_RES', 'dec fc2'), ((Md, 2048, 512), 'GELU', 'dec fc1'),
                   ((Md, 1536, 512), 'BF16', 'dec qkv'), ((Md, 512, 2048), 'BF16', 'dec fc1 dgrad')]
    gate_dbg = [('no-epi-waves', 3), ('kloop', 35), ('mfma', 39), ('epi-no-load', 9), ('epi-no-store', 17), ('epi-y-only', 25), ('no-ystore', 33)]
    print(f'{"shape / epilogue":42s} {"product":>14s} {"product no-epi":>14s} {"OVERLAP":>14s}   decomposition of the overlap form (TFLOP/s | us)')
    for (m, n, k), name, tag in shapes:
        A, Wt, kw = make(m, n, k, name, dev, Lr if 'dec' not in tag else 256)
        o = outputs(m, n, name, dev)
        variants = [('product', dict()), ('product no-epi', dict(nt8_skip_epilogue=1)), ('OVERLAP', dict(nt8_overlap=3))]
        if name != 'BF16':
            variants += [('nl3', dict(nt8_overlap=7))]
        if name == 'GATE_RES':
            variants += [(nm, dict(nt8_overlap=3 | (d << 4))) for nm, d in gate_dbg]
            variants += [('nl3:' + nm, dict(nt8_overlap=7 | (d << 4))) for nm, d in gate_dbg[:3]]
        best = {}
        for r in range(args.rounds):
            for vname, kn in variants:
                tune(**kn)
                ops.gemm_nt(A, Wt, **kw, **o)
                L.mdt_event_record(ev[0], st)
                for _ in range(args.iters):
                    ops.gemm_nt(A, Wt, **kw, **o)
                L.mdt_event_record(ev[1], st)
                torch.cuda.synchronize()
                ms = C.c_float()
                L.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
                best[vname] = min(best.get(vname, 1e9), ms.value / args.iters)
        code, _ = report(L)
        f = 2.0 * m * n * k
        cells = [f'{f / best[v] / 1e9:6.0f} |{best[v] * 1e3:6.0f}' for v, _ in variants[:3]]
        extra = '  '.join(f'{v} {f / best[v] / 1e9:.0f}|{best[v] * 1e3:.0f}' for v, _ in variants[3:])
        print(f'{str((m, n, k)) + " " + name + " " + tag:42s} ' + ' '.join(f'{c:>14s}' for c in cells) + f'   {extra}' + (f'   ABORT {code}' if code else ''), flush=True)
        # stall shares (STATS instantiation)
        for ovb in ((3,) if name == 'BF16' else (3, 7)):
          tune(nt8_overlap=ovb | (1 << 4))
          report(L)
          ops.gemm_nt(A, Wt, **kw, **o)
          code, s = report(L)
          if s[9]:
            w = s[9]
            print(f'{"":38s}nl{2 if ovb == 3 else 3}   stalls: MMA waits for operands {s[0] / max(s[6], 1):.3f} of its {s[6] / w:.0f} clocks; loader waits for a free stage '
                  f'{s[3] / max(s[8], 1):.3f} of {s[8] / w:.0f}; epilogue waits for the tile {s[5] / max(s[7], 1):.3f} of {s[7] / max(w, 1):.0f}'
                  + (f'   ABORT {code}' if code else ''), flush=True)
          if name == 'GATE_RES' and ovb == 3 and 'proj' in tag or (name == 'GATE_RES' and ovb == 3 and 'fc2 fwd' == tag):
            stp = (C.c_uint64 * 192)()
            assert L.mdt_nt8o_stamps(stp) == 0
            t0 = stp[0]
            rows = []
            for T in range(2, 8):  # steady-state tiles of workgroup 0
                rel = [int(stp[(r * 32 + T) * 2 + e]) - int(t0) for r in range(3) for e in range(2)]
                rows.append(f'tile {T}: MMA K loop [{rel[0]:7d}, {rel[1]:7d}]  loader issues [{rel[2]:7d}, {rel[3]:7d}]  epilogue [{rel[4]:7d}, {rel[5]:7d}]')
            print(f'{"":38s}nl2   time line of workgroup 0 (shader clocks from its first MFMA phase; the epilogue of tile T runs under the K loop of tile T + 1):')
            for r_ in rows:
                print(f'{"":44s}{r_}', flush=True)
        del A, Wt, kw, o
        torch.cuda.empty_cache()
    tune()


if __name__ == '__main__':
    main()
