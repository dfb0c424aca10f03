"""Can the weight-gradient GEMMs run UNDER the HBM-bound kernels of the backward pass?  (run on the GPU box)

VERDICT r3 item 5(ii): the four `gemm_tn8` launches of a block (105 ms of a 460 ms step, matrix-bound, off the critical
path) next to the data-gradient chain (`gemm_nt8` K loops, but also ~2.5 ms per block of HBM-bound work: fused
epilogues, LayerNorm / gate backward, attention backward).  Every one of these kernels takes a CU's whole register file
(2 waves / SIMD x 256 VGPRs) and 112-148 KiB of its LDS, so two of them are never co-resident on a CU: the only way to
overlap them is to give each its own CUs.  This tool measures exactly that with HIP's CU-masked streams
(hipExtStreamCreateWithCUMask): ONE XL/2 encoder block's backward at the benchmarked size (131 072 token rows), four
blocks chained, as
    serial      everything on one stream, all CUs                                   (what the engine does)
    partition   data-gradient chain on a stream masked to (1 - f) of the CUs, the four weight-gradient GEMMs on a
                stream masked to f of them, ordered by events (a weight gradient starts when its operand is written)
for several f, with the persistent grids (`nt8_max_cus`) sized to each stream's CU count.  It first prints where the
masked streams' workgroups really run (tools/micro/cu_probe.hip: XCC / SE / CU of every workgroup).

    python tools/cu_partition_bench.py [--blocks 4] [--fracs 0.25,0.375,0.5]"""
import argparse
import collections
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from maskdit_amd import ops  # noqa: E402
from maskdit_amd._lib import lib  # noqa: E402

hip = C.CDLL('libamdhip64.so')
probe = C.CDLL(os.path.join(ROOT, 'tools', 'micro', 'libcu_probe.so'))


def masked_stream(bits):
    """stream restricted to the CUs whose mask bit is set (256 bits)"""
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask failed: {rc}')
    return torch.cuda.ExternalStream(s.value)


def where(stream, label):
    out = torch.zeros(4096, device='cuda', dtype=torch.int32)
    with torch.cuda.stream(stream):
        probe.cu_probe(C.c_void_p(out.data_ptr()), 4096, 40, C.c_void_p(stream.cuda_stream))
    torch.cuda.synchronize()
    v = out.cpu().tolist()
    per_xcc = collections.Counter((x >> 16) & 0xf for x in v)
    cus = {((x >> 16) & 0xf, (x >> 13) & 7, (x >> 12) & 1, (x >> 8) & 0xf) for x in v}
    print(f'  {label}: {len(cus)} distinct (xcc, se, sh, cu); workgroups per XCC {dict(sorted(per_xcc.items()))}', flush=True)
    return len(cus)


class Block:
    """operands of one XL/2 encoder block's backward at M token rows"""

    def __init__(self, M, B, L, W=1152, H=16):
        bf = torch.bfloat16
        r = lambda *s: (torch.randn(*s, device='cuda') * 0.05).to(bf)  # noqa: E731
        self.M, self.B, self.L, self.W, self.H = M, B, L, W, H
        self.dys, self.dys2 = r(M, W), torch.empty(M, W, device='cuda', dtype=bf)
        self.a, self.h = r(M, 4 * W), r(M, 4 * W)
        self.dh = torch.empty(M, 4 * W, device='cuda', dtype=bf)
        self.xn2, self.xn1, self.ao, self.ya = r(M, W), r(M, W), r(M, W), r(M, W)
        self.qkv = r(M, 3 * W)
        self.dxn = torch.empty(M, W, device='cuda', dtype=bf)
        self.dao = torch.empty(M, W, device='cuda', dtype=bf)
        self.xmid, self.xin = torch.randn(M, W, device='cuda'), torch.randn(M, W, device='cuda')
        self.dx = torch.randn(M, W, device='cuda') * 0.01
        self.mod, self.dmod = torch.randn(B, 6 * W, device='cuda') * 0.1, torch.zeros(B, 6 * W, device='cuda')
        self.st = torch.stack([torch.zeros(M, device='cuda'), torch.ones(M, device='cuda')], 1).contiguous()
        self.out, self.lse = ops.attn_fwd(self.qkv, B, L, H, W // H)
        self.W2T, self.W1T = r(4 * W, W), r(W, 4 * W)   # K-major shadows as the engine keeps them
        self.WpT, self.WqT = r(W, W), r(W, 3 * W)
        self.G2, self.G1 = torch.zeros(W, 4 * W, device='cuda'), torch.zeros(4 * W, W, device='cuda')
        self.Gp, self.Gq = torch.zeros(W, W, device='cuda'), torch.zeros(3 * W, W, device='cuda')
        self.gb1, self.gbq, self.gbp = torch.zeros(4 * W, device='cuda'), torch.zeros(3 * W, device='cuda'), torch.zeros(W, device='cuda')

    # ---- the data-gradient chain, in engine._block_bwd's order; `mark(k)` is called after operand k of a weight gradient exists
    def chain(self, mark):
        W, L = self.W, self.L
        m, d = self.mod, self.dmod
        ops.gemm_nt(self.dys, self.W2T, None, ops.EPI_DGELU, out=self.dh, aux=self.h, colsum=self.gb1)
        mark(1)
        ops.gemm_nt(self.dh, self.W1T, None, ops.EPI_BF16, out=self.dxn)
        C_ = ops.call
        C_('mdt_ln_modulate_bwd_gate', ops.p(self.dxn), ops.p(self.xmid), ops.p(self.st), m.data_ptr() + 4 * 4 * W, 6 * W, L, ops.p(self.dx), 1,
           d.data_ptr() + 4 * 3 * W, d.data_ptr() + 4 * 4 * W, 6 * W, self.M, W, ops.p(self.ya), m.data_ptr() + 4 * 2 * W, 6 * W,
           ops.p(self.dys2), d.data_ptr() + 4 * 2 * W, 6 * W, ops.p(self.gbp), ops.stream_ptr())
        mark(2)
        ops.gemm_nt(self.dys2, self.WpT, None, ops.EPI_BF16, out=self.dao)
        self.dqkv = ops.attn_bwd(self.qkv, self.out, self.dao, self.lse, self.B, L, self.H, W // self.H)
        mark(3)
        ops.gemm_nt(self.dqkv, self.WqT, None, ops.EPI_BF16, out=self.dxn)
        ops.ln_modulate_bwd(self.dxn, self.xin, self.st, m[:, W:], 6 * W, L, self.dx, True, d, d[:, W:], 6 * W)

    def wgrad(self, k):
        if k == 0:
            ops.gemm_tn(self.dys, self.a, self.G2)
        elif k == 1:
            ops.gemm_tn(self.dh, self.xn2, self.G1)
        elif k == 2:
            ops.gemm_tn(self.dys2, self.ao, self.Gp)
        else:
            ops.gemm_tn(self.dqkv, self.xn1, self.Gq, colsum_a=self.gbq)


def set_cus(n):
    lib().mdt_set_tuning(b'nt8_max_cus', n)


def run_serial(blocks):
    for b in blocks:
        set_cus(0)
        b.wgrad(0)
        b.chain(lambda k: b.wgrad(k))


def run_partition(blocks, sa, sb, na, nb):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    for b in blocks:
        def mark(k, b=b):
            ev = torch.cuda.Event()
            ev.record(sa)
            sb.wait_event(ev)
            with torch.cuda.stream(sb):
                set_cus(nb)
                b.wgrad(k)
            set_cus(na)
        with torch.cuda.stream(sb):
            set_cus(nb)
            b.wgrad(0)
        with torch.cuda.stream(sa):
            set_cus(na)
            b.chain(mark)
    cur.wait_stream(sa)
    cur.wait_stream(sb)
    set_cus(0)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--blocks', type=int, default=4)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--fracs', default='0.25,0.375,0.5')
    a = ap.parse_args()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    L = 128
    M = a.batch * L
    print(f'{ncu} CUs; one XL/2 encoder block backward at {M} rows, {a.blocks} blocks chained')
    full = masked_stream(range(ncu))
    where(full, 'mask = all bits')
    where(masked_stream(range(ncu // 2)), f'mask = bits 0..{ncu // 2 - 1} (contiguous half)')
    where(masked_stream(range(0, ncu, 2)), 'mask = even bits (strided half)')
    blocks = [Block(M, a.batch, L) for _ in range(a.blocks)]
    t_ser = timed(lambda: run_serial(blocks))
    print(f'serial, all CUs, one stream: {t_ser / a.blocks:.3f} ms per block')
    # the chain alone and the weight gradients alone (what each side costs on the whole chip)
    t_chain = timed(lambda: [b.chain(lambda k: None) for b in blocks])
    t_wg = timed(lambda: [b.wgrad(k) for b in blocks for k in range(4)])
    print(f'  data-gradient chain alone {t_chain / a.blocks:.3f} ms, weight gradients alone {t_wg / a.blocks:.3f} ms per block')
    for style in ('strided', 'contiguous'):
        for f in [float(x) for x in a.fracs.split(',')]:
            nb = int(round(ncu * f / 8)) * 8
            na = ncu - nb
            if style == 'contiguous':
                bits_b = list(range(na, ncu))
            else:  # every (ncu / nb)-th bit, so that whatever the bit -> CU mapping is, both sides get CUs of every XCD
                step = ncu / nb
                bits_b = sorted({int(i * step) for i in range(nb)})
            bits_a = [i for i in range(ncu) if i not in set(bits_b)]
            sa, sb = masked_stream(bits_a), masked_stream(bits_b)
            print(f'partition {style}: chain on {len(bits_a)} CUs, weight gradients on {len(bits_b)} CUs')
            ca, cb = where(sa, '  chain stream'), where(sb, '  wgrad stream')
            t = timed(lambda: run_partition(blocks, sa, sb, ca, cb))
            print(f'  -> {t / a.blocks:.3f} ms per block  ({t_ser / t:.3f}x vs serial)', flush=True)


if __name__ == '__main__':
    main()
