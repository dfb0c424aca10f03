"""Times the attention kernels at the training shapes (run on the GPU box).
    python tools/attn_bench.py [--iters 20]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import _lib, ops  # noqa: E402


def main(iters=20):
    L_ = _lib.lib()
    dev = 'cuda'
    ev = [C.c_void_p() for _ in range(2)]
    for e in ev:
        L_.mdt_event_create(C.byref(e))
    st = torch.cuda.current_stream().cuda_stream

    def timeit(fn):
        fn()
        L_.mdt_event_record(ev[0], st)
        for _ in range(iters):
            fn()
        L_.mdt_event_record(ev[1], st)
        ms = C.c_float()
        L_.mdt_event_elapsed_ms(ev[0], ev[1], C.byref(ms))
        return ms.value / iters * 1e3

    print(f'{"shape (B,L,H,hd)":26s} {"variant":>10s} {"fwd us":>8s} {"bwd us":>8s} {"fwd GB/s":>9s} {"bwd GB/s":>9s} {"fwd TF/s":>9s}')
    # the benchmarked shapes: encoder (B 1024, L 128, hd 72) and decoder (B 1024, T 256, hd 32) of XL/2 at 256^2
    for B, L, H, hd in [(1024, 128, 16, 72), (1024, 256, 16, 32), (64, 512, 16, 72), (128, 256, 16, 72), (256, 1024, 16, 32)]:
        qkv = (torch.randn(B * L, 3 * H * hd, device=dev) * 0.5).to(torch.bfloat16)
        for sp, name in ((1, 'block-loop'), (0, 'default'), (2, 'sp OCC=4')):
            if (sp != 1 and L not in (128, 256, 512, 1024)) or (sp == 2 and L > 256):
                continue
            L_.mdt_set_tuning(b'attn_sp', sp)
            out, lse = ops.attn_fwd(qkv, B, L, H, hd)
            dout = torch.randn_like(out)
            tf = timeit(lambda: ops.attn_fwd(qkv, B, L, H, hd))
            tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, hd))
            by_f = qkv.numel() * 2 + out.numel() * 2
            by_b = qkv.numel() * 2 * 2 + out.numel() * 2 * 2
            fl = 4.0 * B * H * L * L * hd
            print(f'{str((B, L, H, hd)):26s} {name:>10s} {tf:8.1f} {tb:8.1f} {by_f / tf / 1e3:9.0f} {by_b / tb / 1e3:9.0f} {fl / tf / 1e6:9.1f}', flush=True)
    L_.mdt_set_tuning(b'attn_sp', 0)


if __name__ == '__main__':
    main()
