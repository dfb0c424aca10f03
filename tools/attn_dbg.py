import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _explib  # noqa: F401,E402  (experiments build of the library: attn_dbg is not in the product)
from maskdit_amd import _lib, ops
L_ = _lib.lib()
def t_us(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, L, H, hd in [(1024, 128, 16, 72), (1024, 256, 16, 32)]:
    qkv = (torch.randn(B * L, 3 * H * hd, device='cuda') * 0.5).to(torch.bfloat16)
    out, lse = ops.attn_fwd(qkv, B, L, H, hd)
    dout = torch.randn_like(out)
    for d, name in [(0, 'full'), (1, 'no stores'), (2, 'no prefetch of next item'), (3, 'no stores, no fetch'), (4, 'no dK/dV phase'), (8, 'no dQ phase'), (12, 'no compute'), (15, 'nothing but first fetch + LDS staging')]:
        L_.mdt_set_tuning(b'attn_dbg', d)
        print(f'{(B, L, H, hd)} {name:40s} {t_us(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, hd)):8.1f} us', flush=True)
    L_.mdt_set_tuning(b'attn_dbg', 0)
