import sys, torch
sys.path.insert(0, '/root/repo')
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
B, L, H, hd = 1024, 128, 16, 72
qkv = (torch.randn(B * L, 3 * H * hd, device='cuda') * 0.5).to(torch.bfloat16)
out, lse = ops.attn_fwd(qkv, B, L, H, hd)
dout = torch.randn_like(out)
res = {}
for knob, name in [(3, 'register prefetch (attn_sp=3)'), (0, 'LDS-DMA double buffer (default)'), (1, 'block-loop kernels')]:
    L_.mdt_set_tuning(b'attn_sp', knob)
    d = ops.attn_bwd(qkv, out, dout, lse, B, L, H, hd)
    res[knob] = d.float()
    print(f'{name:36s} {t_us(lambda: ops.attn_bwd(qkv, out, dout, lse, B, L, H, hd)):8.1f} us', flush=True)
L_.mdt_set_tuning(b'attn_sp', 0)
print('max |dma - regs| =', (res[0] - res[3]).abs().max().item(), ' max |dma - block| =', (res[0] - res[1]).abs().max().item(), ' scale', res[1].abs().max().item())
