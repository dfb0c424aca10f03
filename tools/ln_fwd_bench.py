import sys, torch
sys.path.insert(0, '/root/repo')
from maskdit_amd import ops
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for B, L, D in [(1024, 128, 1152), (1024, 256, 512)]:
    x = torch.randn(B * L, D, device='cuda')
    mod = torch.randn(B, 3 * D, device='cuda')
    t = t_us(lambda: ops.ln_modulate_fwd(x, mod[:, :D], mod[:, 2 * D:], 3 * D, L))
    print(f'ln_modulate_fwd {B}x{L}x{D}: {t:7.1f} us  {6.0 * B * L * D / t / 1e6:5.2f} TB/s')
