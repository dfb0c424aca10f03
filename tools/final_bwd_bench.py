import sys, torch
sys.path.insert(0, '/root/repo')
import maskdit_amd as M
from maskdit_amd import _lib
from maskdit_amd._lib import call
L_ = _lib.lib()
B, T, Dd, C, p = 1024, 256, 512, 4, 2
O = p * p * C
R = 32
dF = torch.randn(B, C, R, R, device='cuda')
x = torch.randn(B * T, Dd, device='cuda')
stats = torch.stack([x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()], 1).contiguous()
mod = torch.randn(B, 2 * Dd, device='cuda')
W = torch.randn(O, Dd, device='cuda') * 0.05
dx = torch.empty_like(x); dW = torch.zeros_like(W); db = torch.zeros(O, device='cuda'); dmod = torch.zeros_like(mod)
st = torch.cuda.current_stream().cuda_stream
def run():
    call('mdt_final_bwd', dF.data_ptr(), x.data_ptr(), stats.data_ptr(), mod.data_ptr(), mod[:, Dd:].data_ptr(), 2 * Dd, W.data_ptr(), dx.data_ptr(),
         dW.data_ptr(), db.data_ptr(), dmod.data_ptr(), dmod[:, Dd:].data_ptr(), 2 * Dd, B, T, Dd, C, p, st)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f'final_bwd {e0.elapsed_time(e1) * 100:.1f} us')
