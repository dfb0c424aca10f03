"""Fused AdamW + EMA + bf16 shadow kernel at the XL/2 parameter count.  python tools/adamw_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd._lib import call  # noqa: E402

n = 730_115_216
dev = 'cuda'
p, g, m, v, e = (torch.randn(n, device=dev) * 0.01 for _ in range(5))
v.abs_()
w16 = torch.empty(n, device=dev, dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream


def run():
    call('mdt_adamw_ema_step', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), e.data_ptr(), w16.data_ptr(), n, 1e-4, 0.9, 0.999,
         1e-8, 0.0, 0.1, 0.001, 0.9999, 1.0, st)


for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5
print(f'adamw_ema {t:.2f} ms  {38.0 * n / t / 1e9:.2f} TB/s')
