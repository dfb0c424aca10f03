"""Can RCCL run two ranks on ONE GPU (so that the nccl forms of the gradient exchange could finally execute on a 1-GPU box)?
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/lab/rccl_one_gpu_probe.py"""
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    x = torch.full((1024,), float(rank + 1), device='cuda')
    dist.all_reduce(x, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    buf = torch.arange(2048, device='cuda', dtype=torch.float32) + rank
    mine = buf[rank * 1024:(rank + 1) * 1024]
    dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    full = torch.zeros(2048, device='cuda')
    dist.all_gather_into_tensor(full, mine.clone())
    torch.cuda.synchronize()
    print(f'rank {rank}: all_reduce AVG -> {x[0].item()}, reduce_scatter AVG mine[0] -> {mine[0].item()}, all_gather ok {full[0].item()} {full[1024].item()}', flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f'rank {rank}: FAILED {type(e).__name__}: {str(e)[:400]}', flush=True)
    sys.exit(3)
