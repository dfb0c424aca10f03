"""World-size-1 RCCL group on the one GPU: do the call FORMS the data-parallel path uses pass RCCL's argument checks?
(ReduceOp.AVG, in-place reduce_scatter_tensor whose output is a view of its input, all_gather_into_tensor into an arena view, bf16)"""
import os

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29534')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
arena = torch.arange(4096, device='cuda', dtype=torch.float32)
ref = arena.clone()
w = dist.all_reduce(arena[128:1152], op=dist.ReduceOp.AVG, async_op=True); w.wait()
buf = arena[2048:3072]
w = dist.reduce_scatter_tensor(buf[0:1024], buf, op=dist.ReduceOp.AVG, async_op=True); w.wait()
w = dist.all_gather_into_tensor(arena[3072:4096], arena[3072:4096], async_op=True); w.wait()
stage = arena[:512].to(torch.bfloat16)
w = dist.all_reduce(stage, op=dist.ReduceOp.AVG, async_op=True); w.wait()
torch.cuda.synchronize()
print('single-rank RCCL: AVG all-reduce / in-place reduce-scatter / in-place all-gather / bf16 all accepted; arena unchanged:', bool(torch.equal(arena, ref)),
      '| NCCL version', torch.cuda.nccl.version())
dist.destroy_process_group()
