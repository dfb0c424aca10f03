"""Data-parallel wrapper: one process per GPU, full replica per rank, ONE exchange step -- the
mean all-reduce of the gradient arena (reference: accelerate/DDP around
`accelerator.backward`, train.py:178,220).

Not a translation of torch DDP's bucket/hook machinery: gradients already live in one flat
fp32 arena laid out in backward order (decoder blocks, then encoder blocks last-to-first, the
stacked adaLN weights, the embedders), so the backward plan simply announces each finished
*slab* (one transformer block = one contiguous range) and this wrapper launches an RCCL
all-reduce on it right away.  RCCL runs on its own stream: the collective of block i overlaps
the backward kernels of block i-1.  `finish()` (called by the optimizer step hook or
explicitly) waits for the outstanding collectives.  Slab size on XL/2: 23.9 M params
(~96 MB fp32) per encoder block -- large enough that xGMI link bandwidth, not launch latency,
bounds each collective.

Works with backend 'nccl' (= RCCL on ROCm) on GPUs and with 'gloo' for the CPU-side protocol
tests (tests/test_ddp_cpu.py), where a stand-in engine feeds it slabs.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class GradSlabReducer:
    """Averages ranges of a flat gradient tensor across the process group, asynchronously."""

    def __init__(self, process_group=None, bucket_elems: int = 0):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.pending: List = []
        self.flat: Optional[torch.Tensor] = None
        self.enabled = True  # False inside no_sync() (gradient accumulation micro-steps)
        self.reduced_elems = 0

    def attach(self, flat_grad: torch.Tensor):
        self.flat = flat_grad

    def reduce_range(self, name: str, lo: int, hi: int):
        if self.world == 1 or not self.enabled or hi <= lo:
            return
        chunk = self.flat[lo:hi]
        if self.backend == 'nccl':
            work = dist.all_reduce(chunk, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
            self.pending.append((work, None))
        else:  # gloo has no AVG
            work = dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self.pending.append((work, chunk))
        self.reduced_elems += hi - lo

    def finish(self):
        for work, chunk in self.pending:
            work.wait()
            if chunk is not None:
                chunk.div_(self.world)
        self.pending.clear()


class DataParallel(nn.Module):
    """`.module` holds the EDMPrecond (train_utils/loss.py:47 dereferences `net.module`).
    Construction broadcasts rank 0's parameter arena (what DDP does in `accelerator.prepare`,
    train.py:178).  Forward simply delegates; the loss object runs the fused path on
    `.module` and the gradient slabs flow through `GradSlabReducer`."""

    def __init__(self, module: nn.Module, process_group=None):
        super().__init__()
        self.module = module
        self.reducer = GradSlabReducer(process_group)
        eng = module.engine()
        if self.reducer.world > 1:
            dist.broadcast(eng.P, src=0, group=process_group)
            eng.shadows_dirty = True
        eng.ensure_grad()
        self.reducer.attach(eng.G)
        eng.grad_slab_hook = self.reducer.reduce_range

    @property
    def model(self):
        return self.module.model

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def finish_grad_sync(self):
        """Block the current stream until every slab has been averaged (call before optimizer.step)."""
        self.reducer.finish()

    class _NoSync:
        def __init__(self, red):
            self.red = red

        def __enter__(self):
            self.prev, self.red.enabled = self.red.enabled, False

        def __exit__(self, *a):
            self.red.enabled = self.prev

    def no_sync(self):
        """Gradient-accumulation micro-steps (train.py:211-215 `accelerator.accumulate`): skip
        the all-reduce; the final micro-step then reduces the accumulated arena slab by slab."""
        return DataParallel._NoSync(self.reducer)
