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
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.pending: List = []
        self.flat: Optional[torch.Tensor] = None
        self.enabled = True  # False inside no_sync() (gradient accumulation micro-steps)
        self.reduced_elems = 0
        self.shard_bounds: Optional[List[int]] = None  # ZeRO-1: rank r owns arena elements [b[r], b[r+1])

    def attach(self, flat_grad: torch.Tensor):
        self.flat = flat_grad

    def set_owner_shards(self, bounds: List[int]):
        """ZeRO-1 (maskdit_amd/zero.py): every gradient range is reduced TO THE RANK THAT OWNS ITS OPTIMIZER STATE
        instead of all-reduced -- half the bytes on the links; the other half is the all-gather of the updated
        parameters after the sharded optimizer step."""
        assert len(bounds) == self.world + 1 and bounds[0] == 0 and all(a <= b for a, b in zip(bounds, bounds[1:]))
        self.shard_bounds = list(bounds)

    def _reduce_piece(self, chunk: torch.Tensor, dst: Optional[int]):
        avg = self.backend == 'nccl'
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM  # gloo has no AVG: divide after the wait
        if dst is None:
            work = dist.all_reduce(chunk, op=op, group=self.pg, async_op=True)
            self.pending.append((work, None if avg else chunk))
        else:
            work = dist.reduce(chunk, dst=dst, op=op, group=self.pg, async_op=True)
            self.pending.append((work, chunk if (not avg and dst == self.rank) else None))

    def reduce_range(self, name: str, lo: int, hi: int):
        if self.world == 1 or not self.enabled or hi <= lo:
            return
        if self.shard_bounds is None:
            self._reduce_piece(self.flat[lo:hi], None)
        else:  # split the slab at the ownership boundaries: one reduce per owner
            b = self.shard_bounds
            for r in range(self.world):
                a, e = max(lo, b[r]), min(hi, b[r + 1])
                if a < e:
                    self._reduce_piece(self.flat[a:e], r)
        self.reduced_elems += hi - lo

    def finish(self):
        for work, chunk in self.pending:
            work.wait()
            if chunk is not None:
                chunk.div_(self.world)
        self.pending.clear()


def reserve_cus_for_collectives(n_reserved: int) -> int:
    """The persistent GEMM kernels launch exactly one workgroup per CU and each takes the CU's whole register file
    and 112-128 KiB of its LDS: an RCCL all-reduce kernel that becomes resident in the middle of the backward pass
    would leave some of those persistent workgroups waiting for a CU (every GEMM launch would then take two
    rounds), and conversely RCCL could not start while a GEMM holds every CU.  With data parallelism active the
    GEMM grids are therefore capped at (CUs - n_reserved) (`mdt_set_tuning("nt8_max_cus")`), which leaves
    `n_reserved` CUs to the collective for the whole backward.  Cost of the reservation on one GPU: tools/nt8_bench.py
    column "240 CUs" (GEMM time scales with 256 / (256 - n_reserved))."""
    from . import _lib
    import torch
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    cap = max(cus - max(int(n_reserved), 0), cus // 2)
    _lib.check(_lib.lib().mdt_set_tuning(b'nt8_max_cus', cap if n_reserved > 0 else 0), 'mdt_set_tuning')
    return cap


class DataParallel(nn.Module):
    """`.module` holds the EDMPrecond (train_utils/loss.py:47 dereferences `net.module`).
    Construction broadcasts rank 0's parameter arena (what DDP does in `accelerator.prepare`,
    train.py:178).  Forward simply delegates; the loss object runs the fused path on
    `.module` and the gradient slabs flow through `GradSlabReducer`.

    `rccl_cus` (default: env MDT_RCCL_CUS or 16; only with the nccl backend and world > 1): CUs kept free of the
    persistent GEMM workgroups so that the slab all-reduces really run under the backward kernels
    (reserve_cus_for_collectives)."""

    def __init__(self, module: nn.Module, process_group=None, rccl_cus=None):
        super().__init__()
        self.module = module
        self.reducer = GradSlabReducer(process_group)
        if self.reducer.world > 1 and self.reducer.backend == 'nccl':
            import os
            self.reserved_cus = int(os.environ.get('MDT_RCCL_CUS', '16')) if rccl_cus is None else int(rccl_cus)
            self.gemm_cus = reserve_cus_for_collectives(self.reserved_cus)
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
