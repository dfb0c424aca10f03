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


def _global_rank(pg, r: int) -> int:
    """torch.distributed's src / dst arguments are GLOBAL ranks; `r` is a rank inside `pg`."""
    return r if pg is None else dist.get_global_rank(pg, r)


def slab_pieces(lo: int, hi: int, world: int):
    """ZeRO-1 ownership of one gradient slab [lo, hi): W equal 8-element-aligned pieces (rank r owns piece r) plus a
    tail shorter than 8 W elements that stays with the last rank.  Returns (q, [(a_0, e_0), ..., (a_{W-1}, e_{W-1})],
    (tail_lo, tail_hi)); q = elements per equal piece (what reduce_scatter / all_gather move)."""
    q = ((hi - lo) // world) // 8 * 8
    pieces = [(lo + r * q, lo + (r + 1) * q) for r in range(world)]
    return q, pieces, (lo + world * q, hi)


class GradSlabReducer:
    """Averages ranges of a flat gradient tensor across the process group, asynchronously.

    Two exchange forms, both launched slab by slab from the backward plan:
      * all-reduce (default): every rank ends with the mean of the slab;
      * reduce-scatter (`set_zero_sharding()`, ZeRO-1): every slab is split W ways and rank r receives the mean of ITS
        piece only -- ONE balanced `reduce_scatter_tensor` per slab (half the link bytes of an all-reduce, spread over
        all links; round 2 sent whole slabs to single owners).  The parameter `all_gather_into_tensor` after the sharded
        optimizer step is the other half (maskdit_amd/zero.py).
    `wire_dtype=torch.bfloat16`: the slab is cast into a bf16 staging arena, exchanged in bf16 (half the bytes: 1.46
    instead of 2.92 GB per step on XL/2) and accumulated back into the fp32 arena after the wait.
    Backends: 'nccl' (= RCCL; AVG reduction, in-place reduce-scatter) and 'gloo' (CPU protocol tests, and two ranks on
    one GPU: gloo moves CUDA tensors only through all_reduce / broadcast, so the reduce-scatter is an all-reduce whose
    foreign pieces are simply not used)."""

    def __init__(self, process_group=None, bucket_elems: int = 0, wire_dtype=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.pending: List = []
        self.flat: Optional[torch.Tensor] = None
        self.enabled = True  # False inside no_sync() (gradient accumulation micro-steps)
        self.reduced_elems = 0
        self.wire_bytes = 0   # bytes handed to the collectives since the last reset (tests / logs)
        self.zero = False
        self.plain_only = False   # True when the nccl group agreed to run without AVG / the in-place reduce-scatter
        self.use_avg = self.use_scatter = False  # collective forms, agreed by all ranks in _agree_forms()
        self.forms_agreed = False
        self.wire_dtype = wire_dtype if wire_dtype not in (None, torch.float32) else None
        self.stage: Optional[torch.Tensor] = None

    def attach(self, flat_grad: torch.Tensor):
        self.flat = flat_grad
        if self.wire_dtype is not None and self.world > 1:
            self.stage = torch.empty(flat_grad.numel(), device=flat_grad.device, dtype=self.wire_dtype)
        if dist.is_initialized() and not self.forms_agreed:  # (also on a one-rank group: the GPU suite runs the RCCL forms that way)
            self._agree_forms(flat_grad.device, self.wire_dtype or flat_grad.dtype)

    def set_zero_sharding(self, on: bool = True):
        self.zero = bool(on)

    # ---- which collective forms this group runs: decided ONCE, by every rank together --------------------------
    def _agree_forms(self, device, dtype):
        """The RCCL forms of the exchange (AVG reduction, in-place reduce-scatter on views of one buffer) are probed HERE,
        once, on a tiny tensor, and the outcome is agreed across the group with a plain all_reduce(SUM) of the success
        flags: a form is used only if EVERY rank's probe accepted it.  After this point `_launch` catches nothing --
        a rank-local failure in the middle of training (out of memory, an aborted communicator, a watchdog timeout are
        all RuntimeErrors) must end the job, not switch ONE rank to a different collective than its peers are in
        (rounds 1-5 degraded per rank inside `_launch`, which could pair all_reduce(SUM) on one rank with
        reduce_scatter(AVG) on the others: a hang or silently wrong gradients).  gloo has neither form: SUM + divide."""
        self.use_avg = self.use_scatter = False
        if self.backend == 'nccl':
            flags = [1, 1]
            probe = torch.ones(8 * self.world, device=device, dtype=dtype)
            try:
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=self.pg)
            except (ValueError, NotImplementedError, TypeError, RuntimeError) as e:  # argument-level refusal, same on every rank
                flags[0] = 0
                self._note(f'all_reduce(AVG) refused by the backend ({type(e).__name__}: {e})')
            try:
                dist.reduce_scatter_tensor(probe[self.rank * 8:(self.rank + 1) * 8], probe, op=dist.ReduceOp.AVG if flags[0] else dist.ReduceOp.SUM,
                                           group=self.pg)
            except (ValueError, NotImplementedError, TypeError, RuntimeError) as e:
                flags[1] = 0
                self._note(f'in-place reduce_scatter_tensor refused by the backend ({type(e).__name__}: {e})')
            agreed = torch.tensor(flags, device=device, dtype=torch.int32)
            dist.all_reduce(agreed, op=dist.ReduceOp.SUM, group=self.pg)  # the plainest form: what the fallback itself uses
            ok = [int(v) == self.world for v in agreed.tolist()]
            if ok != [bool(f) for f in flags]:
                self._note(f'a peer refused a collective form this rank accepted (local {flags}, agreed {ok})')
            self.use_avg, self.use_scatter = ok
        self.plain_only = self.backend == 'nccl' and not (self.use_avg and self.use_scatter)
        self.forms_agreed = True

    def _note(self, why: str):
        import warnings
        warnings.warn(f'GradSlabReducer: {why}; that form stays off for the whole run (all_reduce(SUM) + divide instead)',
                      RuntimeWarning)

    # ---- one collective on one contiguous range -------------------------------------------------------------
    def _launch(self, lo: int, hi: int, scatter_q: int = 0):
        """all-reduce [lo, hi) (scatter_q = 0) or reduce-scatter it in W pieces of scatter_q elements.  Exceptions
        propagate: the forms were agreed in `_agree_forms`, and a failure here is rank-local by construction."""
        if hi <= lo:
            return
        avg = self.use_avg
        op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM  # gloo has no AVG: divide after the wait
        if self.stage is not None:
            buf = self.stage[lo:hi]
            buf.copy_(self.flat[lo:hi])  # fp32 -> wire dtype on the compute stream; the collective is ordered behind it
        else:
            buf = self.flat[lo:hi]
        if scatter_q and self.use_scatter:
            mine = buf[self.rank * scatter_q:(self.rank + 1) * scatter_q]
            work = dist.reduce_scatter_tensor(mine, buf, op=op, group=self.pg, async_op=True)
            keep = (lo + self.rank * scatter_q, lo + (self.rank + 1) * scatter_q)
        else:
            work = dist.all_reduce(buf, op=op, group=self.pg, async_op=True)
            keep = (lo, hi)
        self.wire_bytes += buf.numel() * buf.element_size()
        self.pending.append((work, keep, not avg))

    def reduce_range(self, name: str, lo: int, hi: int):
        if self.world == 1 or not self.enabled or hi <= lo:
            return
        if not self.zero:
            self._launch(lo, hi)
        else:
            q, _, tail = slab_pieces(lo, hi, self.world)
            if q:
                self._launch(lo, lo + self.world * q, scatter_q=q)
            self._launch(tail[0], tail[1])  # < 8 W elements: plain all-reduce
        self.reduced_elems += hi - lo

    def finish(self):
        for work, (a, e), divide in self.pending:
            work.wait()
            if self.stage is not None:
                self.flat[a:e].copy_(self.stage[a:e])  # back to fp32 (the optimizer reads the fp32 arena)
            if divide:
                self.flat[a:e].div_(self.world)
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
    (reserve_cus_for_collectives).  `grad_wire_dtype=torch.bfloat16`: gradient slabs travel as bf16 (config key
    `train.grad_wire: bf16`)."""

    def __init__(self, module: nn.Module, process_group=None, rccl_cus=None, grad_wire_dtype=None):
        super().__init__()
        self.module = module
        self.reducer = GradSlabReducer(process_group, wire_dtype=grad_wire_dtype)
        if self.reducer.world > 1 and self.reducer.backend == 'nccl':
            import os
            self.reserved_cus = int(os.environ.get('MDT_RCCL_CUS', '16')) if rccl_cus is None else int(rccl_cus)
            self.gemm_cus = reserve_cus_for_collectives(self.reserved_cus)
        eng = module.engine()
        if self.reducer.world > 1:
            dist.broadcast(eng.P, src=_global_rank(process_group, 0), group=process_group)
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
