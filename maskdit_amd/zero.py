"""ZeRO-1 for the data-parallel path: optimizer state and the optimizer / EMA stream sharded over
the ranks.  reference: train.py:226-230 runs the full FusedAdam + update_ema on every replica.

Every trainable tensor, gradient, Adam moment and EMA value already lives in ONE flat arena of the same layout on
every rank (engine.py), so sharding is a matter of ranges, not of per-parameter bookkeeping:

  * every gradient slab (one transformer block, one adaLN row group ...) is split W ways and rank r owns piece r of
    EVERY slab (round 2: one contiguous 1/W of the arena per rank, which sent whole slabs to single owners);
  * backward: each finished slab goes out as ONE balanced `reduce_scatter_tensor` (GradSlabReducer.set_zero_sharding;
    half the bytes of an all-reduce, spread over all xGMI links, still overlapped with the remaining backward kernels);
  * step: the fused AdamW + EMA kernel runs on the owned range only -- 1/W of the 38 B/param optimizer stream and
    1/W of the moment memory (2 x 2.9 GB -> 0.73 GB per rank on XL/2 at W = 8);
  * the updated fp32 parameters are all-gathered in place into the parameter arena (one `all_gather_into_tensor`
    per slab: the other half of the all-reduce bytes), then every rank refreshes its bf16 / K-major GEMM shadows locally;
  * the EMA arena is only current on its owners until `sync_ema()` gathers it (before evaluation / checkpoints);
  * `consolidate()` (collective) gathers the moments, after which rank 0 alone can `state_dict()`: checkpoints keep
    the reference's (apex) layout and load into either optimizer.

The arithmetic per element is the same kernel as the unsharded optimizer: results are bit-identical to FusedAdam on
the same averaged gradient (tests/test_20_ddp_gpu.py::test_zero1_matches_unsharded).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from ._lib import call
from .ddp import _global_rank, slab_pieces
from .optim import FusedAdam, _st


def owned_pieces(slabs, world: int, rank: int) -> List[Tuple[int, int]]:
    """This rank's arena ranges: piece `rank` of every gradient slab (+ the < 8 W-element tails on the last rank),
    in arena order.  `slabs` = engine.lay.slabs (name -> (lo, hi)), which tile the arena."""
    out = []
    for lo, hi in sorted(slabs.values()):
        _, pieces, tail = slab_pieces(lo, hi, world)
        a, e = pieces[rank]
        if e > a:
            out.append((a, e))
        if rank == world - 1 and tail[1] > tail[0]:
            out.append(tail)
    return out


class ShardedFusedAdam(FusedAdam):
    """`FusedAdam` whose step touches only this rank's pieces of the arenas.  Use together with
    `DataParallel(net)`: pass the wrapper so that its reducer switches to reduce-scatter.

    Checkpoints: `state_dict()` needs the moments of every rank.  Gathering them is a COLLECTIVE, so it lives in
    `consolidate()`, which EVERY rank must call (it also brings the EMA arena up to date: `sync_ema()`); afterwards
    rank 0 alone may call `state_dict()` -- the reference's `if rank == 0: torch.save(...)` pattern (train.py:259-264).
    Calling `state_dict()` on a multi-rank group without a consolidation for the current step raises instead of
    dead-locking."""

    def __init__(self, params, data_parallel=None, process_group=None, **kw):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self._pieces: List[Tuple[int, int]] = []
        self._moff: List[int] = []
        self._consolidated = None  # (step, full_m, full_v)
        self._ema_stale = False
        super().__init__(params, **kw)
        if self._arena is None:
            raise ValueError('ShardedFusedAdam needs the complete trainable parameter set of one engine-bound model')
        if data_parallel is not None and self.world > 1:
            data_parallel.reducer.set_zero_sharding(True)

    # ---- layout: moments only for the owned pieces ------------------------------------------
    def _alloc_moments(self, eng):
        self._pieces = owned_pieces(eng.lay.slabs, self.world, self.rank)
        self._moff, n = [], 0
        for a, e in self._pieces:
            self._moff.append(n)
            n += e - a
        dev = eng.P.device
        self._m = torch.zeros(n, device=dev, dtype=torch.float32)
        self._v = torch.zeros(n, device=dev, dtype=torch.float32)

    def _step_mixed(self, group, hyp):
        raise NotImplementedError('ShardedFusedAdam: every trainable parameter needs a gradient (the moments are sharded by '
                                  'arena range, not by tensor); some .grad are None')

    def _step_arena(self, hyp):
        eng = self._arena
        G = eng.G
        if G is None:
            return
        ema_base, decay, ema_eng = None, 0.0, None
        if self._ema is not None:
            ema_eng = self._ema[0].engine()
            if ema_eng.lay.n != eng.lay.n:
                raise ValueError('fuse_ema: EMA model layout differs from the trained model')
            ema_base, decay = ema_eng.P.data_ptr(), self._ema[1]
        lr, b1, b2, eps, wd, bc1, bc2 = hyp
        for (a, e), mo in zip(self._pieces, self._moff):  # one launch per owned piece (one per slab: ~45 on XL/2)
            call('mdt_adamw_ema_step', eng.P.data_ptr() + 4 * a, G.data_ptr() + 4 * a, self._m.data_ptr() + 4 * mo,
                 self._v.data_ptr() + 4 * mo, (ema_base + 4 * a) if ema_base else None, None, e - a, lr, b1, b2, eps, wd, bc1, bc2,
                 decay, float(self.grad_scale), _st())
        self._gather(eng.P, eng.lay.slabs)
        eng.refresh_shadows(cast=True)  # bf16 shadow of the gathered pieces + K-major transposes + label table
        self._consolidated = None
        if ema_eng is not None:
            ema_eng.shadows_dirty = True
            self._ema_stale = True
            eng.ema_applied = (id(ema_eng), decay)

    def _gather(self, flat: torch.Tensor, slabs):
        """In-place all-gather of every rank's pieces of a flat arena: per slab ONE `all_gather_into_tensor` over RCCL
        (the equal pieces) + a broadcast of the tail from the last rank; over gloo (which moves CUDA tensors only by
        broadcast / all-reduce) one broadcast per (slab, owner)."""
        if self.world == 1:
            return
        works = []
        for lo, hi in sorted(slabs.values()):
            q, pieces, tail = slab_pieces(lo, hi, self.world)
            if q:
                if self.backend == 'nccl':
                    works.append(dist.all_gather_into_tensor(flat[lo:lo + self.world * q], flat[pieces[self.rank][0]:pieces[self.rank][1]],
                                                             group=self.pg, async_op=True))
                else:
                    for r, (a, e) in enumerate(pieces):
                        works.append(dist.broadcast(flat[a:e], src=_global_rank(self.pg, r), group=self.pg, async_op=True))
            if tail[1] > tail[0]:
                works.append(dist.broadcast(flat[tail[0]:tail[1]], src=_global_rank(self.pg, self.world - 1), group=self.pg, async_op=True))
        for w in works:
            w.wait()

    def sync_ema(self):
        """Make the EMA arena current on every rank (COLLECTIVE; call before sampling from / saving the EMA model)."""
        if self._ema is not None and self._ema_stale:
            self._gather(self._ema[0].engine().P, self._arena.lay.slabs)
            self._ema[0].engine().shadows_dirty = True
            self._ema_stale = False

    # ---- checkpoints keep the unsharded (apex) layout ------------------------------------------
    def _scatter_into(self, full: torch.Tensor, own: torch.Tensor):
        for (a, e), mo in zip(self._pieces, self._moff):
            full[a:e].copy_(own[mo:mo + e - a])

    def consolidate(self):
        """COLLECTIVE (every rank): gather the sharded moments into full arenas and bring the EMA up to date, so that
        `state_dict()` / `ema.state_dict()` can afterwards be called by rank 0 alone."""
        eng = self._arena
        full_m = torch.zeros(eng.lay.n, device=eng.P.device, dtype=torch.float32)
        full_v = torch.zeros_like(full_m)
        self._scatter_into(full_m, self._m)
        self._scatter_into(full_v, self._v)
        self._gather(full_m, eng.lay.slabs)
        self._gather(full_v, eng.lay.slabs)
        self.sync_ema()
        self._consolidated = (self.param_groups[0].get('step', 0), full_m, full_v)

    def state_dict(self):
        eng = self._arena
        if self.world > 1:
            if self._consolidated is None or self._consolidated[0] != self.param_groups[0].get('step', 0):
                raise RuntimeError('ShardedFusedAdam.state_dict(): the optimizer state is sharded over the ranks -- call '
                                   'consolidate() on EVERY rank first (then rank 0 alone may save)')
            _, full_m, full_v = self._consolidated
        else:
            full_m = torch.zeros(eng.lay.n, device=eng.P.device, dtype=torch.float32)
            full_v = torch.zeros_like(full_m)
            self._scatter_into(full_m, self._m)
            self._scatter_into(full_v, self._v)
        base = eng.P.data_ptr()
        saved = dict(self.state)
        try:
            for p in self.param_groups[0]['params']:
                if p.requires_grad:
                    off = (p.data_ptr() - base) // 4
                    self.state[p] = {'exp_avg': full_m[off:off + p.numel()].view_as(p), 'exp_avg_sq': full_v[off:off + p.numel()].view_as(p)}
            return super().state_dict()
        finally:
            self.state.clear()
            self.state.update(saved)

    def load_state_dict(self, state_dict):
        eng = self._arena
        full_m = torch.zeros(eng.lay.n, device=eng.P.device, dtype=torch.float32)
        full_v = torch.zeros_like(full_m)
        base = eng.P.data_ptr()
        for p in self.param_groups[0]['params']:
            if p.requires_grad:
                off = (p.data_ptr() - base) // 4
                self.state[p] = {'exp_avg': full_m[off:off + p.numel()].view_as(p), 'exp_avg_sq': full_v[off:off + p.numel()].view_as(p)}
        super().load_state_dict(state_dict)
        for (a, e), mo in zip(self._pieces, self._moff):
            self._m[mo:mo + e - a].copy_(full_m[a:e])
            self._v[mo:mo + e - a].copy_(full_v[a:e])
        self.state.clear()
        self._consolidated = None
