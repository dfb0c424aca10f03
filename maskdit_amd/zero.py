"""ZeRO-1 for the data-parallel path: optimizer state and the optimizer / EMA stream sharded over
the ranks.  reference: train.py:226-230 runs the full FusedAdam + update_ema on every replica.

Every trainable tensor, gradient, Adam moment and EMA value already lives in ONE flat arena of the same layout on
every rank (engine.py), so sharding is a matter of ranges, not of per-parameter bookkeeping:

  * rank r owns arena elements [b_r, b_{r+1}) (equal 8-element-aligned ranges);
  * backward: each finished gradient slab is reduced TO ITS OWNER(S) (GradSlabReducer.set_owner_shards; half the
    bytes of an all-reduce, still overlapped with the remaining backward kernels);
  * step: the fused AdamW + EMA kernel runs on the owned range only -- 1/W of the 38 B/param optimizer stream and
    1/W of the moment memory (2 x 2.9 GB -> 0.73 GB per rank on XL/2 at W = 8);
  * the updated fp32 parameters are all-gathered in place into the parameter arena (one broadcast per owner: the
    other half of the all-reduce bytes), then every rank refreshes its bf16 / K-major GEMM shadows locally;
  * the EMA arena is only current on its owner until `sync_ema()` gathers it (before evaluation / checkpoints);
  * `state_dict()` gathers the moments, so checkpoints keep the reference's (apex) layout and load into either
    optimizer.

The arithmetic per element is the same kernel as the unsharded optimizer: results are bit-identical to FusedAdam on
the same averaged gradient (tests/test_ddp_gpu.py::test_zero1_matches_unsharded).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from ._lib import call
from .optim import FusedAdam, _st


def shard_bounds(n: int, world: int) -> List[int]:
    """Equal ranges aligned to 8 elements (32 bytes: the vector width of every arena kernel)."""
    per = (n + world - 1) // world
    per = (per + 7) // 8 * 8
    return [min(r * per, n) for r in range(world + 1)]


class ShardedFusedAdam(FusedAdam):
    """`FusedAdam` whose step touches only this rank's range of the arenas.  Use together with
    `DataParallel(net)`: pass the wrapper so that its reducer switches to reduce-to-owner."""

    def __init__(self, params, data_parallel=None, process_group=None, **kw):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self._bounds: Optional[List[int]] = None
        super().__init__(params, **kw)
        if self._arena is None:
            raise ValueError('ShardedFusedAdam needs the complete trainable parameter set of one engine-bound model')
        if data_parallel is not None and self.world > 1:
            data_parallel.reducer.set_owner_shards(self._bounds)

    # ---- layout: moments only for the owned range ------------------------------------------
    def _alloc_moments(self, eng):
        self._bounds = shard_bounds(eng.lay.n, self.world)
        lo, hi = self._bounds[self.rank], self._bounds[self.rank + 1]
        dev = eng.P.device
        self._m = torch.zeros(hi - lo, device=dev, dtype=torch.float32)
        self._v = torch.zeros(hi - lo, device=dev, dtype=torch.float32)
        self._span = (lo, hi)

    def _step_arena(self, hyp):
        eng = self._arena
        G = eng.G
        if G is None or self._first.grad is None:
            return
        lo, hi = self._span
        ema_ptr, decay, ema_eng = None, 0.0, None
        if self._ema is not None:
            ema_eng = self._ema[0].engine()
            if ema_eng.lay.n != eng.lay.n:
                raise ValueError('fuse_ema: EMA model layout differs from the trained model')
            ema_ptr, decay = ema_eng.P.data_ptr() + 4 * lo, self._ema[1]
        lr, b1, b2, eps, wd, bc1, bc2 = hyp
        if hi > lo:
            call('mdt_adamw_ema_step', eng.P.data_ptr() + 4 * lo, G.data_ptr() + 4 * lo, self._m.data_ptr(), self._v.data_ptr(),
                 ema_ptr, None, hi - lo, lr, b1, b2, eps, wd, bc1, bc2, decay, float(self.grad_scale), _st())
        self._gather(eng.P)
        eng.refresh_shadows(cast=True)  # bf16 shadow of the gathered ranges + K-major transposes + label table
        if ema_eng is not None:
            ema_eng.shadows_dirty = True
            self._ema_stale = True
            eng.ema_applied = (id(ema_eng), decay)

    def _gather(self, flat: torch.Tensor):
        """In-place all-gather of the owned ranges of a flat arena (ranges may differ in length by the tail)."""
        if self.world == 1:
            return
        works = []
        for r in range(self.world):
            a, e = self._bounds[r], self._bounds[r + 1]
            if a < e:
                works.append(dist.broadcast(flat[a:e], src=r, group=self.pg, async_op=True))
        for w in works:
            w.wait()

    def sync_ema(self):
        """Make the EMA arena current on every rank (call before sampling from / saving the EMA model)."""
        if self._ema is not None and getattr(self, '_ema_stale', False):
            self._gather(self._ema[0].engine().P)
            self._ema[0].engine().shadows_dirty = True
            self._ema_stale = False

    # ---- checkpoints keep the unsharded (apex) layout ------------------------------------------
    def _full_moments(self):
        eng = self._arena
        full_m = torch.zeros(eng.lay.n, device=eng.P.device, dtype=torch.float32)
        full_v = torch.zeros_like(full_m)
        lo, hi = self._span
        full_m[lo:hi].copy_(self._m)
        full_v[lo:hi].copy_(self._v)
        self._gather(full_m)
        self._gather(full_v)
        return full_m, full_v

    def state_dict(self):
        eng = self._arena
        full_m, full_v = self._full_moments()
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
        lo, hi = self._span
        self._m.copy_(full_m[lo:hi])
        self._v.copy_(full_v[lo:hi])
        self.state.clear()
