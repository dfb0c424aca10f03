"""MaskDiT engine: flat parameter arenas + pre-bound launch plans over libmaskdit_hip.so.

Design (MI355X-first):
  * every trainable tensor of the reference state dict is a VIEW into one fp32 arena (same
    names / shapes as the reference checkpoint); gradients, Adam moments
    and the EMA live in arenas of the same layout, so the optimizer + EMA + bf16-shadow refresh
    is ONE streaming kernel and the DP gradient all-reduce works on contiguous slabs;
  * all adaLN modulation Linears (28 + 8 + 2 of them, each fed by the same SiLU(c)) are laid
    out contiguously so the forward needs ONE [B, D] x [N_mod, D]^T GEMM for every
    shift/scale/gate of the network, and the backward one dgrad + one wgrad GEMM;
  * GEMM operands are bf16 shadows of the fp32 master weights (N-major for forward,
    K-major transposed copies for the data-gradient GEMMs), refreshed by the optimizer;
  * a forward / backward pass is a *plan*: a list of (C entry point, pre-marshalled args) built
    once per (batch, mode) over preallocated HBM buffers -- replaying it is a tight loop of
    ctypes calls with no allocation, which also makes it hipGraph-capturable (sampler).

Reference semantics: DiT.forward / forward_encoder (models/maskdit.py:467-557), DiTBlock
(:170-192), DecoderLayer (:195-213), FinalLayer (:216-234), EDMPrecond (:756-773), EDMLoss
(train_utils/loss.py:28-60).  Backward = hand-derived gradients of the same graph.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import (EPI_BF16, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_GATE_RES, EPI_GELU, EPI_SILU, F32EPI_GATE_RES,
                   F32EPI_GELU, F32EPI_NONE, F32EPI_SILU, GemmF32Args, GemmNTArgs, GemmTNArgs)

DEC_HIDDEN, DEC_DEPTH, DEC_HEADS = 512, 8, 16  # models/maskdit.py:310-312
MODEL_CONFIGS = {  # models/maskdit.py:649-715  name -> (depth, hidden, patch, heads)
    'DiT-H/2': (32, 1280, 2, 16), 'DiT-H/4': (32, 1280, 4, 16), 'DiT-H/8': (32, 1280, 8, 16),
    'DiT-XL/2': (28, 1152, 2, 16), 'DiT-XL/4': (28, 1152, 4, 16), 'DiT-XL/8': (28, 1152, 8, 16),
    'DiT-L/2': (24, 1024, 2, 16), 'DiT-L/4': (24, 1024, 4, 16), 'DiT-L/8': (24, 1024, 8, 16),
    'DiT-B/2': (12, 768, 2, 12), 'DiT-B/4': (12, 768, 4, 12), 'DiT-B/8': (12, 768, 8, 12),
    'DiT-S/2': (12, 384, 2, 6), 'DiT-S/4': (12, 384, 4, 6), 'DiT-S/8': (12, 384, 8, 6),
}
YPAD = 1024  # label one-hot width padded to a multiple of 128 for the GEMMs
LIVE_ENGINES: 'weakref.WeakSet' = weakref.WeakSet()  # lets the optimizer map a parameter view back to its arena
# mdt_ln_modulate_bwd_gate: LayerNorm backward + the following residual-gate backward in one pass, 18 instead of
# 14 + 8 B/element.  Round 2's column-split kernel (a pair of waves per row, 4 x 12 instead of 4 x 20 running sums per
# lane, second-half loads issued before the row reduction): 631 vs 726 us for the pair on the XL/2 encoder rows at
# micro-batch 1024, 538 vs 606 us on the decoder rows (tools/ln_gate_bench.py); whole step 520.0 -> 518.5 ms on the
# same box.  History: the row-per-wave build (214 VGPRs) lost to the separate kernels (705-762 us), an LDS-atomics
# build was 4x slower (3017 us).  MDT_FUSE_LN_GATE=0 selects the two separate kernels (A/B runs).
FUSE_LN_GATE = os.environ.get('MDT_FUSE_LN_GATE', '1') != '0'
ADA_GROUP = 7  # encoder blocks per adaLN weight-gradient group (XL/2: 4 groups of 7 + the decoder-side group)
FUSE_QKV_COLSUM = os.environ.get('MDT_FUSE_QKV_COLSUM', '1') != '0'  # qkv bias gradient out of the qkv weight-gradient GEMM (A/B switch)
FUSE_COLSUM = os.environ.get('MDT_FUSE_COLSUM', '1') != '0'  # fc1 bias gradient out of the DGELU epilogue (A/B switch)
# Round 6: in TRAINING plans the residual add `x + gate * f(x)` (models/maskdit.py:190-191) is formed by the LayerNorm pass that
# consumes it (mdt_ln_modulate_fwd_res) instead of by the epilogue of the GEMM that produced f(x): the proj / fc2 GEMMs store only
# their bf16 output (plain class) -- 14 instead of 16 B per element, bit-identical x.  MDT_FUSE_RES_LN=0: the MDT_EPI_GATE_RES
# epilogues of rounds 1-5 (A/B runs).  Inference plans keep GATE_RES: they do not store the branch output at all.
FUSE_RES_LN = os.environ.get('MDT_FUSE_RES_LN', '1') != '0'
FUSE_RES_LN_EVAL = os.environ.get('MDT_FUSE_RES_LN_EVAL', '0') != '0'  # the same in inference plans (A/B)


def _rup(x, m):
    return (x + m - 1) // m * m


@dataclass
class Spec:
    model_type: str
    depth: int
    D: int
    heads: int
    patch: int
    R: int
    C: int
    num_classes: int
    mae: bool
    Dd: int = DEC_HIDDEN
    ddepth: int = DEC_DEPTH
    dheads: int = DEC_HEADS

    @property
    def T(self):
        return (self.R // self.patch) ** 2

    @property
    def hd(self):
        return self.D // self.heads

    @property
    def dhd(self):
        return self.Dd // self.dheads

    @property
    def pp(self):
        return self.patch * self.patch * self.C

    @property
    def n_mod(self):
        return self.depth * 6 * self.D + self.ddepth * 6 * self.Dd + 2 * self.D + 2 * self.Dd

    def mod_off(self, kind, i=0):
        """Row offset of an adaLN Linear inside the stacked [N_mod, D] weight."""
        if kind == 'enc':
            return i * 6 * self.D
        base = self.depth * 6 * self.D
        if kind == 'dec':
            return base + i * 6 * self.Dd
        base += self.ddepth * 6 * self.Dd
        if kind == 'dl':
            return base
        return base + 2 * self.D  # final


def make_spec(model_type, img_resolution, img_channels, num_classes, use_decoder=True, mae_loss_coef=0.1):
    if model_type not in MODEL_CONFIGS:
        raise ValueError(f'unknown model_type {model_type}')
    if not use_decoder:
        raise NotImplementedError('maskdit_amd accelerates the shipped configuration (use_decoder=True) only')
    depth, D, p, heads = MODEL_CONFIGS[model_type]
    sp = Spec(model_type, depth, D, heads, p, img_resolution, img_channels, num_classes, mae_loss_coef > 0)
    if sp.hd not in (32, 64, 72, 80):
        raise NotImplementedError(f'head_dim {sp.hd} unsupported')
    if sp.pp > 16 or sp.num_classes > YPAD or sp.num_classes <= 0:
        raise NotImplementedError('patch vector > 16 elements or num_classes outside (0, 1024]')
    T = sp.T
    if T & (T - 1) or T > 1024 or T % 64:
        raise NotImplementedError(f'token count {T} must be a power of two in [64, 1024]')
    return sp


# ------------------------------------------------------------------------------------------
# parameter layout

def param_table(sp: Spec) -> List[Tuple[str, tuple]]:
    """(state-dict key, shape) for every TRAINABLE tensor, in arena order."""
    D, Dd = sp.D, sp.Dd
    ada_w, ada_b, rest = [], [], []

    def ada(prefix, n):
        ada_w.append((f'{prefix}.adaLN_modulation.1.weight', (n, D)))
        ada_b.append((f'{prefix}.adaLN_modulation.1.bias', (n,)))

    def block(prefix, W):
        h = 4 * W
        rest.extend([(f'{prefix}.attn.qkv.weight', (3 * W, W)), (f'{prefix}.attn.qkv.bias', (3 * W,)),
                     (f'{prefix}.attn.proj.weight', (W, W)), (f'{prefix}.attn.proj.bias', (W,)),
                     (f'{prefix}.mlp.fc1.weight', (h, W)), (f'{prefix}.mlp.fc1.bias', (h,)),
                     (f'{prefix}.mlp.fc2.weight', (W, h)), (f'{prefix}.mlp.fc2.bias', (W,))])

    for i in range(sp.depth):
        ada(f'model.blocks.{i}', 6 * D)
        block(f'model.blocks.{i}', D)
    for i in range(sp.ddepth):
        ada(f'model.decoder_blocks.{i}', 6 * Dd)
        block(f'model.decoder_blocks.{i}', Dd)
    ada('model.decoder_layer', 2 * D)
    ada('model.final_layer', 2 * Dd)
    rest.extend([('model.decoder_layer.linear.weight', (Dd, D)), ('model.decoder_layer.linear.bias', (Dd,)),
                 ('model.final_layer.linear.weight', (sp.pp, Dd)), ('model.final_layer.linear.bias', (sp.pp,)),
                 ('model.x_embedder.proj.weight', (D, sp.C, sp.patch, sp.patch)), ('model.x_embedder.proj.bias', (D,)),
                 ('model.t_embedder.mlp.0.weight', (D, 256)), ('model.t_embedder.mlp.0.bias', (D,)),
                 ('model.t_embedder.mlp.2.weight', (D, D)), ('model.t_embedder.mlp.2.bias', (D,)),
                 ('model.y_embedder.embedding_table.weight', (D, sp.num_classes))])
    if sp.mae:
        rest.append(('model.mask_token', (1, 1, Dd)))
    return ada_w + ada_b + rest


class Layout:
    """Offsets (in elements) of every trainable tensor inside the flat arenas."""

    def __init__(self, sp: Spec):
        self.sp = sp
        self.off: Dict[str, int] = {}
        self.shape: Dict[str, tuple] = {}
        o = 0
        for name, shp in param_table(sp):
            n = int(np.prod(shp))
            self.off[name] = o
            self.shape[name] = shp
            o += _rup(n, 8)  # adaLN tensors are multiples of 8 elements, so their stacks stay dense
        self.n = _rup(o, 8)
        self.ada_w = self.off['model.blocks.0.adaLN_modulation.1.weight']
        self.ada_b = self.off['model.blocks.0.adaLN_modulation.1.bias']
        # K-major (transposed) shadows for the data-gradient GEMMs
        self.t_off: Dict[str, int] = {}
        self.t_entries: List[Tuple[int, int, int, int]] = []
        to = 0

        def addT(key, src, rows, cols):
            nonlocal to
            self.t_off[key] = to
            self.t_entries.append((src, to, rows, cols))
            to += _rup(rows * cols, 8)

        addT('ada', self.ada_w, sp.n_mod, sp.D)
        for name, shp in param_table(sp):
            if name.endswith(('attn.qkv.weight', 'attn.proj.weight', 'mlp.fc1.weight', 'mlp.fc2.weight')) \
                    or name in ('model.decoder_layer.linear.weight', 'model.t_embedder.mlp.2.weight'):
                addT(name, self.off[name], shp[0], shp[1])
        self.nt = to
        # block-wise slabs of the arena for gradient all-reduce overlap (name -> (start, end))
        self.slabs: Dict[str, Tuple[int, int]] = {}

        def span(prefix_first, prefix_last_end_name):
            return self.off[prefix_first], self.off[prefix_last_end_name] + _rup(int(np.prod(self.shape[prefix_last_end_name])), 8)

        for i in range(sp.depth):
            self.slabs[f'enc{i}'] = span(f'model.blocks.{i}.attn.qkv.weight', f'model.blocks.{i}.mlp.fc2.bias')
        for i in range(sp.ddepth):
            self.slabs[f'dec{i}'] = span(f'model.decoder_blocks.{i}.attn.qkv.weight', f'model.decoder_blocks.{i}.mlp.fc2.bias')
        # the stacked adaLN weight [N_mod, D] is 35 % of all gradient bytes: its rows are reduced in groups
        # whose modulation gradients become final at different points of the backward pass (decoder side
        # first, then the encoder blocks from the top in groups of ADA_GROUP), not in one piece at the end
        self.ada_groups: List[Tuple[str, int, int]] = []  # (slab name, first row, end row) in backward order
        enc_rows = sp.depth * 6 * sp.D
        self.ada_groups.append(('ada_w_dec', enc_rows, sp.n_mod))
        hi_blk = sp.depth
        while hi_blk > 0:
            lo_blk = max(0, hi_blk - ADA_GROUP)
            self.ada_groups.append((f'ada_w_enc{lo_blk}', lo_blk * 6 * sp.D, hi_blk * 6 * sp.D))
            hi_blk = lo_blk
        for name, r0, r1 in self.ada_groups:
            self.slabs[name] = (self.ada_w + r0 * sp.D, self.ada_w + r1 * sp.D)
        self.slabs['ada_b'] = (self.ada_b, self.off['model.blocks.0.attn.qkv.weight'])
        self.slabs['misc'] = (self.off['model.decoder_layer.linear.weight'], self.n)


# ------------------------------------------------------------------------------------------
# plans

class Plan:
    """A replayable list of pre-marshalled launches (and optional python callbacks)."""

    def __init__(self):
        self.calls: list = []
        self.keep: list = []

    def add(self, name, *args):
        fn = getattr(_lib.lib(), name)
        self.calls.append((fn, args, name))

    def add_callback(self, cb: Callable[[], None]):
        self.calls.append((None, cb, 'callback'))

    def run(self, stream: int):
        for fn, args, name in self.calls:
            if fn is None:
                args()
                continue
            rc = fn(*args, stream)
            if rc != 0:
                raise _lib.MaskDiTLibError(f'{name} failed ({rc}): {_lib.lib().mdt_last_error().decode()}')


class Engine:
    """Owns arenas, shadows and launch plans for one EDMPrecond/DiT instance on one GPU."""

    def __init__(self, sp: Spec, device):
        self.sp = sp
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.MaskDiTLibError('maskdit_amd.Engine needs a CUDA/HIP device: there is no CPU path')
        _lib.lib()
        self.lay = Layout(sp)
        dev = self.device
        self.P = torch.zeros(self.lay.n, device=dev, dtype=torch.float32)
        self.G: Optional[torch.Tensor] = None
        self.W16 = torch.zeros(self.lay.n, device=dev, dtype=torch.bfloat16)
        self.WT16 = torch.zeros(self.lay.nt, device=dev, dtype=torch.bfloat16)
        self.Wy16 = torch.zeros(sp.D, YPAD, device=dev, dtype=torch.bfloat16)
        self.pos = torch.zeros(sp.T, sp.D, device=dev, dtype=torch.float32)
        self.dpos = torch.zeros(sp.T, sp.Dd, device=dev, dtype=torch.float32)
        tab, tiles = [], 0
        for src, dst, rows, cols in self.lay.t_entries:
            tab += [src, dst, rows, cols, tiles]
            tiles += ((rows + 63) // 64) * ((cols + 63) // 64)
        self._t_table = torch.tensor(tab, dtype=torch.int64, device=dev)
        self._t_tiles = tiles
        self.shadows_dirty = True
        self._plans: Dict[tuple, 'PassPlan'] = {}   # insertion order = LRU order (plan() re-inserts on every hit)
        self.plan_headroom = 0.05                   # fraction of the device a new plan leaves free (see plan())
        self.grad_slab_hook: Optional[Callable[[str, int, int], None]] = None  # DP overlap (ddp.py)
        self.ema_applied = None
        LIVE_ENGINES.add(self)

    # ---- arenas ------------------------------------------------------------------------
    def view(self, arena: torch.Tensor, name: str) -> torch.Tensor:
        o, shp = self.lay.off[name], self.lay.shape[name]
        return arena[o:o + int(np.prod(shp))].view(shp)

    def ensure_grad(self) -> torch.Tensor:
        if self.G is None:
            self.G = torch.zeros(self.lay.n, device=self.device, dtype=torch.float32)
        return self.G

    def refresh_shadows(self, cast: bool = True):
        """bf16 N-major shadow (optional: the optimizer already wrote it), K-major transposes,
        padded label table."""
        st = torch.cuda.current_stream().cuda_stream
        sp, lay = self.sp, self.lay
        if cast:
            _lib.call('mdt_cast_f32_bf16', self.P.data_ptr(), lay.n, self.W16.data_ptr(), lay.n, 1, lay.n, 0, st)
        _lib.call('mdt_transpose_bf16_batched', self.W16.data_ptr(), self.WT16.data_ptr(), self._t_table.data_ptr(),
                  len(lay.t_entries), self._t_tiles, st)
        yo = lay.off['model.y_embedder.embedding_table.weight']
        _lib.call('mdt_cast_f32_bf16', self.P.data_ptr() + 4 * yo, sp.num_classes, self.Wy16.data_ptr(), YPAD, sp.D,
                  sp.num_classes, 0, st)
        self.shadows_dirty = False

    # ---- plans -------------------------------------------------------------------------
    def plan(self, B: int, masked: bool, train: bool, L: Optional[int] = None, precision: str = 'bf16') -> 'PassPlan':
        """Plans are keyed on the kept-token count ROUNDED UP to the 64-row tile: a mask-ratio schedule
        (train_utils/helper.py:9-27) changes the exact count nearly every step, but buffers and launch lists only
        depend on the padded count -- the exact one is a run-time argument of the four launches that read it
        (PassPlan.set_valid)."""
        global _PLAN_CLOCK
        if precision not in ('bf16', 'fp32'):
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        if precision == 'fp32' and (masked or train):
            raise NotImplementedError('the fp32-faithful path covers inference (the sampler / eval forward: sample.py:56); '
                                      'fp32 TRAINING (train.py --no_amp) is not provided')
        Lv = (L if L is not None else self.sp.T // 2) if masked else None
        key = plan_key(B, masked, train, Lv, precision)
        pl = self._plans.pop(key, None)
        if pl is None:
            # LRU cache under a DEVICE-wide memory budget.  Round 2 dropped every training plan whenever another shape
            # was requested; round 3 budgeted each engine against 0.80 of the device on its own, so with two engines
            # alive (train.py: net + ema; a test process: whatever the previous tests left behind) the second one could
            # neither see nor evict the first one's plans and died with OutOfMemoryError.  Now: what a new
            # plan needs is compared with what the device really has free (torch.cuda.mem_get_info + the caching
            # allocator's idle blocks), and the least-recently-used plan of ANY live engine on this device is dropped
            # until it fits (5 % of the device stays free for the caller's own tensors) or nothing is left to drop.
            while len(self._plans) >= 8:
                _drop_plan(self, next(iter(self._plans)))
            need = PassPlan.estimate_bytes(self.sp, B, masked, train, Lv) * (2 if precision == 'fp32' else 1)
            # (a plan something else still refers to -- a pending backward, a captured sampler graph -- gives no memory back
            # when its cache entry goes: after two evictions in a row that freed nothing the loop stops instead of emptying
            # every engine's cache for no gain)
            fruitless = 0
            while need + self.plan_headroom * _device_total(self.device) > _device_available(self.device) and fruitless < 2:
                before = _device_available(self.device)
                if not _evict_lru_plan(self.device):
                    break
                fruitless = fruitless + 1 if _device_available(self.device) <= before else 0
            while True:
                try:
                    pl = PassPlan(self, B, masked, train, Lv, precision)
                    break
                except torch.cuda.OutOfMemoryError:
                    if not _evict_lru_plan(self.device):
                        raise
                    torch.cuda.empty_cache()
        elif masked:
            pl.set_valid(Lv)
        _PLAN_CLOCK += 1
        pl.last_use = _PLAN_CLOCK
        self._plans[key] = pl  # most recently used = last
        return pl

    def release_plans(self):
        for key in list(self._plans):
            _drop_plan(self, key)


_PLAN_CLOCK = 0


def plan_key(B: int, masked: bool, train: bool, Lv: Optional[int] = None, precision: str = 'bf16') -> tuple:
    """Cache key of Engine.plan(): bf16 keys keep their round-1..5 form; an fp32 plan is another entry."""
    key = (B, masked, train, _rup(Lv, 64) if masked else None)
    return key if precision == 'bf16' else key + (precision,)


def _device_total(device) -> int:
    return torch.cuda.get_device_properties(device).total_memory


def _device_available(device) -> int:
    """Bytes a new allocation can get: free device memory + blocks the caching allocator holds but does not use."""
    free, _ = torch.cuda.mem_get_info(device)
    return free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)


def _drop_plan(eng, key):
    """THE way a plan leaves a cache: whoever replays its launch list (the sampler's captured graphs) is told."""
    pl = eng._plans.pop(key)
    for hook in pl.evict_hooks:
        hook()
    pl.evict_hooks = []


def _evict_lru_plan(device) -> bool:
    """Drop the least-recently-used cached plan among ALL live engines on `device`; False when none is cached.  (A plan a
    pending backward still refers to stays alive through that reference -- only the cache entry goes.)"""
    best = None
    for eng in list(LIVE_ENGINES):
        if eng.device != torch.device(device):
            continue
        for key, pl in eng._plans.items():
            if best is None or pl.last_use < best[2].last_use:
                best = (eng, key, pl)
    if best is None:
        return False
    _drop_plan(best[0], best[1])
    return True


def _nt(A, lda, Bw, ldb, M, N, K, bias=0, epi=EPI_BF16, out=0, ldo=0, out2=0, ldo2=0, outf=0, ldof=0, res=0, ldres=0,
        gate=0, gate_ld=0, rps=1, aux=0, ldaux=0, k_splits=0, colsum=0):
    a = GemmNTArgs()
    a.A, a.lda, a.B, a.ldb, a.M, a.N, a.K = A, lda, Bw, ldb, M, N, K
    a.bias, a.epi = bias or None, epi
    a.out, a.ldo, a.out2, a.ldo2 = out or None, ldo, out2 or None, ldo2
    a.outf, a.ldof, a.res, a.ldres = outf or None, ldof, res or None, ldres
    a.gate, a.gate_ld, a.rows_per_sample = gate or None, gate_ld, rps
    a.aux, a.ldaux = aux or None, ldaux
    a.k_splits = k_splits
    a.colsum = colsum or None
    return a


def _tn(A, lda, Bm, ldb, M, N1, N2, Cc, ldc, n1v=0, n2v=0, splits=0, colsum_a=0):
    a = GemmTNArgs()
    a.A, a.lda, a.B, a.ldb, a.M, a.N1, a.N2 = A, lda, Bm, ldb, M, N1, N2
    a.C, a.ldc, a.n1_valid, a.n2_valid, a.splits = Cc, ldc, n1v, n2v, splits
    a.colsum_a = colsum_a or None
    return a


class PassPlan:
    """Buffers + forward/backward launch lists for a fixed (batch, masked?, train?) shape."""

    def __init__(self, eng: Engine, B: int, masked: bool, train: bool, L: Optional[int], precision: str = 'bf16'):
        sp = eng.sp
        self.precision = precision
        # the engine caches its plans, so a plan must not own its engine: Engine <-> PassPlan reference cycles kept
        # 20-240 GB of a dropped model alive until Python's cyclic collector happened to run
        self._eng_ref = weakref.ref(eng)
        self.B, self.masked, self.train = B, masked, train
        self.last_use = 0
        self.evict_hooks: List[Callable[[], None]] = []  # run when the plan cache drops this plan for memory
        self.T = sp.T
        # Lv = kept tokens per sample (models/maskdit.py:99: int(T * (1 - mask_ratio)), any value that a
        # mask-ratio schedule produces); the encoder runs on L = Lv rounded up to the 64-row tile: the
        # padding rows gather (removed) tokens, are masked out as attention KEYS, are skipped by the
        # un-masking gather, and receive exactly zero gradient (mdt_unmask_bwd), so every other kernel
        # simply sees L rows per sample.
        self.Lv = (L if L is not None else sp.T // 2) if masked else sp.T
        if not (1 <= self.Lv <= sp.T):
            raise ValueError(f'kept-token count {self.Lv} outside [1, {sp.T}]')
        self.L = _rup(self.Lv, 64)
        # the exact count as a MUTABLE C int inside the pre-marshalled argument tuples (ctypes reads .value at call
        # time): attention key masking + the un-masking gather / scatter are the only launches that see it
        self.lv_arg = C.c_int(self.Lv)
        self.lv_attn = C.c_int(self.Lv if masked else 0)
        self.Bp = _rup(B, 64)
        self.buf: Dict[str, torch.Tensor] = {}
        self.fwd = Plan()
        self.bwd = Plan()
        self.gen = 0  # forward generation: the saved activations belong to the LAST forward through this plan
        if precision == 'fp32':
            self._build_f32()
        else:
            self._build()

    @property
    def eng(self) -> Engine:
        e = self._eng_ref()
        if e is None:
            raise RuntimeError('maskdit_amd: this plan outlived its Engine (the model was deleted or re-bound)')
        return e

    @property
    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.buf.values())

    @staticmethod
    def estimate_bytes(sp, B: int, masked: bool, train: bool, Lv: Optional[int]) -> int:
        """Upper estimate of a plan's buffers before it is built (measured: 244 GB in use for XL/2 at B = 1024 incl.
        30 GB of arenas; inference plans keep one block's worth of activations)."""
        L = _rup(Lv, 64) if masked else sp.T
        per_tok_enc = 40 * sp.D if train else 0      # bytes of saved activations per encoder token and block
        per_tok_dec = 40 * sp.Dd if train else 0
        live = 64 * max(sp.D, sp.Dd) * sp.T          # per-sample working set of one block (fwd-only plans)
        io = 6 * 4 * sp.C * sp.R * sp.R              # yn / D / y / noise-sized fp32 images of the loss + precond algebra
        return int(B * (sp.depth * per_tok_enc * L + sp.ddepth * per_tok_dec * sp.T + live + io) * 1.05)

    def set_valid(self, Lv: int):
        if _rup(Lv, 64) != self.L or not (1 <= Lv <= self.T):
            raise ValueError(f'kept-token count {Lv} does not belong to this plan (padded count {self.L})')
        self.Lv = Lv
        self.lv_arg.value = Lv
        self.lv_attn.value = Lv

    # ---- buffers -----------------------------------------------------------------------
    def t(self, name, shape, dtype):
        if name in self.buf:
            return self.buf[name]
        x = torch.zeros(shape, device=self.eng.device, dtype=dtype)
        self.buf[name] = x
        return x

    def f32(self, name, *shape):
        return self.t(name, shape, torch.float32)

    def b16(self, name, *shape):
        return self.t(name, shape, torch.bfloat16)

    def _build(self):
        eng, sp, lay = self.eng, self.eng.sp, self.eng.lay
        B, Bp, T, L, D, Dd = self.B, self.Bp, self.T, self.L, sp.D, sp.Dd
        NM = sp.n_mod
        train = self.train
        Pp, W16p, WTp = eng.P.data_ptr(), eng.W16.data_ptr(), eng.WT16.data_ptr()

        def Pf(name):  # fp32 master pointer
            return Pp + 4 * lay.off[name]

        def Wp(name):  # bf16 N-major shadow pointer
            return W16p + 2 * lay.off[name]

        def WT(key):  # bf16 K-major shadow pointer
            return WTp + 2 * lay.t_off[key]

        f, g = self.fwd, self.bwd
        # ---------------- inputs -----------------------------------------------------------
        xin = self.f32('xin', B, sp.C, sp.R, sp.R)
        coef = self.f32('coef', 8, B)  # rows: c_skip, c_out, c_in, c_noise, weight, sigma, -, -
        cn = coef[3]
        self.buf['c_noise'] = cn
        lab = self.f32('labels', B, sp.num_classes)
        ids32 = self.t('ids32', (B, 2 * T), torch.int32) if self.masked else None
        Fx = self.f32('F', B, sp.C, sp.R, sp.R)
        # EDMLoss's own buffers (loss.py; 16 KB per sample): allocated here so that they are covered by plan()'s
        # out-of-memory handling like everything else
        for nm in ('yn', 'D', 'y'):
            self.f32(nm, B, sp.C, sp.R, sp.R)
        # ---------------- conditioning path ---------------------------------------------------
        temb = self.b16('temb', Bp, 256)
        h1, a1 = self.b16('h1', Bp, D), self.b16('a1', Bp, D)
        c_t, c_y, c = self.f32('c_t', Bp, D), self.f32('c_y', Bp, D), self.f32('c', Bp, D)
        lab16 = self.b16('lab16', Bp, YPAD)
        sc16 = self.b16('sc16', Bp, D)
        mod = self.f32('mod', Bp, NM)
        f.add('mdt_timestep_embed', cn.data_ptr(), temb.data_ptr(), 256, B, 256)
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(temb.data_ptr(), 256, Wp('model.t_embedder.mlp.0.weight'), 256, B, D, 256,
                                               bias=Pf('model.t_embedder.mlp.0.bias'), epi=EPI_SILU, out=h1.data_ptr(), ldo=D,
                                               out2=a1.data_ptr(), ldo2=D))))
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(a1.data_ptr(), D, Wp('model.t_embedder.mlp.2.weight'), D, B, D, D,
                                               bias=Pf('model.t_embedder.mlp.2.bias'), epi=EPI_F32, outf=c_t.data_ptr(), ldof=D))))
        f.add('mdt_cast_f32_bf16', lab.data_ptr(), sp.num_classes, lab16.data_ptr(), YPAD, B, sp.num_classes, 0)
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(lab16.data_ptr(), YPAD, eng.Wy16.data_ptr(), YPAD, B, D, YPAD, epi=EPI_F32,
                                               outf=c_y.data_ptr(), ldof=D))))
        f.add('mdt_add_f32', c_t.data_ptr(), c_y.data_ptr(), c.data_ptr(), B * D)
        f.add('mdt_cast_f32_bf16', c.data_ptr(), D, sc16.data_ptr(), D, B, D, 1)
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(sc16.data_ptr(), D, W16p + 2 * lay.ada_w, D, B, NM, D, bias=Pp + 4 * lay.ada_b,
                                               epi=EPI_F32, outf=mod.data_ptr(), ldof=NM))))
        # ---------------- encoder --------------------------------------------------------------
        Me = B * L
        x0 = self.f32('x_e0', Me, D)
        f.add('mdt_patch_embed_fwd', xin.data_ptr(), None, Pf('model.x_embedder.proj.weight'),
              Pf('model.x_embedder.proj.bias'), eng.pos.data_ptr(), ids32.data_ptr() if ids32 is not None else None,
              2 * T, x0.data_ptr(), B, sp.C, sp.R, sp.patch, L, D)
        xs_e = [x0]
        fuse = FUSE_RES_LN and (train or FUSE_RES_LN_EVAL)
        pend = None  # (xres, y, gate address): the residual add the NEXT LayerNorm pass has to perform into xs_e[-1]
        for i in range(sp.depth):
            xo, pend = self._block_fwd(f'model.blocks.{i}', 'e', i, xs_e[-1], mod, sp.mod_off('enc', i), D, sp.heads, L, Me,
                                       lvalid=self.lv_attn, pending=pend, defer_out=fuse)
            xs_e.append(xo)
        # ---------------- decoder layer + unmask ----------------------------------------------
        self.marks = {'enc_fwd_end': len(f.calls)}  # launch index where the encoder (+ conditioning path) forward ends
        odl = sp.mod_off('dl')
        xnd = self.b16('xn_dl', Me, D)
        st_dl = self.f32('st_dl', Me, 2)
        xdec = self.b16('xdec', Me, Dd)
        if pend is not None:  # the top encoder block's MLP residual is formed here
            f.add('mdt_ln_modulate_fwd_res', pend[0].data_ptr(), pend[1].data_ptr(), pend[2], NM, mod.data_ptr() + 4 * odl,
                  mod.data_ptr() + 4 * (odl + D), NM, L, xs_e[-1].data_ptr(), xnd.data_ptr(), st_dl.data_ptr(), Me, D)
            self.marks['enc_fwd_end'] = len(f.calls)  # (it is encoder work: the mark moves behind it)
        else:
            f.add('mdt_ln_modulate_fwd', xs_e[-1].data_ptr(), mod.data_ptr() + 4 * odl, mod.data_ptr() + 4 * (odl + D), NM, L,
                  xnd.data_ptr(), st_dl.data_ptr(), Me, D)
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(xnd.data_ptr(), D, Wp('model.decoder_layer.linear.weight'), D, Me, Dd, D,
                                               bias=Pf('model.decoder_layer.linear.bias'), epi=EPI_BF16, out=xdec.data_ptr(), ldo=Dd))))
        Md = B * T
        xd0 = self.f32('x_d0', Md, Dd)
        use_mt = self.masked and sp.mae
        f.add('mdt_unmask_fwd', xdec.data_ptr(), (ids32.data_ptr() + 4 * T) if self.masked else None, 2 * T,
              Pf('model.mask_token') if use_mt else None, eng.dpos.data_ptr(), xd0.data_ptr(), B, T, self.lv_arg, Dd, L)
        xs_d = [xd0]
        pend = None
        for i in range(sp.ddepth):  # (the last block forms its own output: mdt_final_fwd reads it)
            xo, pend = self._block_fwd(f'model.decoder_blocks.{i}', 'd', i, xs_d[-1], mod, sp.mod_off('dec', i), Dd, sp.dheads, T, Md,
                                       pending=pend, defer_out=fuse and i + 1 < sp.ddepth)
            xs_d.append(xo)
        ofin = sp.mod_off('fin')
        st_f = self.f32('st_f', Md, 2)
        f.add('mdt_final_fwd', xs_d[-1].data_ptr(), mod.data_ptr() + 4 * ofin, mod.data_ptr() + 4 * (ofin + Dd), NM,
              Pf('model.final_layer.linear.weight'), Pf('model.final_layer.linear.bias'), Fx.data_ptr(), st_f.data_ptr(),
              B, T, Dd, sp.C, sp.patch)
        if not train:
            return
        # =================== backward ===========================================================
        G = eng.ensure_grad()
        Gp = G.data_ptr()

        def Gf(name):
            return Gp + 4 * lay.off[name]

        dF = self.f32('dF', B, sp.C, sp.R, sp.R)
        dmod = self.f32('dmod', Bp, NM)
        wmax = max(Me * D, Md * Dd)
        dxe = self.f32('dx_e', Me, D)
        dxd = self.f32('dx_d', Md, Dd)
        ws = dict(dys=self.b16('ws_dys', wmax), dh=self.b16('ws_dh', 4 * wmax), dxn=self.b16('ws_dxn', wmax),
                  dao=self.b16('ws_dao', wmax), dqkv=self.b16('ws_dqkv', 3 * wmax),
                  delta=self.f32('ws_delta', max(B * sp.heads * L, B * sp.dheads * T)))
        self._ws = ws
        g.add_callback(lambda: dmod.zero_())
        g.add('mdt_final_bwd', dF.data_ptr(), xs_d[-1].data_ptr(), st_f.data_ptr(), mod.data_ptr() + 4 * ofin,
              mod.data_ptr() + 4 * (ofin + Dd), NM, Pf('model.final_layer.linear.weight'), dxd.data_ptr(),
              Gf('model.final_layer.linear.weight'), Gf('model.final_layer.linear.bias'), dmod.data_ptr() + 4 * ofin,
              dmod.data_ptr() + 4 * (ofin + Dd), NM, B, T, Dd, sp.C, sp.patch)
        for i in reversed(range(sp.ddepth)):
            # the LN1 backward that ends block i also runs the MLP-gate backward that would open block i-1
            nxt = self._gate_info(f'model.decoder_blocks.{i - 1}', 'd', i - 1, mod, dmod, sp.mod_off('dec', i - 1), Dd, Gf) \
                if (i > 0 and FUSE_LN_GATE) else None
            self._block_bwd(f'model.decoder_blocks.{i}', 'd', i, xs_d[i], mod, dmod, sp.mod_off('dec', i), Dd, sp.dheads, T, Md,
                            dxd, Gf, fuse_next=nxt, skip_first_gate=(FUSE_LN_GATE and i < sp.ddepth - 1))
            self._slab(f'dec{i}')
        dxdec = self.b16('dxdec', Me, Dd)
        g.add('mdt_unmask_bwd', dxd.data_ptr(), ids32.data_ptr() if self.masked else None, 2 * T, dxdec.data_ptr(),
              Gf('model.mask_token') if use_mt else None, B, T, self.lv_arg, Dd, L)
        g.add('mdt_gemm_tn', C.byref(self._k(_tn(dxdec.data_ptr(), Dd, xnd.data_ptr(), D, Me, Dd, D,
                                               Gf('model.decoder_layer.linear.weight'), D))))
        g.add('mdt_colsum_bf16', dxdec.data_ptr(), Dd, Gf('model.decoder_layer.linear.bias'), Me, Dd)
        g.add('mdt_gemm_nt', C.byref(self._k(_nt(dxdec.data_ptr(), Dd, WT('model.decoder_layer.linear.weight'), Dd, Me, D, Dd,
                                               epi=EPI_BF16, out=ws['dxn'].data_ptr(), ldo=D))))
        if FUSE_LN_GATE:
            top = self._gate_info(f'model.blocks.{sp.depth - 1}', 'e', sp.depth - 1, mod, dmod, sp.mod_off('enc', sp.depth - 1), D, Gf)
            g.add('mdt_ln_modulate_bwd_gate', ws['dxn'].data_ptr(), xs_e[-1].data_ptr(), st_dl.data_ptr(), mod.data_ptr() + 4 * (odl + D),
                  NM, L, dxe.data_ptr(), 0, dmod.data_ptr() + 4 * odl, dmod.data_ptr() + 4 * (odl + D), NM, Me, D, *top)
        else:
            g.add('mdt_ln_modulate_bwd', ws['dxn'].data_ptr(), xs_e[-1].data_ptr(), st_dl.data_ptr(), mod.data_ptr() + 4 * (odl + D),
                  NM, L, dxe.data_ptr(), 0, dmod.data_ptr() + 4 * odl, dmod.data_ptr() + 4 * (odl + D), NM, Me, D)
        dmod16 = self.b16('dmod16', Bp, NM)

        def ada_group(name, r0, r1):
            """modulation columns [r0, r1) are final: bias grads, weight grads of those adaLN rows, slab hook"""
            n = r1 - r0
            g.add('mdt_cast_f32_bf16', dmod.data_ptr() + 4 * r0, NM, dmod16.data_ptr() + 2 * r0, NM, B, n, 0)
            g.add('mdt_colsum_bf16', dmod16.data_ptr() + 2 * r0, NM, Gp + 4 * (lay.ada_b + r0), B, n)
            g.add('mdt_gemm_tn', C.byref(self._k(_tn(dmod16.data_ptr() + 2 * r0, NM, sc16.data_ptr(), D, Bp, n, D,
                                                   Gp + 4 * (lay.ada_w + r0 * D), D))))
            self._slab(name)

        groups = {name: (r0, r1) for name, r0, r1 in lay.ada_groups}
        ada_group('ada_w_dec', *groups['ada_w_dec'])  # final layer, decoder blocks, decoder layer: all done above
        self.marks['enc_bwd_begin'] = len(g.calls)  # everything from here on is encoder / conditioning-path backward
        for i in reversed(range(sp.depth)):
            nxt = self._gate_info(f'model.blocks.{i - 1}', 'e', i - 1, mod, dmod, sp.mod_off('enc', i - 1), D, Gf) \
                if (i > 0 and FUSE_LN_GATE) else None
            self._block_bwd(f'model.blocks.{i}', 'e', i, xs_e[i], mod, dmod, sp.mod_off('enc', i), D, sp.heads, L, Me, dxe, Gf,
                            fuse_next=nxt, skip_first_gate=FUSE_LN_GATE, lvalid=self.lv_attn)
            self._slab(f'enc{i}')
            if f'ada_w_enc{i}' in groups:  # block i is the lowest block of its group
                ada_group(f'ada_w_enc{i}', *groups[f'ada_w_enc{i}'])
        g.add('mdt_patch_embed_bwd', xin.data_ptr(), None, dxe.data_ptr(), ids32.data_ptr() if ids32 is not None else None,
              2 * T, Gf('model.x_embedder.proj.weight'), Gf('model.x_embedder.proj.bias'), B, sp.C, sp.R, sp.patch, L, D)
        # ---- conditioning path backward ------------------------------------------------------
        dsc = self.f32('dsc', Bp, D)
        dc16 = self.b16('dc16', Bp, D)
        dh1 = self.b16('dh1', Bp, D)
        self._slab('ada_b')  # every group has written its part of the stacked adaLN bias gradient
        # M = batch, N = D, K = every modulation output (221 k on XL/2): split the contraction
        g.add('mdt_gemm_nt', C.byref(self._k(_nt(dmod16.data_ptr(), NM, WT('ada'), NM, B, D, NM, epi=EPI_F32,
                                               outf=dsc.data_ptr(), ldof=D,
                                               k_splits=max(1, min(64, NM // 2048, 1024 // (((B + 127) // 128) * (D // 128))))))))
        g.add('mdt_silu_bwd', dsc.data_ptr(), c.data_ptr(), dc16.data_ptr(), B * D)
        g.add('mdt_gemm_tn', C.byref(self._k(_tn(dc16.data_ptr(), D, lab16.data_ptr(), YPAD, Bp, D, YPAD,
                                               Gf('model.y_embedder.embedding_table.weight'), sp.num_classes,
                                               n1v=D, n2v=sp.num_classes))))
        g.add('mdt_gemm_tn', C.byref(self._k(_tn(dc16.data_ptr(), D, a1.data_ptr(), D, Bp, D, D,
                                               Gf('model.t_embedder.mlp.2.weight'), D))))
        g.add('mdt_colsum_bf16', dc16.data_ptr(), D, Gf('model.t_embedder.mlp.2.bias'), B, D)
        g.add('mdt_gemm_nt', C.byref(self._k(_nt(dc16.data_ptr(), D, WT('model.t_embedder.mlp.2.weight'), D, B, D, D,
                                               epi=EPI_DSILU, out=dh1.data_ptr(), ldo=D, aux=h1.data_ptr(), ldaux=D))))
        g.add('mdt_gemm_tn', C.byref(self._k(_tn(dh1.data_ptr(), D, temb.data_ptr(), 256, Bp, D, 256,
                                               Gf('model.t_embedder.mlp.0.weight'), 256))))
        g.add('mdt_colsum_bf16', dh1.data_ptr(), D, Gf('model.t_embedder.mlp.0.bias'), B, D)
        self._slab('misc')

    # ---- the fp32-faithful inference plan (csrc/f32path.hip) ------------------------------------------------------
    def _build_f32(self):
        """The eval forward in EXACT fp32 -- what the reference's sampler runs (sample.py:56 `net(x_hat.float(), ...)`, no
        autocast in generate.py): fp32 master weights straight from the parameter arena (no bf16 shadow is read), fp32
        activations, fp32-input MFMA GEMMs (mdt_gemm_f32), attention as q k^T -> row softmax -> p v over the packed qkv
        buffer.  Same graph as _build (DiT.forward, models/maskdit.py:511-557, unmasked), same input / output buffers
        ('xin', 'coef', 'labels', 'F'), so the sampler and EDMPrecond.forward drive either plan the same way."""
        eng, sp, lay = self.eng, self.eng.sp, self.eng.lay
        B, T, D, Dd = self.B, self.T, sp.D, sp.Dd
        NM = sp.n_mod
        assert not self.masked and not self.train
        if sp.num_classes % 4:
            raise NotImplementedError('fp32 path: num_classes must be a multiple of 4 (16-byte label rows)')
        if B * max(sp.heads, sp.dheads) > 65535:
            raise NotImplementedError('fp32 path: batch * heads must not exceed 65535 (one grid row per (sample, head))')
        Pp = eng.P.data_ptr()

        def Pf(name):
            return Pp + 4 * lay.off[name]

        f = self.fwd
        xin = self.f32('xin', B, sp.C, sp.R, sp.R)
        coef = self.f32('coef', 8, B)
        cn = coef[3]
        self.buf['c_noise'] = cn
        lab = self.f32('labels', B, sp.num_classes)
        Fx = self.f32('F', B, sp.C, sp.R, sp.R)
        # ---------------- conditioning path (TimestepEmbedder :34-60, LabelEmbedder :75, SiLU + adaLN Linears :183-186)
        temb, a1 = self.f32('temb', B, 256), self.f32('a1', B, D)
        c_t, c, sc = self.f32('c_t', B, D), self.f32('c', B, D), self.f32('sc', B, D)
        mod = self.f32('mod', B, NM)
        f.add('mdt_timestep_embed_f32', cn.data_ptr(), temb.data_ptr(), 256, B, 256)
        self._g32(temb, 256, Pf('model.t_embedder.mlp.0.weight'), 256, B, D, 256, a1, D, bias=Pf('model.t_embedder.mlp.0.bias'),
                  epi=F32EPI_SILU)
        self._g32(a1, D, Pf('model.t_embedder.mlp.2.weight'), D, B, D, D, c_t, D, bias=Pf('model.t_embedder.mlp.2.bias'))
        # c = t_emb + y @ table^T (the one-hot / zero / soft label row times the embedding table)
        self._g32(lab, sp.num_classes, Pf('model.y_embedder.embedding_table.weight'), sp.num_classes, B, D, sp.num_classes, c, D,
                  epi=F32EPI_GATE_RES, res=c_t, ldres=D, rps=1)
        f.add('mdt_silu_f32', c.data_ptr(), sc.data_ptr(), B * D)
        self._g32(sc, D, Pp + 4 * lay.ada_w, D, B, NM, D, mod, NM, bias=Pp + 4 * lay.ada_b)
        # ---------------- encoder (all T tokens: masking applies in train mode only, models/maskdit.py:482,539) ---------
        Me = B * T
        x = self.f32('x_e0', Me, D)
        f.add('mdt_patch_embed_fwd', xin.data_ptr(), None, Pf('model.x_embedder.proj.weight'), Pf('model.x_embedder.proj.bias'),
              eng.pos.data_ptr(), None, 2 * T, x.data_ptr(), B, sp.C, sp.R, sp.patch, T, D)
        for i in range(sp.depth):
            x = self._block_fwd_f32(f'model.blocks.{i}', 'e', i, x, mod, sp.mod_off('enc', i), D, sp.heads, T, Me)
        self.marks = {'enc_fwd_end': len(f.calls)}
        # ---------------- DecoderLayer (:195-213) + decoder_pos_embed (:545) -------------------------------------------
        odl = sp.mod_off('dl')
        xnd, xdec = self.f32('xn_e', Me, D), self.f32('xdec', Me, Dd)
        f.add('mdt_ln_modulate_f32', x.data_ptr(), mod.data_ptr() + 4 * odl, mod.data_ptr() + 4 * (odl + D), NM, T, xnd.data_ptr(), Me, D)
        self._g32(xnd, D, Pf('model.decoder_layer.linear.weight'), D, Me, Dd, D, xdec, Dd, bias=Pf('model.decoder_layer.linear.bias'))
        x = self.f32('x_d0', Me, Dd)
        f.add('mdt_add_rows_f32', xdec.data_ptr(), eng.dpos.data_ptr(), x.data_ptr(), Me, T, Dd)
        for i in range(sp.ddepth):
            x = self._block_fwd_f32(f'model.decoder_blocks.{i}', 'd', i, x, mod, sp.mod_off('dec', i), Dd, sp.dheads, T, Me)
        ofin = sp.mod_off('fin')
        st_f = self.f32('st_f', Me, 2)
        f.add('mdt_final_fwd', x.data_ptr(), mod.data_ptr() + 4 * ofin, mod.data_ptr() + 4 * (ofin + Dd), NM,
              Pf('model.final_layer.linear.weight'), Pf('model.final_layer.linear.bias'), Fx.data_ptr(), st_f.data_ptr(),
              B, T, Dd, sp.C, sp.patch)

    def _g32(self, A, lda, Bw, ldb, M, N, K, out, ldo, bias=0, epi=F32EPI_NONE, res=None, ldres=0, gate=0, gate_ld=0, rps=1,
             b_kmajor=0, batch=0, heads=0, a_s=(0, 0), b_s=(0, 0), o_s=(0, 0)):
        """One mdt_gemm_f32 launch; A / out / res are tensors or raw addresses, Bw / bias / gate raw addresses."""
        ptr = lambda t: t if isinstance(t, int) else t.data_ptr()  # noqa: E731
        a = GemmF32Args()
        a.A, a.lda, a.B, a.ldb, a.b_kmajor = ptr(A), lda, Bw, ldb, b_kmajor
        a.M, a.N, a.K = M, N, K
        a.bias, a.epi = bias or None, epi
        a.out, a.ldo = ptr(out), ldo
        a.res, a.ldres = (ptr(res) if res is not None else None), ldres
        a.gate, a.gate_ld, a.rows_per_sample = gate or None, gate_ld, rps
        a.batch, a.heads = batch, heads
        a.a_stride_b, a.a_stride_h = a_s
        a.b_stride_b, a.b_stride_h = b_s
        a.o_stride_b, a.o_stride_h = o_s
        self.fwd.add('mdt_gemm_f32', C.byref(self._k(a)))

    def _block_fwd_f32(self, prefix, tag, i, x_in, mod, moff, W, heads, rows, M):
        """DiTBlock.forward (models/maskdit.py:188-192) in fp32; all blocks of a stack share one buffer set."""
        eng, lay, f = self.eng, self.eng.lay, self.fwd
        NM = eng.sp.n_mod
        hd = W // heads
        B = self.B
        Pp = eng.P.data_ptr()
        Pf = lambda n: Pp + 4 * lay.off[f'{prefix}.{n}']  # noqa: E731
        mp = mod.data_ptr()
        sh1, sc1, g1, sh2, sc2, g2 = (mp + 4 * (moff + k * W) for k in range(6))
        xn = self.f32(f'xn_{tag}', M, W)
        qkv = self.f32(f'qkv_{tag}', M, 3 * W)
        ws_floats = int(_lib.lib().mdt_attn_f32_ws_floats(B, rows, heads, hd))  # 0: the fused single-launch kernel serves this shape
        S = self.f32(f'scores_{tag}', ws_floats) if ws_floats else None
        ao = self.f32(f'ao_{tag}', M, W)
        xmid = self.f32(f'xmid_{tag}', M, W)
        h = self.f32(f'h_{tag}', M, 4 * W)
        xout = self.f32(f'x_{tag}pp{(i + 1) % 2}', M, W)
        f.add('mdt_ln_modulate_f32', x_in.data_ptr(), sh1, sc1, NM, rows, xn.data_ptr(), M, W)
        self._g32(xn, W, Pf('attn.qkv.weight'), W, M, 3 * W, W, qkv, 3 * W, bias=Pf('attn.qkv.bias'))
        # timm Attention: softmax(q k^T * hd^-0.5) v per (sample, head) on the packed [M, (3, heads, hd)] buffer
        f.add('mdt_attn_f32', qkv.data_ptr(), ao.data_ptr(), S.data_ptr() if S is not None else None, B, rows, heads, hd)
        self._g32(ao, W, Pf('attn.proj.weight'), W, M, W, W, xmid, W, bias=Pf('attn.proj.bias'), epi=F32EPI_GATE_RES,
                  res=x_in, ldres=W, gate=g1, gate_ld=NM, rps=rows)
        f.add('mdt_ln_modulate_f32', xmid.data_ptr(), sh2, sc2, NM, rows, xn.data_ptr(), M, W)
        self._g32(xn, W, Pf('mlp.fc1.weight'), W, M, 4 * W, W, h, 4 * W, bias=Pf('mlp.fc1.bias'), epi=F32EPI_GELU)
        self._g32(h, 4 * W, Pf('mlp.fc2.weight'), 4 * W, M, W, 4 * W, xout, W, bias=Pf('mlp.fc2.bias'), epi=F32EPI_GATE_RES,
                  res=xmid, ldres=W, gate=g2, gate_ld=NM, rps=rows)
        return xout

    def _k(self, obj):
        self.fwd.keep.append(obj)
        return obj

    def _slab(self, name):
        ref = self._eng_ref
        lo, hi = self.eng.lay.slabs[name]

        def cb():
            eng = ref()
            if eng is not None and eng.grad_slab_hook is not None:
                eng.grad_slab_hook(name, lo, hi)

        self.bwd.add_callback(cb)

    # ---- one DiT block ---------------------------------------------------------------------
    def _block_fwd(self, prefix, tag, i, x_in, mod, moff, W, heads, rows, M, lvalid=0, pending=None, defer_out=False):
        """DiTBlock.forward (models/maskdit.py:188-192) as 7 launches.  `pending` = (xres, y, gate) of the previous block's
        MLP branch: this block's first LayerNorm pass then forms x_in = xres + gate * y itself (FUSE_RES_LN); `defer_out`
        hands this block's own MLP residual to whoever normalises its output next.  Returns (x_out buffer, pending)."""
        eng, lay, f = self.eng, self.eng.lay, self.fwd
        NM = eng.sp.n_mod
        hd = W // heads
        B = self.B
        s = f'{tag}{i}' if self.train else f'{tag}'  # eval: all blocks share one buffer set
        Pp, W16p = eng.P.data_ptr(), eng.W16.data_ptr()
        Pf = lambda n: Pp + 4 * lay.off[f'{prefix}.{n}']  # noqa: E731
        Wp = lambda n: W16p + 2 * lay.off[f'{prefix}.{n}']  # noqa: E731
        mp = mod.data_ptr()
        sh1, sc1, g1, sh2, sc2, g2 = (mp + 4 * (moff + k * W) for k in range(6))
        xn1, st1 = self.b16(f'xn1_{s}', M, W), self.f32(f'st1_{s}', M, 2)
        qkv = self.b16(f'qkv_{s}', M, 3 * W)
        ao, lse = self.b16(f'ao_{s}', M, W), self.f32(f'lse_{s}', B * heads * rows)
        ya = self.b16(f'ya_{s}', M, W)
        xmid = self.f32(f'xmid_{s}', M, W)
        xn2, st2 = self.b16(f'xn2_{s}', M, W), self.f32(f'st2_{s}', M, 2)
        h, a = self.b16(f'h_{s}', M, 4 * W), self.b16(f'a_{s}', M, 4 * W)
        ym = self.b16(f'ym_{s}', M, W)
        if self.train:
            xout = self.f32(f'x_{tag}{i + 1}', M, W)
        else:  # ping-pong
            xout = self.f32(f'x_{tag}pp{(i + 1) % 2}', M, W)
        tr = self.train
        fuse = FUSE_RES_LN and (tr or FUSE_RES_LN_EVAL)
        if pending is not None:  # x_in = xres + gate * y (the previous block's MLP residual) is formed by this pass
            f.add('mdt_ln_modulate_fwd_res', pending[0].data_ptr(), pending[1].data_ptr(), pending[2], NM, sh1, sc1, NM, rows,
                  x_in.data_ptr(), xn1.data_ptr(), st1.data_ptr(), M, W)
        else:
            f.add('mdt_ln_modulate_fwd', x_in.data_ptr(), sh1, sc1, NM, rows, xn1.data_ptr(), st1.data_ptr(), M, W)
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(xn1.data_ptr(), W, Wp('attn.qkv.weight'), W, M, 3 * W, W, bias=Pf('attn.qkv.bias'),
                                               epi=EPI_BF16, out=qkv.data_ptr(), ldo=3 * W))))
        f.add('mdt_attn_fwd', qkv.data_ptr(), ao.data_ptr(), lse.data_ptr(), B, rows, heads, hd, lvalid)
        # the bf16 copies of the branch outputs (ya, ym) and the pre-activation h are saved for the
        # backward only: inference plans skip those stores (2 of 10 resp. 2 of 4 epilogue bytes / element)
        if fuse:  # the GEMM stores y only; x_mid = x_in + g1 * y is formed by the LayerNorm pass that consumes it
            f.add('mdt_gemm_nt', C.byref(self._k(_nt(ao.data_ptr(), W, Wp('attn.proj.weight'), W, M, W, W, bias=Pf('attn.proj.bias'),
                                                   epi=EPI_BF16, out=ya.data_ptr(), ldo=W))))
            f.add('mdt_ln_modulate_fwd_res', x_in.data_ptr(), ya.data_ptr(), g1, NM, sh2, sc2, NM, rows, xmid.data_ptr(),
                  xn2.data_ptr(), st2.data_ptr(), M, W)
        else:
            f.add('mdt_gemm_nt', C.byref(self._k(_nt(ao.data_ptr(), W, Wp('attn.proj.weight'), W, M, W, W, bias=Pf('attn.proj.bias'),
                                                   epi=EPI_GATE_RES, out=ya.data_ptr() if tr else 0, ldo=W, outf=xmid.data_ptr(), ldof=W,
                                                   res=x_in.data_ptr(), ldres=W, gate=g1, gate_ld=NM, rps=rows))))
            f.add('mdt_ln_modulate_fwd', xmid.data_ptr(), sh2, sc2, NM, rows, xn2.data_ptr(), st2.data_ptr(), M, W)
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(xn2.data_ptr(), W, Wp('mlp.fc1.weight'), W, M, 4 * W, W, bias=Pf('mlp.fc1.bias'),
                                               epi=EPI_GELU, out=h.data_ptr() if tr else 0, ldo=4 * W, out2=a.data_ptr(), ldo2=4 * W))))
        if fuse and defer_out:
            f.add('mdt_gemm_nt', C.byref(self._k(_nt(a.data_ptr(), 4 * W, Wp('mlp.fc2.weight'), 4 * W, M, W, 4 * W, bias=Pf('mlp.fc2.bias'),
                                                   epi=EPI_BF16, out=ym.data_ptr(), ldo=W))))
            return xout, (xmid, ym, g2)
        f.add('mdt_gemm_nt', C.byref(self._k(_nt(a.data_ptr(), 4 * W, Wp('mlp.fc2.weight'), 4 * W, M, W, 4 * W, bias=Pf('mlp.fc2.bias'),
                                               epi=EPI_GATE_RES, out=ym.data_ptr() if tr else 0, ldo=W, outf=xout.data_ptr(), ldof=W,
                                               res=xmid.data_ptr(), ldres=W, gate=g2, gate_ld=NM, rps=rows))))
        return xout, None

    def _gate_info(self, prefix, tag, i, mod, dmod, moff, W, Gf):
        """Trailing arguments of mdt_ln_modulate_bwd_gate for the MLP residual gate of block (tag, i):
        (y = fc2 output, gate, gate_ld, dys, dgate, dgate_ld, dbias)."""
        NM = self.eng.sp.n_mod
        ym = self.buf[f'ym_{tag}{i}']
        return (ym.data_ptr(), mod.data_ptr() + 4 * (moff + 5 * W), NM, self._ws['dys'].data_ptr(),
                dmod.data_ptr() + 4 * (moff + 5 * W), NM, Gf(f'{prefix}.mlp.fc2.bias'))

    def _block_bwd(self, prefix, tag, i, x_in, mod, dmod, moff, W, heads, rows, M, dx, Gf, fuse_next=None, skip_first_gate=False,
                   lvalid=0):
        """Backward of DiTBlock: dx (fp32 residual-stream gradient) is updated in place.  The gate
        backward that opens the block is folded into the LayerNorm backward that precedes it in stream
        order (`skip_first_gate`), and the block's last LayerNorm backward carries the next block's
        (`fuse_next`)."""
        eng, lay, g, ws = self.eng, self.eng.lay, self.bwd, self._ws
        NM = eng.sp.n_mod
        hd = W // heads
        B = self.B
        s = f'{tag}{i}'
        W16p, WTp = eng.W16.data_ptr(), eng.WT16.data_ptr()
        WT = lambda n: WTp + 2 * lay.t_off[f'{prefix}.{n}']  # noqa: E731
        Gn = lambda n: Gf(f'{prefix}.{n}')  # noqa: E731
        mp, dp = mod.data_ptr(), dmod.data_ptr()
        sc1, g1, sc2, g2 = mp + 4 * (moff + W), mp + 4 * (moff + 2 * W), mp + 4 * (moff + 4 * W), mp + 4 * (moff + 5 * W)
        dsh1, dsc1, dg1, dsh2, dsc2, dg2 = (dp + 4 * (moff + k * W) for k in range(6))
        b = self.buf
        xn1, st1, qkv, ao, lse = b[f'xn1_{s}'], b[f'st1_{s}'], b[f'qkv_{s}'], b[f'ao_{s}'], b[f'lse_{s}']
        ya, xmid, xn2, st2, h, a, ym = b[f'ya_{s}'], b[f'xmid_{s}'], b[f'xn2_{s}'], b[f'st2_{s}'], b[f'h_{s}'], b[f'a_{s}'], b[f'ym_{s}']
        dys, dh, dxn, dao, dqkv, delta = (ws[k].data_ptr() for k in ('dys', 'dh', 'dxn', 'dao', 'dqkv', 'delta'))
        dxp = dx.data_ptr()
        K = self._k
        # --- MLP branch: x_out = x_mid + g2 * (fc2(gelu(fc1(xn2))))
        if not skip_first_gate:
            g.add('mdt_gate_bwd', dxp, ym.data_ptr(), g2, NM, rows, dys, dg2, NM, Gn('mlp.fc2.bias'), M, W)
        g.add('mdt_gemm_tn', C.byref(K(_tn(dys, W, a.data_ptr(), 4 * W, M, W, 4 * W, Gn('mlp.fc2.weight'), 4 * W))))
        # dh = (dys W2) * gelu'(h); its column sums (= d fc1.bias) come out of the same epilogue
        g.add('mdt_gemm_nt', C.byref(K(_nt(dys, W, WT('mlp.fc2.weight'), W, M, 4 * W, W, epi=EPI_DGELU, out=dh, ldo=4 * W,
                                         aux=h.data_ptr(), ldaux=4 * W, colsum=Gn('mlp.fc1.bias') if FUSE_COLSUM else 0))))
        g.add('mdt_gemm_tn', C.byref(K(_tn(dh, 4 * W, xn2.data_ptr(), W, M, 4 * W, W, Gn('mlp.fc1.weight'), W))))
        if not FUSE_COLSUM:
            g.add('mdt_colsum_bf16', dh, 4 * W, Gn('mlp.fc1.bias'), M, 4 * W)
        g.add('mdt_gemm_nt', C.byref(K(_nt(dh, 4 * W, WT('mlp.fc1.weight'), 4 * W, M, W, 4 * W, epi=EPI_BF16, out=dxn, ldo=W))))
        # --- attention branch: x_mid = x_in + g1 * proj(attn(qkv(xn1)))
        if FUSE_LN_GATE:  # the gate backward rides on LN2's backward
            g.add('mdt_ln_modulate_bwd_gate', dxn, xmid.data_ptr(), st2.data_ptr(), sc2, NM, rows, dxp, 1, dsh2, dsc2, NM, M, W,
                  ya.data_ptr(), g1, NM, dys, dg1, NM, Gn('attn.proj.bias'))
        else:
            g.add('mdt_ln_modulate_bwd', dxn, xmid.data_ptr(), st2.data_ptr(), sc2, NM, rows, dxp, 1, dsh2, dsc2, NM, M, W)
            g.add('mdt_gate_bwd', dxp, ya.data_ptr(), g1, NM, rows, dys, dg1, NM, Gn('attn.proj.bias'), M, W)
        g.add('mdt_gemm_tn', C.byref(K(_tn(dys, W, ao.data_ptr(), W, M, W, W, Gn('attn.proj.weight'), W))))
        g.add('mdt_gemm_nt', C.byref(K(_nt(dys, W, WT('attn.proj.weight'), W, M, W, W, epi=EPI_BF16, out=dao, ldo=W))))
        g.add('mdt_attn_bwd', qkv.data_ptr(), ao.data_ptr(), dao, lse.data_ptr(), delta, dqkv, B, rows, heads, hd, lvalid)
        # the qkv bias gradient (column sums of dqkv) comes out of the weight-gradient GEMM, which streams dqkv through
        # LDS anyway (mdt_gemm_tn_args.colsum_a): one pass over dqkv less per block
        g.add('mdt_gemm_tn', C.byref(K(_tn(dqkv, 3 * W, xn1.data_ptr(), W, M, 3 * W, W, Gn('attn.qkv.weight'), W,
                                          colsum_a=Gn('attn.qkv.bias') if FUSE_QKV_COLSUM else 0))))
        if not FUSE_QKV_COLSUM:
            g.add('mdt_colsum_bf16', dqkv, 3 * W, Gn('attn.qkv.bias'), M, 3 * W)
        g.add('mdt_gemm_nt', C.byref(K(_nt(dqkv, 3 * W, WT('attn.qkv.weight'), 3 * W, M, W, 3 * W, epi=EPI_BF16, out=dxn, ldo=W))))
        if fuse_next is not None:
            g.add('mdt_ln_modulate_bwd_gate', dxn, x_in.data_ptr(), st1.data_ptr(), sc1, NM, rows, dxp, 1, dsh1, dsc1, NM, M, W,
                  *fuse_next)
        else:
            g.add('mdt_ln_modulate_bwd', dxn, x_in.data_ptr(), st1.data_ptr(), sc1, NM, rows, dxp, 1, dsh1, dsc1, NM, M, W)

    # ---- execution ---------------------------------------------------------------------------
    def run_forward(self) -> int:
        if self.eng.shadows_dirty and self.precision != 'fp32':  # (an fp32 plan reads the master arena itself)
            self.eng.refresh_shadows()
        self.gen += 1
        self.fwd.run(torch.cuda.current_stream().cuda_stream)
        return self.gen

    def run_backward(self, gen: Optional[int] = None):
        """`gen` = the value run_forward returned for the forward being differentiated.  The activations live in
        this plan's buffers, not in the autograd graph: a later forward of the same shape overwrites them, and a
        backward of the EARLIER forward would silently differentiate the wrong activations -- so it raises."""
        if gen is not None and gen != self.gen:
            raise RuntimeError(
                'maskdit_amd: backward of a forward whose saved activations were overwritten by a later forward of the '
                f'same shape (forward #{gen}, buffers now hold #{self.gen}).  Call backward() before the next forward '
                '(gradient accumulation: one forward/backward per micro-batch), or evaluate under torch.no_grad().')
        self.bwd.run(torch.cuda.current_stream().cuda_stream)
