"""Drop-ins for `apex.optimizers.FusedAdam` (train.py:14,141,226) and
`train_utils.helper.update_ema` (train_utils/helper.py:47-58).

When the parameters are the complete trainable set of an engine-bound EDMPrecond (the normal
case: `FusedAdam(model.parameters(), ...)`), `step()` is ONE streaming HIP kernel over the flat
arenas -- p, g, exp_avg, exp_avg_sq (+ the EMA arena when `fuse_ema` was called, + the bf16
GEMM-operand shadow) -- followed by the batched transposes that refresh the K-major shadows.
Otherwise it runs the same kernel per tensor.  There is no torch-eager fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import call
from .engine import LIVE_ENGINES, Engine


def _engine_of(p: torch.Tensor) -> Optional[Engine]:
    ptr = p.data_ptr()
    for eng in list(LIVE_ENGINES):
        lo = eng.P.data_ptr()
        if lo <= ptr < lo + 4 * eng.lay.n:
            return eng
    return None


def _st():
    return torch.cuda.current_stream().cuda_stream


class FusedAdam(torch.optim.Optimizer):
    """API of apex.optimizers.FusedAdam as the reference uses it: constructor keywords
    (`lr`, `betas`, `eps`, `adam_w_mode`, `weight_decay`, `bias_correction`), writable
    `param_groups[i]['lr']` (train.py:224-225), `step()`, `state_dict()/load_state_dict()`
    (train.py:153-157,264).  State layout follows apex: `group['step']` plus per-parameter
    `exp_avg` / `exp_avg_sq` (here: views into two flat moment arenas)."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True,
                 weight_decay=0.0, amsgrad=False, set_grad_none=True):
        if amsgrad:
            raise RuntimeError('FusedAdam does not support the AMSGrad variant.')
        defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if not adam_w_mode and weight_decay != 0:
            raise NotImplementedError('L2-regularisation mode (adam_w_mode=False) with weight_decay != 0')
        self.adam_w_mode = adam_w_mode
        self.set_grad_none = set_grad_none
        self.grad_scale = 1.0  # multiplied into the gradient inside the kernel (DP: 1/world for summed grads)
        self._ema = None  # (ema_model, decay) when fuse_ema() was called
        self._arena: Optional[Engine] = None
        self._m = self._v = None
        self._resolve_arena()

    # ---- layout ---------------------------------------------------------------------------
    def _resolve_arena(self):
        """Arena mode iff ONE param group holds exactly the trainable parameters of one engine."""
        self._arena = None
        if len(self.param_groups) != 1:
            return
        ps = self.param_groups[0]['params']
        trainable = [p for p in ps if p.requires_grad]
        if not trainable or not trainable[0].is_cuda:
            return
        eng = _engine_of(trainable[0])
        if eng is None:
            return
        base = eng.P.data_ptr()
        want = {base + 4 * o for o in eng.lay.off.values()}
        have = {p.data_ptr() for p in trainable}
        if want == have:
            self._first = trainable[0]
            self._trainable = trainable
            self._arena = eng
            self._alloc_moments(eng)

    def _alloc_moments(self, eng):
        """Moment arenas of the engine's layout + apex-style per-parameter views (ShardedFusedAdam overrides)."""
        dev = eng.P.device
        base = eng.P.data_ptr()
        if self._m is None or self._m.numel() != eng.lay.n or self._m.device != dev:
            self._m = torch.zeros(eng.lay.n, device=dev, dtype=torch.float32)
            self._v = torch.zeros(eng.lay.n, device=dev, dtype=torch.float32)
            for p in self.param_groups[0]['params']:
                if p.requires_grad:
                    off = (p.data_ptr() - base) // 4
                    self.state[p] = {'exp_avg': self._m[off:off + p.numel()].view_as(p),
                                     'exp_avg_sq': self._v[off:off + p.numel()].view_as(p)}

    def zero_grad(self, set_to_none: Optional[bool] = None):
        super().zero_grad(set_to_none=self.set_grad_none if set_to_none is None else set_to_none)

    def fuse_ema(self, ema_model, decay=0.9999):
        """Fold `update_ema(ema_model, model, decay)` (train.py:230) into the optimizer kernel.
        The next `update_ema` call with the same arguments becomes a no-op for that step."""
        self._ema = (ema_model, float(decay))

    # ---- step -----------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._arena is not None and _engine_of(self._first) is not self._arena:
            self._resolve_arena()  # the module was moved / re-bound
        for group in self.param_groups:
            group['step'] = group.get('step', 0) + 1
            t = group['step']
            b1, b2 = group['betas']
            bc1 = 1 - b1 ** t if group['bias_correction'] else 1.0
            bc2 = 1 - b2 ** t if group['bias_correction'] else 1.0
            hyp = (float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']), float(bc1),
                   float(bc2))
            if self._arena is not None and group is self.param_groups[0]:
                # apex skips parameters without a gradient: the one-kernel arena step is only the same thing when
                # EVERY trainable parameter has one (or none has: nothing to do).  A mixed set -- the caller dropped
                # single gradients -- takes the per-tensor path for this step.
                n_none = sum(1 for p in self._trainable if p.grad is None)
                if n_none == 0:
                    self._step_arena(hyp)
                elif n_none < len(self._trainable):
                    self._step_mixed(group, hyp)
            else:
                self._step_tensors(group, hyp)
        return loss

    def _step_mixed(self, group, hyp):
        """Some (not all) gradients of an arena-bound model are None: per-tensor kernels on the parameters that have one."""
        if not getattr(self, '_warned_mixed', False):
            self._warned_mixed = True
            import warnings
            frozen = sum(1 for p in self._trainable if not p.requires_grad)
            warnings.warn(f'FusedAdam: {sum(1 for p in self._trainable if p.grad is None)} of {len(self._trainable)} arena parameters have no '
                          f'gradient ({frozen} of them no longer require one): this step -- and every step like it -- runs one kernel per '
                          'tensor (hundreds of launches, EMA and bf16 shadows refreshed by separate passes) instead of the single '
                          'arena kernel.  Results are the same; if the set is frozen for good, build the optimizer from the '
                          'parameters that still train.', RuntimeWarning, stacklevel=3)
        self._step_tensors(group, hyp)

    def _step_arena(self, hyp):
        eng = self._arena
        G = eng.G
        if G is None:
            return  # nothing was back-propagated (apex skips params without grad)
        ema_ptr, decay = None, 0.0
        if self._ema is not None:
            ema_eng = self._ema[0].engine()
            if ema_eng.lay.n != eng.lay.n:
                raise ValueError('fuse_ema: EMA model layout differs from the trained model')
            ema_ptr, decay = ema_eng.P.data_ptr(), self._ema[1]
        lr, b1, b2, eps, wd, bc1, bc2 = hyp
        call('mdt_adamw_ema_step', eng.P.data_ptr(), G.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), ema_ptr,
             eng.W16.data_ptr(), eng.lay.n, lr, b1, b2, eps, wd, bc1, bc2, decay, float(self.grad_scale), _st())
        eng.refresh_shadows(cast=False)  # K-major transposes + padded label table from the fresh bf16 shadow
        if self._ema is not None:
            ema_eng.shadows_dirty = True
            eng.ema_applied = (id(ema_eng), decay)

    def _step_tensors(self, group, hyp):
        lr, b1, b2, eps, wd, bc1, bc2 = hyp
        touched = set()
        for p in group['params']:
            if p.grad is None:
                continue
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.MaskDiTLibError('FusedAdam: parameters must be contiguous fp32 tensors on a HIP device')
            st = self.state[p]
            if 'exp_avg' not in st:
                st['exp_avg'] = torch.zeros_like(p)
                st['exp_avg_sq'] = torch.zeros_like(p)
            g = p.grad.contiguous()
            call('mdt_adamw_ema_step', p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), None,
                 None, p.numel(), lr, b1, b2, eps, wd, bc1, bc2, 0.0, float(self.grad_scale), _st())
            eng = _engine_of(p)
            if eng is not None:
                touched.add(eng)
        for eng in touched:
            eng.shadows_dirty = True

    # ---- checkpoint -----------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """In-place restore so the moment views keep pointing into the arenas (train.py:153-157)."""
        groups = state_dict['param_groups']
        if len(groups) != len(self.param_groups):
            raise ValueError('loaded state dict has a different number of parameter groups')
        with torch.no_grad():
            for g_new, g in zip(groups, self.param_groups):
                if len(g_new['params']) != len(g['params']):
                    raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
                for k, v in g_new.items():
                    if k != 'params':
                        g[k] = v
                for idx, p in zip(g_new['params'], g['params']):
                    src = state_dict['state'].get(idx)
                    if src is None:
                        continue
                    st = self.state[p]
                    for key in ('exp_avg', 'exp_avg_sq'):
                        if key not in st:
                            st[key] = torch.zeros_like(p)
                        st[key].copy_(src[key])
                    if 'step' in src and 'step' not in g_new:  # torch.optim.AdamW-style checkpoint
                        g['step'] = int(src['step'])


@torch.no_grad()
def update_ema(ema_model, model, decay=0.9999):
    """train_utils/helper.py:47-58: ema = decay * ema + (1 - decay) * p for every trainable
    parameter.  One kernel over the arenas when both sides are engine-bound EDMPrecond models
    with all arena parameters trainable; per-tensor kernels otherwise."""
    from .loss import unwrap_model
    from .precond import EDMPrecond
    model = unwrap_model(model)
    ema_model = unwrap_model(ema_model)
    if isinstance(model, EDMPrecond) and isinstance(ema_model, EDMPrecond):
        eng, ema_eng = model.engine(), ema_model.engine()
        applied = getattr(eng, 'ema_applied', None)
        if applied is not None:
            eng.ema_applied = None
            if applied == (id(ema_eng), float(decay)):
                return  # already folded into FusedAdam.step() for this step
        if eng.lay.n == ema_eng.lay.n and all(p.requires_grad for n, p in model.named_parameters() if n in eng.lay.off):
            call('mdt_ema_update', ema_eng.P.data_ptr(), eng.P.data_ptr(), eng.lay.n, float(decay), _st())
            ema_eng.shadows_dirty = True
            return
    ema_params = dict(ema_model.named_parameters())
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        e = ema_params[name.replace('_orig_mod.', '')]
        if not (p.is_cuda and e.is_cuda and p.is_contiguous() and e.is_contiguous()):
            raise _lib.MaskDiTLibError('update_ema: parameters must be contiguous tensors on a HIP device')
        call('mdt_ema_update', e.data_ptr(), p.data_ptr(), p.numel(), float(decay), _st())
        eng = _engine_of(e)
        if eng is not None:
            eng.shadows_dirty = True
