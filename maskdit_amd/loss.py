"""Drop-in for `train_utils/loss.py`: `Losses['edm']` -> EDMLoss (train_utils/loss.py:22-60)
with the MAE term (:88-101), fused end to end in HIP when the network is a maskdit_amd
EDMPrecond:

    randn draws (torch, same order as the reference: rnd_normal then randn_like, loss.py:35,39)
    -> mdt_edm_prep (sigma, EDM coefficients, y+n, c_in*(y+n))            loss.py:36-39, maskdit.py:764-767
    -> get_mask (torch.rand + HIP bitonic argsort)                         maskdit.py:88-113
    -> DiT forward plan                                                    maskdit.py:467-557
    -> mdt_edm_loss_fwd (D = c_skip*x + c_out*F, weighted MSE, patch pool,
       unmasked mean, + mae_coef * MAE on masked patches) -> loss [N]      loss.py:44-52
    backward: mdt_edm_loss_bwd -> DiT backward plan (gradients land in the arena).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import call
from .precond import EDMPrecond, _fill_plan_inputs, _ids32_from_dict, _stream, get_mask


def unwrap_model(model):
    """train_utils/helper.py:61-69, extended to this package's DataParallel wrapper."""
    mod = getattr(torch, '_dynamo', None)
    if mod is not None and isinstance(model, torch._dynamo.eval_frame.OptimizedModule):
        model = model._orig_mod
    while hasattr(model, 'module') and not isinstance(model, EDMPrecond):
        model = model.module
    return model


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: EDMPrecond, images, labels, rnd_normal, noise, mask_dict, mae_loss_coef, P_mean, P_std, need_grad, anchor):
        eng = net.engine()
        sp = net.spec
        B = images.shape[0]
        chw = sp.C * sp.R * sp.R
        masked = mask_dict is not None
        L = mask_dict['ids_keep'].shape[1] if masked else None
        need_grad = bool(need_grad)
        pl = eng.plan(B, masked, need_grad, L)
        st = _stream()
        yn = pl.f32('yn', B, sp.C, sp.R, sp.R)
        D = pl.f32('D', B, sp.C, sp.R, sp.R)
        ybuf = pl.f32('y', B, sp.C, sp.R, sp.R)
        ybuf.copy_(images)
        coef = pl.buf['coef']
        call('mdt_edm_prep', ybuf.data_ptr(), rnd_normal.data_ptr(), noise.data_ptr(), coef.data_ptr(), yn.data_ptr(),
             pl.buf['xin'].data_ptr(), B, chw, float(P_mean), float(P_std), float(net.sigma_data), st)
        _fill_plan_inputs(pl, labels, _ids32_from_dict(mask_dict, sp.T, L) if masked else None)
        ctx.gen = pl.run_forward()
        loss = torch.empty(B, device=images.device, dtype=torch.float32)
        mask = mask_dict['mask'].contiguous() if masked else None
        call('mdt_edm_loss_fwd', pl.buf['F'].data_ptr(), yn.data_ptr(), ybuf.data_ptr(), coef.data_ptr(),
             mask.data_ptr() if masked else None, float(mae_loss_coef) if masked else 0.0, D.data_ptr(), loss.data_ptr(),
             B, sp.C, sp.R, sp.patch, st)
        ctx.net, ctx.pl, ctx.mask, ctx.mae = net, pl, mask, (float(mae_loss_coef) if masked else 0.0)
        ctx.need_grad = need_grad
        return loss

    @staticmethod
    def backward(ctx, dloss):
        if not ctx.need_grad:
            raise RuntimeError('backward through a loss that was computed without gradient buffers')
        net, pl = ctx.net, ctx.pl
        sp = net.spec
        net._prepare_grad_arena()
        dloss = dloss.contiguous().float()
        b = pl.buf
        call('mdt_edm_loss_bwd', dloss.data_ptr(), b['D'].data_ptr(), b['yn'].data_ptr(), b['y'].data_ptr(), b['coef'].data_ptr(),
             ctx.mask.data_ptr() if ctx.mask is not None else None, ctx.mae, b['dF'].data_ptr(), pl.B, sp.C, sp.R, sp.patch,
             _stream())
        pl.run_backward(ctx.gen)
        return (None,) * 11


class EDMLoss:
    """train_utils/loss.py:22-60.  `net` may be the bare EDMPrecond, this package's
    DataParallel wrapper, or anything exposing `.module` (the reference requires a DDP-like
    wrapper, loss.py:47; that restriction is not reproduced)."""

    def __init__(self, P_mean=-1.2, P_std=1.2, sigma_data=0.5):
        self.P_mean = P_mean
        self.P_std = P_std
        self.sigma_data = sigma_data

    def __call__(self, net, images, labels=None, mask_ratio=0, mae_loss_coef=0, feat=None, augment_pipe=None):
        raw = unwrap_model(net)
        if not isinstance(raw, EDMPrecond):
            raise TypeError('maskdit_amd.Losses expects a maskdit_amd EDMPrecond (possibly wrapped); there is no '
                            f'eager fallback for {type(raw).__name__}')
        if feat is not None or augment_pipe is not None:
            raise NotImplementedError('feat / augment_pipe are outside the shipped configurations')
        if raw.sigma_data != self.sigma_data:
            raise ValueError('EDMLoss.sigma_data differs from the network sigma_data')
        if not images.is_cuda:
            raise _lib.MaskDiTLibError('maskdit_amd: images are not on a HIP device; there is no CPU path')
        B = images.shape[0]
        sp = raw.spec
        images = images.to(torch.float32).contiguous()
        # reference draw order (train_utils/loss.py:35,39): sigma noise first, then pixel noise
        rnd_normal = torch.randn([B, 1, 1, 1], device=images.device)
        noise = torch.randn_like(images)
        labels = raw._labels(labels, B, images.device).contiguous()
        mask_dict = None
        if mask_ratio > 0:
            assert raw.training, 'mask_ratio > 0 requires train mode (train_utils/loss.py:46)'
            mask_dict = get_mask(B, sp.T, mask_ratio, images.device)  # maskdit.py:476-477 (drawn after the noises)
        self.last_mask_dict = mask_dict
        anchor = raw._anchor()
        return _LossFn.apply(raw, images, labels, rnd_normal, noise, mask_dict, mae_loss_coef, self.P_mean, self.P_std,
                             torch.is_grad_enabled() and anchor is not None, anchor)

    # entry used by parity tests: identical arithmetic with the random draws supplied
    def with_draws(self, net, images, labels, rnd_normal, noise, mask_dict, mae_loss_coef=0):
        raw = unwrap_model(net)
        B = images.shape[0]
        labels = raw._labels(labels, B, images.device).contiguous()
        anchor = raw._anchor()
        return _LossFn.apply(raw, images.float().contiguous(), labels, rnd_normal.contiguous(), noise.contiguous(), mask_dict,
                             mae_loss_coef, self.P_mean, self.P_std, torch.is_grad_enabled() and anchor is not None, anchor)


Losses = {'edm': EDMLoss}
