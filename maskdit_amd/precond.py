"""Drop-in for the reference's `models/maskdit.py` object surface: `Precond_models['edm']`
(EDMPrecond, models/maskdit.py:722-781) wrapping a `DiT` (models/maskdit.py:242-587).

Both are `nn.Module`s carrying parameters under the REFERENCE state-dict names and shapes
(`model.blocks.0.attn.qkv.weight [3D, D]`, `model.x_embedder.proj.weight [D, C, p, p]`, ...), so
published `.pt` checkpoints load with `load_state_dict`, `deepcopy` gives an EMA copy,
`named_parameters()` feeds `update_ema`, and an optimizer sees ordinary Parameters.  The
arithmetic, however, is not torch: once the module sits on a HIP device every parameter is a
VIEW into the engine's flat fp32 arena (maskdit_amd/engine.py) and `forward` replays a
pre-bound plan of libmaskdit_hip.so launches.  `.grad` of each parameter is a view into the
gradient arena that the hand-written backward fills.  There is no CPU / eager fallback:
calling the module without the library or off-GPU raises.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import call
from .engine import MODEL_CONFIGS, Engine, Spec, make_spec


# ------------------------------------------------------------------------------------------
# fixed 2-D sin/cos positional table (models/maskdit.py:595-642)

def _sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    # models/maskdit.py:624-642: omega_i = 10000^(-i/(dim/2)); [sin | cos]
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    ang = pos.reshape(-1)[:, None] * omega[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_pos_embed(dim: int, grid: int) -> torch.Tensor:
    """[T, dim] float32.  The first dim/2 channels encode the w coordinate: the reference
    builds meshgrid(grid_w, grid_h) (models/maskdit.py:603) and embeds grid[0] first (:617)."""
    ax = np.arange(grid, dtype=np.float32)
    gw, gh = np.meshgrid(ax, ax)  # gw[i, j] = j (w index), gh[i, j] = i
    emb = np.concatenate([_sincos_1d(dim // 2, gw), _sincos_1d(dim // 2, gh)], axis=1)
    return torch.from_numpy(emb).float()


# ------------------------------------------------------------------------------------------
# masking (models/maskdit.py:88-113)

def get_mask(batch: int, length: int, mask_ratio: float, device, noise: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Same contract as the reference get_mask: draws `torch.rand(batch, length)` on `device`
    (same philox stream position as the reference would use) and sorts it with the HIP
    bitonic kernel.  Tie rule = stable (lower index first).  Extra key 'ids32' (int32
    [B, 2T] = shuffle | restore) is what the gather / scatter kernels consume."""
    len_keep = int(length * (1 - mask_ratio))
    if noise is None:
        noise = torch.rand(batch, length, device=device)
    noise = noise.contiguous()
    dev = noise.device
    ids_shuffle = torch.empty(batch, length, device=dev, dtype=torch.int64)
    ids_restore = torch.empty(batch, length, device=dev, dtype=torch.int64)
    mask = torch.empty(batch, length, device=dev, dtype=torch.float32)
    ids32 = torch.empty(batch, 2 * length, device=dev, dtype=torch.int32)
    call('mdt_mask_sort', noise.data_ptr(), batch, length, len_keep, ids_shuffle.data_ptr(), ids_restore.data_ptr(),
         mask.data_ptr(), ids32.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return {'mask': mask, 'ids_keep': ids_shuffle[:, :len_keep], 'ids_restore': ids_restore, 'ids32': ids32}


def _ids32_from_dict(mask_dict, T: int, L: int) -> torch.Tensor:
    """A caller-supplied mask_dict (reference keys only) -> the int32 [B, 2T] table.  The
    shuffle half needs the removed ids too (order among them is irrelevant to the result, it
    only has to be the inverse of ids_restore), so it is rebuilt from ids_restore."""
    if 'ids32' in mask_dict:
        return mask_dict['ids32']
    restore = mask_dict['ids_restore']
    B = restore.shape[0]
    shuffle = torch.empty_like(restore)
    shuffle.scatter_(1, restore, torch.arange(T, device=restore.device).expand(B, T))
    keep = mask_dict['ids_keep']
    if keep.shape[1] != L or not torch.equal(shuffle[:, :L], keep):
        raise AssertionError('mask_dict: ids_keep is not the first len_keep entries of argsort(ids_restore)')
    return torch.cat([shuffle, restore], dim=1).to(torch.int32).contiguous()


# ------------------------------------------------------------------------------------------

class _Node(nn.Module):
    """Name-space node so that parameters get the reference's dotted state-dict keys.  Nodes whose children are
    numbered (`blocks`, `decoder_blocks`, the `mlp` / `adaLN_modulation` Sequentials) index like the reference's
    ModuleList / Sequential: `model.blocks[3].attn.qkv.weight`."""

    def __getitem__(self, i):
        return self._modules[str(i if i >= 0 else len(self._modules) + i)]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())


def _attach(root: nn.Module, dotted: str, p: nn.Parameter):
    parts = dotted.split('.')
    node = root
    for k in parts[:-1]:
        if k not in node._modules:
            node.add_module(k, _Node())
        node = node._modules[k]
    node.register_parameter(parts[-1], p)


def reference_param_order(spec: Spec):
    """Names (below `model.`) in the order `parameters()` yields them on the reference DiT."""
    def block(prefix):
        return [f'{prefix}.{n}' for n in ('attn.qkv.weight', 'attn.qkv.bias', 'attn.proj.weight', 'attn.proj.bias',
                                          'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias',
                                          'adaLN_modulation.1.weight', 'adaLN_modulation.1.bias')]
    names = ['pos_embed', 'decoder_pos_embed'] + (['mask_token'] if spec.mae else [])
    names += ['x_embedder.proj.weight', 'x_embedder.proj.bias', 't_embedder.mlp.0.weight', 't_embedder.mlp.0.bias',
              't_embedder.mlp.2.weight', 't_embedder.mlp.2.bias', 'y_embedder.embedding_table.weight']
    for i in range(spec.depth):
        names += block(f'blocks.{i}')
    names += ['decoder_layer.linear.weight', 'decoder_layer.linear.bias', 'decoder_layer.adaLN_modulation.1.weight',
              'decoder_layer.adaLN_modulation.1.bias']
    for i in range(spec.ddepth):
        names += block(f'decoder_blocks.{i}')
    names += ['final_layer.linear.weight', 'final_layer.linear.bias', 'final_layer.adaLN_modulation.1.weight',
              'final_layer.adaLN_modulation.1.bias']
    return names


class DiT(nn.Module):
    """Parameter container with the reference DiT's names, shapes, init distributions
    (models/maskdit.py:242-409) and the attributes the reference callers read
    (`patch_size`, `out_channels`, `extras`, `cls_token`, `mask_token`: train.py:138,
    train_utils/loss.py:47,57,89).  Supported flag set = the shipped configs: use_decoder,
    no cls token, no external features, no learn_sigma."""

    def __init__(self, spec: Spec):
        super().__init__()
        self.spec = spec
        self.patch_size = spec.patch
        self.in_channels = spec.C
        self.out_channels = spec.C
        self.num_classes = spec.num_classes
        self.extras = 0
        self.decoder_extras = 0
        self.cls_token = None
        self.use_decoder = True
        self.use_encoder_feat = False
        from .engine import param_table
        T, D, Dd = spec.T, spec.D, spec.Dd
        # REGISTRATION order = the reference module tree's `parameters()` order (models/maskdit.py:242-330: the
        # DiT's own parameters pos_embed, decoder_pos_embed, mask_token; then x_embedder, t_embedder, y_embedder,
        # blocks[i].{attn.qkv, attn.proj, mlp.fc1, mlp.fc2, adaLN_modulation.1}, decoder_layer.{linear, adaLN},
        # decoder_blocks, final_layer.{linear, adaLN}), because optimizer state dicts are positional
        # (train.py:141 hands `model.parameters()` to FusedAdam, :153/:264 save / load its state by index).  The
        # ARENA order (engine.param_table) is independent of it.
        shapes = {name[len('model.'):]: shp for name, shp in param_table(spec)}
        shapes['pos_embed'], shapes['decoder_pos_embed'] = (1, T, D), (1, T, Dd)
        for name in reference_param_order(spec):
            _attach(self, name, nn.Parameter(torch.zeros(shapes.pop(name)), requires_grad=name not in ('pos_embed', 'decoder_pos_embed')))
        assert not shapes, f'parameters missing from the registration order: {sorted(shapes)}'
        if not spec.mae:
            self.mask_token = None
        self.initialize_weights()

    @torch.no_grad()
    def initialize_weights(self):
        """models/maskdit.py:334-409: xavier-uniform Linears with zero bias; patch-embed
        xavier on the [D, C*p*p] view; N(0, .02) label table, timestep MLP and mask token;
        zero adaLN Linears, final_layer.linear and decoder_layer.linear; sincos tables."""
        sp = self.spec
        for name, p in self.named_parameters():
            if name in ('pos_embed', 'decoder_pos_embed'):
                continue
            if name.endswith('.bias') or 'adaLN_modulation' in name or name.startswith(('final_layer.linear', 'decoder_layer.linear')):
                p.zero_()
            elif name in ('y_embedder.embedding_table.weight', 't_embedder.mlp.0.weight', 't_embedder.mlp.2.weight', 'mask_token'):
                p.normal_(std=0.02)
            else:
                fan_out, fan_in = p.shape[0], int(np.prod(p.shape[1:]))
                a = math.sqrt(6.0 / (fan_in + fan_out))
                p.uniform_(-a, a)
        g = int(sp.T ** 0.5)
        self.pos_embed.copy_(sincos_pos_embed(sp.D, g).unsqueeze(0))
        self.decoder_pos_embed.copy_(sincos_pos_embed(sp.Dd, g).unsqueeze(0))


class EDMPrecond(nn.Module):
    """models/maskdit.py:722-776.  Constructor keywords follow the reference
    (`train.py:123-131`, `generate.py:31-40`)."""

    def __init__(self, img_resolution, img_channels, num_classes=0, sigma_min=0, sigma_max=float('inf'), sigma_data=0.5,
                 model_type='DiT-B/2', use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False, ext_feature_dim=0,
                 use_encoder_feat=False, direct_cls_token=False, learn_sigma=False, **unused):
        super().__init__()
        if pad_cls_token or ext_feature_dim or use_encoder_feat or direct_cls_token or learn_sigma:
            raise NotImplementedError('maskdit_amd accelerates the shipped flag set only: pad_cls_token=False, '
                                      'ext_feature_dim=0, use_encoder_feat=False, learn_sigma=False')
        if num_classes <= 0:
            raise NotImplementedError('unconditional models (num_classes=0) are outside the shipped configs')
        self._ctor = dict(img_resolution=img_resolution, img_channels=img_channels, num_classes=num_classes,
                          sigma_min=sigma_min, sigma_max=sigma_max, sigma_data=sigma_data, model_type=model_type,
                          use_decoder=use_decoder, mae_loss_coef=mae_loss_coef)
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.num_classes = num_classes
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.sigma_data = sigma_data
        self.spec = make_spec(model_type, img_resolution, img_channels, num_classes, use_decoder, mae_loss_coef)
        self.model = DiT(self.spec)
        self._engine: Optional[Engine] = None
        self._seen_version = -1
        self._plist = None
        self._grad_items = None  # (gradient arena, [(parameter, its arena view)]) -- see _prepare_grad_arena
        # arithmetic of INFERENCE evaluations (no-grad forward, forward_with_cfg, edm_sampler): 'bf16' = the training
        # kernels (bf16 MFMA operands, fp32 residual stream -- the reference under autocast); 'fp32' = exact fp32 throughout
        # (csrc/f32path.hip), what the reference's own sampler runs (sample.py:56, no autocast in generate.py).
        self.eval_precision = 'bf16'

    # ---- engine binding ------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        # .to()/.cuda()/.cpu(): torch re-creates every parameter's storage, so the arena views
        # are re-established afterwards (models/maskdit.py users call `.to(device)`, train.py:131)
        self._engine = None
        self._grad_items = None
        super()._apply(fn, *a, **k)
        p0 = next(self.parameters())
        if p0.is_cuda:
            self._bind(p0.device)
        return self

    def _bind(self, device):
        for p in self.parameters():
            if p.dtype != torch.float32:
                raise TypeError('maskdit_amd keeps fp32 master parameters (bf16 compute shadows are internal); '
                                f'got a {p.dtype} parameter')
        eng = Engine(self.spec, device)
        sp = self.spec
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name == 'model.pos_embed':
                    view = eng.pos.view(1, sp.T, sp.D)
                elif name == 'model.decoder_pos_embed':
                    view = eng.dpos.view(1, sp.T, sp.Dd)
                else:
                    view = eng.view(eng.P, name)
                view.copy_(p.data)
                p.data = view
                p.grad = None
        self._engine = eng
        self._seen_version = -1
        self._plist = None
        self._grad_items = None

    def engine(self) -> Engine:
        if self._engine is None:
            p0 = next(self.parameters())
            if not p0.is_cuda:
                raise _lib.MaskDiTLibError('maskdit_amd: the model is not on a HIP device (call .to("cuda")); '
                                           'there is no CPU path')
            self._bind(p0.device)
        eng = self._engine
        # any in-place torch write to a parameter (load_state_dict, the reference's update_ema, a foreign
        # optimizer) bumps THAT parameter's version counter (after `p.data = view` a Parameter keeps its own
        # counter, the arena's does not move) -> the bf16 / K-major shadows are stale.  The engine's own
        # optimizer kernel writes the arena and the shadows together and bumps nothing.
        plist = self._plist
        if plist is None:  # (the module tree is fixed after construction: walk it once, not on every forward)
            plist = self._plist = tuple(self.parameters())
        v = eng.P._version + eng.pos._version + eng.dpos._version + sum([p._version for p in plist])
        if v != self._seen_version:
            eng.shadows_dirty = True
            self._seen_version = v
        return eng

    def __deepcopy__(self, memo):
        # train.py:134 `ema = deepcopy(model)`: a fresh module + its own engine, same weights
        new = EDMPrecond(**self._ctor)
        p0 = next(self.parameters())
        new.to(p0.device)
        with torch.no_grad():
            src = dict(self.named_parameters())
            for name, p in new.named_parameters():
                p.copy_(src[name])
                p.requires_grad_(src[name].requires_grad)
        new.train(self.training)
        new.eval_precision = self.eval_precision
        memo[id(self)] = new
        return new

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)

    def set_eval_precision(self, precision: str):
        """'bf16' (default) or 'fp32' (the reference sampler's own arithmetic; ~1/8 of the bf16 throughput)."""
        if precision not in ('bf16', 'fp32'):
            raise ValueError(f"precision must be 'bf16' or 'fp32', got {precision!r}")
        self.eval_precision = precision
        return self

    # ---- gradient plumbing ---------------------------------------------------------------
    def _prepare_grad_arena(self):
        """Called right before a backward plan runs: the hand-written backward ACCUMULATES into the arena, matching
        autograd's `.grad +=`.  Decided PER PARAMETER (one sentinel tensor used to stand for all of
        them): a parameter whose `.grad` is None (`zero_grad(set_to_none=True)`, train.py:206, or a user dropping one
        gradient) gets its arena range cleared and its `.grad` re-pointed at it; a `.grad` that is some other tensor
        (assigned by the caller) is copied into the arena first and re-pointed, so that accumulation continues from it."""
        eng = self.engine()
        G = eng.ensure_grad()
        items = self._grad_items
        if items is None or items[0] is not G:  # (the module tree is fixed: walk it once per gradient arena)
            items = self._grad_items = (G, [(p, eng.view(G, name)) for name, p in self.named_parameters() if name in eng.lay.off])
        pairs = [pv for pv in items[1] if pv[0].requires_grad]  # requires_grad may be toggled between steps (finetuning)
        if not pairs:
            return G
        missing = [(p, v) for p, v in pairs if p.grad is None]
        if len(missing) == len(pairs):
            G.zero_()  # the usual case: one fill of the whole arena
        else:
            for _, v in missing:
                v.zero_()
            for p, v in pairs:
                g = p.grad
                if g is not None and g.data_ptr() != v.data_ptr():
                    v.copy_(g)
                    p.grad = v
        for p, v in missing:
            p.grad = v
        return G

    # ---- forward -------------------------------------------------------------------------
    def _labels(self, class_labels, B, device):
        if class_labels is None:
            return torch.zeros(B, self.num_classes, device=device)
        return class_labels.to(torch.float32).reshape(-1, self.num_classes)

    def forward(self, x, sigma, class_labels=None, cfg_scale=None, **model_kwargs):
        """EDMPrecond.forward (models/maskdit.py:756-773) -> {'x': D_x [, 'mask']}."""
        mask_ratio = model_kwargs.pop('mask_ratio', 0)
        mask_dict = model_kwargs.pop('mask_dict', None)
        feat = model_kwargs.pop('feat', None)
        if feat is not None or model_kwargs:
            raise NotImplementedError(f'unsupported model kwargs: feat / {sorted(model_kwargs)}')
        if not x.is_cuda:
            raise _lib.MaskDiTLibError('maskdit_amd: input is not on a HIP device; there is no CPU path')
        if x.dtype != torch.float32:
            raise TypeError(f'EDMPrecond.forward expects float32 input (sampler passes x.float()), got {x.dtype}')
        sp = self.spec
        assert x.shape[1:] == (sp.C, sp.R, sp.R), f'input shape {tuple(x.shape)} != [N,{sp.C},{sp.R},{sp.R}]'
        B = x.shape[0]
        sigma_b = torch.as_tensor(sigma, device=x.device).to(torch.float32).reshape(-1)
        if sigma_b.numel() == 1:
            sigma_b = sigma_b.expand(B)
        sigma_b = sigma_b.contiguous()
        assert sigma_b.numel() == B
        labels = self._labels(class_labels, B, x.device)
        out = {}
        if cfg_scale is not None:
            out['x'] = _run_cfg(self, x.contiguous(), sigma_b, labels, float(cfg_scale))
            return out
        ids32 = None
        L = None
        if mask_ratio > 0:
            if mask_dict is None:
                mask_dict = get_mask(B, sp.T, mask_ratio, x.device)
            out['mask'] = mask_dict['mask']
            if self.training:  # masking is applied in train mode only (models/maskdit.py:482,539)
                L = mask_dict['ids_keep'].shape[1]
                ids32 = _ids32_from_dict(mask_dict, sp.T, L)
        anchor = self._anchor()
        need_grad = torch.is_grad_enabled() and anchor is not None
        out['x'] = _NetFn.apply(self, x.contiguous(), sigma_b, labels.contiguous(), ids32, L, need_grad, anchor)
        return out

    def _engine_params(self):
        return [p for p in self.parameters() if p.requires_grad]

    def _anchor(self):
        # a leaf that requires grad so that autograd calls _NetFn.backward; parameter gradients
        # themselves are written by the HIP backward straight into the gradient arena
        ps = self._engine_params()
        return ps[0] if ps else None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _fill_plan_inputs(pl, labels, ids32):
    pl.buf['labels'].copy_(labels)
    if ids32 is not None:
        pl.buf['ids32'].copy_(ids32)


class _NetFn(torch.autograd.Function):
    """One EDMPrecond evaluation (no CFG): coefficients -> DiT plan -> D_x."""

    @staticmethod
    def forward(ctx, net: EDMPrecond, x, sigma, labels, ids32, L, need_grad, anchor):
        eng = net.engine()
        sp = net.spec
        B = x.shape[0]
        chw = sp.C * sp.R * sp.R
        masked = ids32 is not None
        prec = net.eval_precision if not (need_grad or masked) else 'bf16'  # training / masked forwards: bf16 kernels only
        pl = eng.plan(B, masked, bool(need_grad), L, prec)
        st = _stream()
        coef = pl.buf['coef']
        call('mdt_precond_coef', sigma.data_ptr(), coef.data_ptr(), B, float(net.sigma_data), st)
        call('mdt_scale_rows', x.data_ptr(), coef.data_ptr(), 2, pl.buf['xin'].data_ptr(), B, chw, st)
        _fill_plan_inputs(pl, labels, ids32)
        ctx.gen = pl.run_forward()
        D = torch.empty_like(x)
        call('mdt_precond_out', x.data_ptr(), pl.buf['F'].data_ptr(), coef.data_ptr(), D.data_ptr(), B, chw, st)
        ctx.net, ctx.pl = net, pl
        ctx.need_grad = bool(need_grad)
        return D

    @staticmethod
    def backward(ctx, dD):
        if not ctx.need_grad:
            raise RuntimeError('backward through a forward that ran without gradient buffers')
        net, pl = ctx.net, ctx.pl
        sp = net.spec
        B = pl.B
        net._prepare_grad_arena()
        dD = dD.contiguous()
        # D = c_skip x + c_out F  =>  dF = c_out dD   (gradient w.r.t. x is not produced)
        call('mdt_scale_rows', dD.data_ptr(), pl.buf['coef'].data_ptr(), 1, pl.buf['dF'].data_ptr(), B,
             sp.C * sp.R * sp.R, _stream())
        pl.run_backward(ctx.gen)
        return (None,) * 8


def _run_cfg(net: EDMPrecond, x, sigma, labels, cfg_scale: float):
    """forward_with_cfg (models/maskdit.py:559-587): one 2B-row evaluation on [x; x] with labels
    [y; 0], guidance on all in_channels (:580), then the EDM output blend.  Inference only."""
    eng = net.engine()
    sp = net.spec
    B = x.shape[0]
    chw = sp.C * sp.R * sp.R
    st = _stream()
    pl = eng.plan(2 * B, False, False, None, net.eval_precision)
    # plan-side coefficients (c_noise feeds the timestep embedder) are laid out for 2B rows;
    # the input scaling / output blend act on the B real samples with their own table
    sig2 = torch.cat([sigma, sigma])
    call('mdt_precond_coef', sig2.data_ptr(), pl.buf['coef'].data_ptr(), 2 * B, float(net.sigma_data), st)
    coef = torch.empty(8, B, device=x.device, dtype=torch.float32)
    call('mdt_precond_coef', sigma.data_ptr(), coef.data_ptr(), B, float(net.sigma_data), st)
    xin = pl.buf['xin']
    call('mdt_scale_rows', x.data_ptr(), coef.data_ptr(), 2, xin.data_ptr(), B, chw, st)
    xin[B:].copy_(xin[:B])
    lab = pl.buf['labels']
    lab[:B].copy_(labels)
    lab[B:].zero_()
    pl.run_forward()
    Fg = torch.empty_like(x)
    call('mdt_cfg_combine', pl.buf['F'].data_ptr(), cfg_scale, Fg.data_ptr(), B * chw, st)
    D = torch.empty_like(x)
    call('mdt_precond_out', x.data_ptr(), Fg.data_ptr(), coef.data_ptr(), D.data_ptr(), B, chw, st)
    return D


Precond_models = {'edm': EDMPrecond}


def _dit_constructor(model_type):
    """models/maskdit.py:649-715 `DiT_XL_2(**kwargs)` ...: constructor registry of the bare DiT parameter containers."""
    def build(input_size=32, in_channels=4, num_classes=1000, use_decoder=True, mae_loss_coef=0.1, **unused):
        return DiT(make_spec(model_type, input_size, in_channels, num_classes, use_decoder, mae_loss_coef))
    build.__name__ = model_type.replace('-', '_').replace('/', '_')
    return build


DiT_models = {name: _dit_constructor(name) for name in MODEL_CONFIGS}  # models/maskdit.py:718-724
