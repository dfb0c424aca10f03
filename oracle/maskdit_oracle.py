"""CPU oracle for the MaskDiT hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional restatement (plain torch-CPU fp32 / numpy)
of the arithmetic on the reference's hot path.  Nothing in the product package
(`maskdit_amd/`) may import it: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` do, and only as the checker / the timed CPU
baseline.  The product path fails loudly when the HIP library is missing.

Pinning: the reference ships no tests / golden vectors, so
the oracle is pinned against *outputs of the reference itself* run in the build
container: `tests/golden/make_golden.py` imports `/root/reference` (with a timm
shim), writes `tests/golden/*.npz`, and `tests/test_oracle_golden.py` checks this
restatement against those fixtures.  Third-party arithmetic not under
/root/reference: `timm` (unpinned, reference Dockerfile:3; PatchEmbed / Attention /
Mlp, call sites models/maskdit.py:16,178,182,278) and `apex.FusedAdam` (unpinned;
train.py:141) -- restated here from their published semantics (timm >= 0.9
vision_transformer; Adam with decoupled weight decay, bias-corrected).

Every function cites the reference file:line it follows (paths relative to the
reference repo root).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# model configurations (models/maskdit.py:649-715); decoder constants :310-312

MODEL_CONFIGS = {
    # name: (depth, hidden, patch, heads)
    'DiT-H/2': (32, 1280, 2, 16), 'DiT-H/4': (32, 1280, 4, 16), 'DiT-H/8': (32, 1280, 8, 16),
    'DiT-XL/2': (28, 1152, 2, 16), 'DiT-XL/4': (28, 1152, 4, 16), 'DiT-XL/8': (28, 1152, 8, 16),
    'DiT-L/2': (24, 1024, 2, 16), 'DiT-L/4': (24, 1024, 4, 16), 'DiT-L/8': (24, 1024, 8, 16),
    'DiT-B/2': (12, 768, 2, 12), 'DiT-B/4': (12, 768, 4, 12), 'DiT-B/8': (12, 768, 8, 12),
    'DiT-S/2': (12, 384, 2, 6), 'DiT-S/4': (12, 384, 4, 6), 'DiT-S/8': (12, 384, 8, 6),
}
DEC_HIDDEN, DEC_DEPTH, DEC_HEADS = 512, 8, 16  # models/maskdit.py:310-312


def make_cfg(model_type='DiT-S/2', img_resolution=32, img_channels=4, num_classes=1000,
             use_decoder=True, mae_loss_coef=0.1, depth=None, hidden=None, heads=None, patch=None):
    d, h, p, nh = MODEL_CONFIGS.get(model_type, (depth, hidden, patch, heads))
    return dict(depth=depth or d, D=hidden or h, patch=patch or p, heads=heads or nh,
                R=img_resolution, C=img_channels, num_classes=num_classes,
                use_decoder=use_decoder, mae_loss_coef=mae_loss_coef,
                Dd=DEC_HIDDEN, ddepth=DEC_DEPTH, dheads=DEC_HEADS, mlp_ratio=4.0)


# ----------------------------------------------------------------------------
# positional embedding (models/maskdit.py:595-642)

def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """models/maskdit.py:624-642: [sin(pos*w), cos(pos*w)], w_i = 10000^(-i/(dim/2))."""
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, grid_size: int) -> np.ndarray:
    """models/maskdit.py:595-621.  NOTE meshgrid(grid_w, grid_h): the first half of
    the channels encodes the *w* coordinate (:603, :617)."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    e0 = sincos_1d(embed_dim // 2, grid[0])
    e1 = sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([e0, e1], axis=1)  # [T, D] float64


# ----------------------------------------------------------------------------
# parameter construction (names / shapes of the reference state dict)

def param_shapes(cfg) -> Dict[str, tuple]:
    """State-dict keys and shapes of EDMPrecond (models/maskdit.py:242-332, 722-741) for the
    shipped flag set (pad_cls_token False, ext_feature_dim 0, no learn_sigma)."""
    D, C, p, R = cfg['D'], cfg['C'], cfg['patch'], cfg['R']
    T = (R // p) ** 2
    Dd = cfg['Dd']
    hid = int(D * cfg['mlp_ratio'])
    s = {}
    s['model.pos_embed'] = (1, T, D)
    s['model.x_embedder.proj.weight'] = (D, C, p, p)
    s['model.x_embedder.proj.bias'] = (D,)
    s['model.t_embedder.mlp.0.weight'] = (D, 256)
    s['model.t_embedder.mlp.0.bias'] = (D,)
    s['model.t_embedder.mlp.2.weight'] = (D, D)
    s['model.t_embedder.mlp.2.bias'] = (D,)
    s['model.y_embedder.embedding_table.weight'] = (D, cfg['num_classes'])

    def block(prefix, W, cdim):
        h = int(W * cfg['mlp_ratio'])
        s[f'{prefix}.attn.qkv.weight'] = (3 * W, W)
        s[f'{prefix}.attn.qkv.bias'] = (3 * W,)
        s[f'{prefix}.attn.proj.weight'] = (W, W)
        s[f'{prefix}.attn.proj.bias'] = (W,)
        s[f'{prefix}.mlp.fc1.weight'] = (h, W)
        s[f'{prefix}.mlp.fc1.bias'] = (h,)
        s[f'{prefix}.mlp.fc2.weight'] = (W, h)
        s[f'{prefix}.mlp.fc2.bias'] = (W,)
        s[f'{prefix}.adaLN_modulation.1.weight'] = (6 * W, cdim)
        s[f'{prefix}.adaLN_modulation.1.bias'] = (6 * W,)

    for i in range(cfg['depth']):
        block(f'model.blocks.{i}', D, D)
    fin = D
    if cfg['use_decoder']:
        s['model.decoder_pos_embed'] = (1, T, Dd)
        s['model.decoder_layer.linear.weight'] = (Dd, D)
        s['model.decoder_layer.linear.bias'] = (Dd,)
        s['model.decoder_layer.adaLN_modulation.1.weight'] = (2 * D, D)
        s['model.decoder_layer.adaLN_modulation.1.bias'] = (2 * D,)
        for i in range(cfg['ddepth']):
            block(f'model.decoder_blocks.{i}', Dd, D)
        if cfg['mae_loss_coef'] > 0:
            s['model.mask_token'] = (1, 1, Dd)
        fin = Dd
    s['model.final_layer.linear.weight'] = (p * p * C, fin)
    s['model.final_layer.linear.bias'] = (p * p * C,)
    s['model.final_layer.adaLN_modulation.1.weight'] = (2 * fin, D)
    s['model.final_layer.adaLN_modulation.1.bias'] = (2 * fin,)
    return s


NON_TRAINABLE = ('model.pos_embed', 'model.decoder_pos_embed')  # models/maskdit.py:296,315-317


def init_params(cfg, seed=0, dezero=True) -> Dict[str, torch.Tensor]:
    """Fresh parameters with the reference's init *distributions* (models/maskdit.py:334-409);
    RNG stream parity with the reference is not a goal.  `dezero=True` redraws every
    tensor the reference zero-initialises from N(0, 0.02) so that parity tests are not
    vacuous."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shp in param_shapes(cfg).items():
        if name == 'model.pos_embed':
            P[name] = torch.from_numpy(sincos_2d(cfg['D'], cfg['R'] // cfg['patch'])).float().unsqueeze(0)
        elif name == 'model.decoder_pos_embed':
            P[name] = torch.from_numpy(sincos_2d(cfg['Dd'], cfg['R'] // cfg['patch'])).float().unsqueeze(0)
        elif name.endswith('.bias'):
            P[name] = torch.zeros(shp)
        elif name in ('model.y_embedder.embedding_table.weight', 'model.t_embedder.mlp.0.weight',
                      'model.t_embedder.mlp.2.weight', 'model.mask_token'):
            P[name] = torch.randn(shp, generator=g) * 0.02
        elif 'adaLN_modulation' in name or name.startswith('model.final_layer.linear') \
                or name.startswith('model.decoder_layer.linear'):
            P[name] = torch.zeros(shp)
        else:  # xavier uniform on the [out, prod(rest)] view
            fan_out, fan_in = shp[0], int(np.prod(shp[1:]))
            a = math.sqrt(6.0 / (fan_in + fan_out))
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * a
    if dezero:
        dezero_(P, seed + 1)
    return P


def dezero_(P: Dict[str, torch.Tensor], seed=1):
    g = torch.Generator().manual_seed(seed)
    for name, t in P.items():
        if name not in NON_TRAINABLE and float(t.abs().max()) == 0.0:
            t.copy_(torch.randn(t.shape, generator=g) * 0.02)
    return P


# ----------------------------------------------------------------------------
# masking (models/maskdit.py:88-127, 157-163)

def get_mask_from_noise(noise: np.ndarray, mask_ratio: float):
    """models/maskdit.py:88-113 given the noise tensor.  Tie rule: *stable* argsort
    (lower index first); the reference's torch.argsort is unspecified on ties, so parity
    is asserted on tie-free rows plus the permutation invariants."""
    B, T = noise.shape
    len_keep = int(T * (1 - mask_ratio))
    ids_shuffle = np.argsort(noise, axis=1, kind='stable')
    ids_restore = np.argsort(ids_shuffle, axis=1, kind='stable')
    ids_keep = ids_shuffle[:, :len_keep]
    mask = np.ones((B, T), dtype=np.float32)
    mask[:, :len_keep] = 0
    mask = np.take_along_axis(mask, ids_restore, axis=1)
    return dict(mask=mask, ids_keep=ids_keep.astype(np.int64), ids_restore=ids_restore.astype(np.int64),
                ids_shuffle=ids_shuffle.astype(np.int64))


# ----------------------------------------------------------------------------
# model forward (functional)

def _modulate(x, shift, scale):
    """models/maskdit.py:19-20."""
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def _ln(x):
    """nn.LayerNorm(elementwise_affine=False, eps=1e-6), models/maskdit.py:177."""
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def _attention(x, Wqkv, bqkv, Wp, bp, heads):
    """timm Attention(dim, num_heads, qkv_bias=True) (call site models/maskdit.py:178):
    qkv Linear -> [B,N,3,H,hd] -> softmax(q k^T / sqrt(hd)) v -> proj Linear."""
    B, N, Cw = x.shape
    hd = Cw // heads
    qkv = F.linear(x, Wqkv, bqkv).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q * hd ** -0.5) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, Cw)
    return F.linear(o, Wp, bp)


def _block(P, prefix, x, c, heads):
    """DiTBlock.forward, models/maskdit.py:188-192 (chunk order :189)."""
    mod = F.linear(F.silu(c), P[f'{prefix}.adaLN_modulation.1.weight'], P[f'{prefix}.adaLN_modulation.1.bias'])
    sh1, sc1, g1, sh2, sc2, g2 = mod.chunk(6, dim=1)
    a = _attention(_modulate(_ln(x), sh1, sc1), P[f'{prefix}.attn.qkv.weight'], P[f'{prefix}.attn.qkv.bias'],
                   P[f'{prefix}.attn.proj.weight'], P[f'{prefix}.attn.proj.bias'], heads)
    x = x + g1.unsqueeze(1) * a
    h = F.linear(_modulate(_ln(x), sh2, sc2), P[f'{prefix}.mlp.fc1.weight'], P[f'{prefix}.mlp.fc1.bias'])
    h = F.gelu(h, approximate='tanh')
    h = F.linear(h, P[f'{prefix}.mlp.fc2.weight'], P[f'{prefix}.mlp.fc2.bias'])
    return x + g2.unsqueeze(1) * h


def timestep_embedding(t, dim=256, max_period=10000):
    """models/maskdit.py:41-60: [cos, sin] (cos first)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def dit_forward(P, cfg, x, t, y, mask_dict=None, training=True):
    """DiT.forward for the shipped flag set, models/maskdit.py:467-557.
    x [N,C,R,R]; t [N] (or [1]); y [N,num_classes] float; mask_dict with torch int64
    'ids_keep','ids_restore' (masking applied only when `training`, :482,:539)."""
    D, C, p, R = cfg['D'], cfg['C'], cfg['patch'], cfg['R']
    N = x.shape[0]
    # x_embedder: Conv2d(k=s=p) == per-patch linear on (c,p,q)-ordered patch vectors (:278,:475)
    tok = F.conv2d(x, P['model.x_embedder.proj.weight'], P['model.x_embedder.proj.bias'], stride=p)
    tok = tok.flatten(2).transpose(1, 2) + P['model.pos_embed']
    if mask_dict is not None and training:
        ids_keep = mask_dict['ids_keep']
        tok = torch.gather(tok, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))  # mask_out_token :116-127
    temb = timestep_embedding(t)
    temb = F.linear(temb, P['model.t_embedder.mlp.0.weight'], P['model.t_embedder.mlp.0.bias'])
    temb = F.linear(F.silu(temb), P['model.t_embedder.mlp.2.weight'], P['model.t_embedder.mlp.2.bias'])
    c = temb + F.linear(y, P['model.y_embedder.embedding_table.weight'])  # :491-495
    for i in range(cfg['depth']):
        tok = _block(P, f'model.blocks.{i}', tok, c, cfg['heads'])
    if cfg['use_decoder']:
        mod = F.linear(F.silu(c), P['model.decoder_layer.adaLN_modulation.1.weight'],
                       P['model.decoder_layer.adaLN_modulation.1.bias'])
        sh, sc = mod.chunk(2, dim=1)
        tok = F.linear(_modulate(_ln(tok), sh, sc), P['model.decoder_layer.linear.weight'],
                       P['model.decoder_layer.linear.bias'])  # DecoderLayer :209-213
        if mask_dict is not None and training:
            ids_restore = mask_dict['ids_restore']
            T = ids_restore.shape[1]
            mt = P.get('model.mask_token')
            if mt is None:
                mt = torch.zeros(1, 1, tok.shape[2])
            full = torch.cat([tok, mt.expand(N, T - tok.shape[1], -1)], dim=1)  # unmask_tokens :157-163
            tok = torch.gather(full, 1, ids_restore.unsqueeze(-1).expand(-1, -1, tok.shape[2]))
        tok = tok + P['model.decoder_pos_embed']
        for i in range(cfg['ddepth']):
            tok = _block(P, f'model.decoder_blocks.{i}', tok, c, cfg['dheads'])
    mod = F.linear(F.silu(c), P['model.final_layer.adaLN_modulation.1.weight'],
                   P['model.final_layer.adaLN_modulation.1.bias'])
    sh, sc = mod.chunk(2, dim=1)
    tok = F.linear(_modulate(_ln(tok), sh, sc), P['model.final_layer.linear.weight'],
                   P['model.final_layer.linear.bias'])  # FinalLayer :230-234
    if not cfg['use_decoder'] and mask_dict is not None and training:
        ids_restore = mask_dict['ids_restore']
        T = ids_restore.shape[1]
        full = torch.cat([tok, torch.zeros(N, T - tok.shape[1], tok.shape[2])], dim=1)
        tok = torch.gather(full, 1, ids_restore.unsqueeze(-1).expand(-1, -1, tok.shape[2]))
    return unpatchify(tok, p, C)


def unpatchify(tok, p, C):
    """models/maskdit.py:411-424: 'nhwpqc->nchpwq'."""
    N, T, _ = tok.shape
    h = w = int(T ** 0.5)
    x = tok.reshape(N, h, w, p, p, C)
    return torch.einsum('nhwpqc->nchpwq', x).reshape(N, C, h * p, w * p)


def patchify(imgs, p, C):
    """train_utils/loss.py:73-85: 'nchpwq->nhwpqc'."""
    N = imgs.shape[0]
    h = w = imgs.shape[2] // p
    x = imgs.reshape(N, C, h, p, w, p)
    return torch.einsum('nchpwq->nhwpqc', x).reshape(N, h * w, p * p * C)


def dit_forward_with_cfg(P, cfg, x, t, y, cfg_scale):
    """models/maskdit.py:559-587 (guidance on all in_channels, :580)."""
    xx = torch.cat([x, x], 0)
    yy = torch.cat([y, torch.zeros_like(y)], 0)
    out = dit_forward(P, cfg, xx, t, yy, mask_dict=None, training=False)
    cond, uncond = torch.split(out, len(out) // 2, dim=0)
    return uncond + cfg_scale * (cond - uncond)


SIGMA_DATA = 0.5


def precond_forward(P, cfg, x, sigma, class_labels, cfg_scale=None, mask_dict=None, training=True):
    """EDMPrecond.forward, models/maskdit.py:756-773."""
    sigma = sigma.to(x.dtype).reshape(-1, 1, 1, 1)
    sd = SIGMA_DATA
    c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
    c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
    c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
    c_noise = sigma.log() / 4
    if cfg_scale is None:
        Fx = dit_forward(P, cfg, c_in * x, c_noise.flatten(), class_labels, mask_dict=mask_dict, training=training)
    else:
        Fx = dit_forward_with_cfg(P, cfg, c_in * x, c_noise.flatten(), class_labels, cfg_scale)
    return c_skip * x + c_out * Fx


# ----------------------------------------------------------------------------
# loss (train_utils/loss.py:22-101)

def edm_loss(P, cfg, images, labels, rnd_normal, noise, mask_dict=None, mae_loss_coef=0.0,
             P_mean=-1.2, P_std=1.2):
    """EDMLoss.__call__, train_utils/loss.py:28-60, with the random draws passed in
    (rnd_normal [N,1,1,1] is drawn first, then noise = randn_like(images), :35,:39)."""
    sd = SIGMA_DATA
    sigma = (rnd_normal * P_std + P_mean).exp()
    weight = (sigma ** 2 + sd ** 2) / (sigma * sd) ** 2
    y = images
    n = noise * sigma
    D_yn = precond_forward(P, cfg, y + n, sigma, labels, mask_dict=mask_dict, training=True)
    loss = weight * ((D_yn - y) ** 2)
    if mask_dict is not None:
        p = cfg['patch']
        loss = F.avg_pool2d(loss.mean(dim=1), p).flatten(1)  # :47
        unmask = 1 - mask_dict['mask']
        loss = (loss * unmask).sum(dim=1) / unmask.sum(dim=1)  # :48-49
        if mae_loss_coef > 0:
            loss = loss + mae_loss_coef * mae_loss(cfg, y + n, D_yn, 1 - unmask)  # :51-52
    else:
        loss = loss.mean(dim=[1, 2, 3])
    return loss, D_yn


def mae_loss(cfg, target, pred, mask):
    """train_utils/loss.py:88-101 (unbiased var, eps 1e-6)."""
    p, C = cfg['patch'], cfg['C']
    target = patchify(target, p, C)
    pred = patchify(pred, p, C)
    mean = target.mean(dim=-1, keepdim=True)
    var = target.var(dim=-1, keepdim=True)
    target = (target - mean) / (var + 1.e-6) ** .5
    loss = ((pred - target) ** 2).mean(dim=-1)
    return (loss * mask).sum(dim=1) / mask.sum(dim=1)


def loss_and_grads(P, cfg, images, labels, rnd_normal, noise, mask_dict, mae_loss_coef):
    """Forward + autograd backward of loss.mean() (train.py:216-220).  Returns
    (loss[N], D_yn, grads dict for the trainable tensors)."""
    names = [k for k in P if k not in NON_TRAINABLE]
    leaves = {k: P[k].detach().clone().requires_grad_(True) for k in names}
    Q = dict(P)
    Q.update(leaves)
    loss, D = edm_loss(Q, cfg, images, labels, rnd_normal, noise, mask_dict, mae_loss_coef)
    loss.mean().backward()
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])) for k in names}
    return loss.detach(), D.detach(), grads


# ----------------------------------------------------------------------------
# optimizer + EMA (train.py:141,223-230; train_utils/helper.py:47-58)

def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """apex FusedAdam(adam_w_mode=True) == Adam with decoupled weight decay, bias
    corrected (reference shows the torch.optim.AdamW equivalent at train_wds.py:202).
    In-place on p, m, v; `step` is the 1-based step count."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def ema_update(ema, p, decay=0.9999):
    """train_utils/helper.py:47-58."""
    ema.mul_(decay).add_(p, alpha=1 - decay)


def sample_moments(moments, randn, scale_factor=0.18215):
    """utils.py:59-65 with the randn_like(mean) draw passed in."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return scale_factor * (mean + torch.exp(0.5 * logvar) * randn)


# ----------------------------------------------------------------------------
# sampler (sample.py:30-66)

def edm_t_steps(num_steps, sigma_min=0.002, sigma_max=80.0, rho=7):
    """sample.py:40-43 (float64; t_N = 0 appended)."""
    i = torch.arange(num_steps, dtype=torch.float64)
    t = (sigma_max ** (1 / rho) + i / (num_steps - 1) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho
    return torch.cat([t, torch.zeros(1, dtype=torch.float64)])


def edm_sampler(P, cfg, latents, class_labels, cfg_scale=None, num_steps=18, sigma_min=0.002,
                sigma_max=80.0, rho=7, S_churn=0, S_min=0, S_max=float('inf'), S_noise=1, randn_like=torch.randn_like):
    """sample.py:30-66; fp64 state, fp32 net.  S_churn = 0 (every shipped config): gamma = 0, x_hat = x_cur; the
    stochastic branch (sample.py:51-53) draws one `randn_like` per step like the reference does."""
    t_steps = edm_t_steps(num_steps, sigma_min, sigma_max, rho)
    x_next = latents.to(torch.float64) * t_steps[0]
    with torch.no_grad():
        for i in range(num_steps):
            t_cur, t_next = t_steps[i], t_steps[i + 1]
            gamma = min(S_churn / num_steps, 2 ** 0.5 - 1) if S_min <= float(t_cur) <= S_max else 0
            t_hat = t_cur + gamma * t_cur  # round_sigma is the identity for EDMPrecond (models/maskdit.py:775)
            eps = randn_like(x_next)       # drawn even when its coefficient is 0 (generator state parity)
            x_hat = x_next + (t_hat ** 2 - t_cur ** 2).sqrt() * S_noise * eps
            den = precond_forward(P, cfg, x_hat.float(), t_hat, class_labels, cfg_scale=cfg_scale,
                                  training=False).to(torch.float64)
            d_cur = (x_hat - den) / t_hat
            x_next = x_hat + (t_next - t_hat) * d_cur
            if i < num_steps - 1:
                den = precond_forward(P, cfg, x_next.float(), t_next, class_labels, cfg_scale=cfg_scale,
                                      training=False).to(torch.float64)
                d_prime = (x_next - den) / t_next
                x_next = x_hat + (t_next - t_hat) * (0.5 * d_cur + 0.5 * d_prime)
    return x_next


# ----------------------------------------------------------------------------
# one full training step on CPU (used by tests and by bench.py's cpu_baseline leg)

def train_step(P, M, V, EMA, cfg, images, labels, rnd_normal, noise, mask_noise, mask_ratio, mae_loss_coef,
               step, lr=1e-4, ema_decay=0.9999):
    md = get_mask_from_noise(mask_noise.numpy(), mask_ratio)
    mask_dict = {k: torch.from_numpy(v) for k, v in md.items()}
    loss, D, grads = loss_and_grads(P, cfg, images, labels, rnd_normal, noise, mask_dict, mae_loss_coef)
    for k, g in grads.items():
        adamw_step(P[k], g, M[k], V[k], step, lr)
        ema_update(EMA[k], P[k], ema_decay)
    return loss, grads
