"""Drop-in for the DECODE side of the reference's `autoencoder.py` (FrozenAutoencoderKL / get_model, autoencoder.py:
412-474): the latent -> image step that follows the sampler (sample.py:248,273-284).

    vae = maskdit_amd.autoencoder.get_model('assets/stable_diffusion/autoencoder_kl.pth')   # reference call, sample.py:248
    images = vae.decode(z)          # z [B, 4, 32, 32] (or 64x64) fp32 on the HIP device -> [B, 3, 8R, 8R] fp32

Parameters carry the reference's state-dict names and shapes (`decoder.mid.block_1.conv1.weight` [512, 512, 3, 3], ...),
so the published `autoencoder_kl.pth` loads with `load_state_dict` (its `encoder.*` / `quant_conv.*` entries are
accepted and ignored: encoding is outside the sampling path -- training consumes pre-computed latents).

Arithmetic (maskdit_amd/csrc/vae.hip + the bf16 MFMA GEMMs): activations NHWC fp32; `mdt_gn_im2col` applies GroupNorm(32,
eps 1e-6) + swish and writes the bf16 operand; every 3x3 convolution with a multiple of 128 input channels is ONE implicit
GEMM (`mdt_conv3x3_nhwc`, round 3: the MFMA kernel gathers the nine taps, the zero padding and the nearest 2x up-sampling
from the NHWC activation -- rounds 1-2 materialised an im2col matrix of 9x the activation bytes), the 1x1 convolutions and
conv_in (4 channels) are `mdt_gemm_nt` on the activation / a small im2col matrix; the mid-block attention
(1024 tokens, one head of 512 channels) is three GEMMs per image around `mdt_softmax_rows`.  No torch arithmetic, no
CPU fallback.  Supported latent sides: 16, 32, 48, 64 (decode() raises for others).  Determinism: with the fused
convolution epilogue (default) the next GroupNorm's statistics are accumulated with fp32 atomics across waves, so two decodes of
the same latents can differ in the last bits; MDT_VAE_FUSE=0 selects the separate mdt_gn_stats pass, which is run-to-run
bitwise reproducible (use it where that matters, e.g. FID bookkeeping across runs).  ddconfig is the reference's (ch 128, ch_mult (1, 2, 4, 4), 2 res blocks, no attention resolutions,
z_channels 4, 3 output channels)."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import call

CH, CH_MULT, NUM_RES_BLOCKS, Z_CH, OUT_CH, GROUPS = 128, (1, 2, 4, 4), 2, 4, 3, 32
FUSE_EPILOGUE = os.environ.get('MDT_VAE_FUSE', '1') != '0'  # A/B switch (see _conv)
# widest convolution (output channels) that takes the fused epilogue: it exists for 128-column tiles only (the 256-wide
# kernel spills with it), so for 256 / 512 channels fusing trades a ~15-20 % slower GEMM against the saved passes
FUSE_MAX_COUT = int(os.environ.get('MDT_VAE_FUSE_MAXC', '256'))  # measured at batch 64: 128 -> 62.4 ms, 256 -> 61.7, 512 -> 64.4, off -> 67.4


def decoder_param_table() -> List[Tuple[str, tuple]]:
    """(state-dict key, shape) of post_quant_conv + decoder, in the reference's registration order
    (autoencoder.py:306-372, 419)."""
    t: List[Tuple[str, tuple]] = []

    def conv(name, cin, cout, k):
        t.extend([(f'{name}.weight', (cout, cin, k, k)), (f'{name}.bias', (cout,))])

    def norm(name, c):
        t.extend([(f'{name}.weight', (c,)), (f'{name}.bias', (c,))])

    def res(name, cin, cout):
        norm(f'{name}.norm1', cin)
        conv(f'{name}.conv1', cin, cout, 3)
        norm(f'{name}.norm2', cout)
        conv(f'{name}.conv2', cout, cout, 3)
        if cin != cout:
            conv(f'{name}.nin_shortcut', cin, cout, 1)

    block_in = CH * CH_MULT[-1]
    conv('decoder.conv_in', Z_CH, block_in, 3)
    res('decoder.mid.block_1', block_in, block_in)
    norm('decoder.mid.attn_1.norm', block_in)
    for n in ('q', 'k', 'v', 'proj_out'):
        conv(f'decoder.mid.attn_1.{n}', block_in, block_in, 1)
    res('decoder.mid.block_2', block_in, block_in)
    ups: Dict[int, List[Tuple[str, tuple]]] = {}
    for i_level in reversed(range(len(CH_MULT))):  # built top level first, stored under up.{i_level}
        cur: List[Tuple[str, tuple]] = []
        saved, t = t, cur
        block_out = CH * CH_MULT[i_level]
        for j in range(NUM_RES_BLOCKS + 1):
            res(f'decoder.up.{i_level}.block.{j}', block_in, block_out)
            block_in = block_out
        if i_level != 0:
            conv(f'decoder.up.{i_level}.upsample.conv', block_in, block_in, 3)
        t = saved
        ups[i_level] = cur
    for i_level in range(len(CH_MULT)):  # `self.up.insert(0, up)`: module order is up.0 .. up.3
        t.extend(ups[i_level])
    norm('decoder.norm_out', block_in)
    conv('decoder.conv_out', block_in, OUT_CH, 3)
    return [('post_quant_conv.weight', (Z_CH, Z_CH, 1, 1)), ('post_quant_conv.bias', (Z_CH,))] + t


def _rup(x, m):
    return (x + m - 1) // m * m


class FrozenAutoencoderKL(nn.Module):
    """Decode-only counterpart of autoencoder.py:412-466."""

    def __init__(self, pretrained_path: Optional[str] = None, scale_factor: float = 0.18215):
        super().__init__()
        self.scale_factor = scale_factor
        self.embed_dim = Z_CH
        self._names: List[str] = []
        for name, shp in decoder_param_table():
            p = nn.Parameter(torch.zeros(shp), requires_grad=False)
            self.register_parameter(name.replace('.', '__'), p)  # flat registration, reference names restored below
            self._names.append(name)
        self._packed: Optional[dict] = None
        self._wdict: Optional[dict] = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        if pretrained_path is not None:
            sd = torch.load(pretrained_path, map_location='cpu')
            self.load_state_dict(sd)
        self.eval()

    # ---- state dict under the reference's dotted names ------------------------------------
    def named_weights(self):
        for name in self._names:
            yield name, getattr(self, name.replace('.', '__'))

    def state_dict(self, *a, **k):
        return {name: p.detach() for name, p in self.named_weights()}

    def load_state_dict(self, sd, strict: bool = True):
        """Accepts the full reference checkpoint: `encoder.*` / `quant_conv.*` are not part of the decode path and are
        skipped; every decode-side key must be present (strict) with the reference shape."""
        own = dict(self.named_weights())
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own and not k.startswith(('encoder.', 'quant_conv.'))]
        if strict and (missing or unexpected):
            raise RuntimeError(f'autoencoder state dict: missing {missing[:4]}... unexpected {unexpected[:4]}...')
        with torch.no_grad():
            for k, p in own.items():
                if k in sd:
                    if tuple(sd[k].shape) != tuple(p.shape):
                        raise RuntimeError(f'{k}: shape {tuple(sd[k].shape)} != {tuple(p.shape)}')
                    p.copy_(sd[k])
        self._packed = None
        return missing, unexpected

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._wdict = None
        self._act_zeroed = None
        self._ws.clear()
        return super()._apply(fn, *a, **k)

    # ---- GEMM-side weight images ------------------------------------------------------------
    def _pack(self):
        """conv weight [Cout, Cin, k, k] -> bf16 [Np, Kp] with K ordered (ky, kx, cin) like the im2col rows, N padded
        to 128 and K to 64; biases fp32 [Np].  The attention value bias is folded into the output projection's
        (softmax rows sum to 1: P (h Wv^T + 1 bv^T) = P h Wv^T + 1 bv^T)."""
        W = dict(self.named_weights())
        dev = next(self.parameters()).device
        pk = {}
        self._cout = {}
        for name, p in W.items():
            if name.endswith('.weight') and p.dim() == 4:
                base = name[:-len('.weight')]
                cout, cin, k, _ = p.shape
                Kp, Np = _rup(k * k * cin, 64), _rup(cout, 128)
                m = torch.zeros(Np, Kp, device=dev, dtype=torch.float32)
                m[:cout, :k * k * cin] = p.detach().permute(0, 2, 3, 1).reshape(cout, -1)
                b = torch.zeros(Np, device=dev, dtype=torch.float32)
                b[:cout] = W[base + '.bias'].detach()
                pk[base] = (m.to(torch.bfloat16).contiguous(), b, Kp, Np)
                self._cout[base] = cout
        a = 'decoder.mid.attn_1'
        wp = W[a + '.proj_out.weight'].detach().reshape(W[a + '.proj_out.weight'].shape[0], -1)
        beff = W[a + '.proj_out.bias'].detach() + wp @ W[a + '.v.bias'].detach()
        m, _, Kp, Np = pk[a + '.proj_out']
        pk[a + '.proj_out'] = (m, beff.contiguous(), Kp, Np)
        self._packed = pk
        return pk

    def _weights(self):
        """name -> parameter, built once per device binding (round 2 rebuilt this dict for every convolution)"""
        if self._wdict is None:
            self._wdict = dict(self.named_weights())
        return self._wdict

    def _buf(self, key, shape, dtype):
        t = self._ws.get(key)
        n = 1
        for s in shape:
            n *= s
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, device=next(self.parameters()).device, dtype=dtype)
            self._ws[key] = t
        return t[:n].view(shape)

    # ---- building blocks -----------------------------------------------------------------------
    def _conv(self, x, B, H, cin, name, k=3, norm=None, swish=False, up=0, slot='a', in_stats=None, res=None, want_stats=False):
        """x: fp32 [B*H*H, cin] (NHWC) -> (fp32 [B*Ho*Ho, Np], stats).  `in_stats`: GroupNorm sums of x that the
        PRODUCER of x already accumulated (round 4: the implicit-GEMM convolution's epilogue), else mdt_gn_stats runs;
        `res`: fp32 [B*Ho*Ho, Np] added to the result inside the epilogue where the implicit-GEMM kernel runs (else by
        mdt_add_f32); `want_stats`: return the sums [B, 32, 2] of the result when the epilogue can produce them."""
        st = ops.stream_ptr()
        W = self._weights()
        wmat, bias, Kp, Np = self._packed[name]
        sums = gamma = beta = None
        if norm is not None:
            sums = in_stats
            if sums is None:
                sums = self._buf('sums', (B, GROUPS, 2), torch.float32)
                call('mdt_gn_stats', x.data_ptr(), sums.data_ptr(), B, H * H, cin, GROUPS, st)
            gamma, beta = W[norm + '.weight'], W[norm + '.bias']
        Ho = H << up
        M = B * Ho * Ho
        # the implicit-GEMM kernel's shape domain (mdt_conv3x3_nhwc): power-of-two image sides >= 8, whole 256-row tiles,
        # 8-bit batch index, 32-bit source offsets; anything else inside decode()'s domain (R = 48 latents: 48, 96, 192, 384
        # pixel sides; a batch whose pixel rows are not whole 256-row tiles) takes the materialised-im2col GEMM below, which has
        # no shape restriction of its own.  decode() itself accepts R = 16, 32, 48, 64 only (the mid-block attention's T x T
        # GEMMs need R * R % 128 == 0): R = 8 / 24 / 40 raise NotImplementedError there -- the reference decoder is
        # size-agnostic, this one covers the sides the shipped configs (32, 64) and their neighbours use
        implicit_ok = (k == 3 and cin % 128 == 0 and H >= 8 and (H & (H - 1)) == 0 and M % 256 == 0 and B < 256
                       and Ho <= 2048 and B * H * H * cin * 2 + 256 < (1 << 32))
        if implicit_ok:
            # implicit GEMM (round 3): the normalised activation is written ONCE as bf16 NHWC (ksize-1 form of
            # mdt_gn_im2col) behind a 256-byte zero line, the MFMA kernel gathers the nine taps (and the 2x up-sampling)
            # itself -- no im2col matrix (9x the activation bytes per convolution in rounds 1-2)
            raw = self._buf('act', (128 + B * H * H * cin,), torch.bfloat16)
            if getattr(self, '_act_zeroed', None) != raw.data_ptr():
                raw[:128].zero_()
                self._act_zeroed = raw.data_ptr()
            act = raw[128:]
            call('mdt_gn_im2col', x.data_ptr(), sums.data_ptr() if sums is not None else None,
                 gamma.data_ptr() if gamma is not None else None, beta.data_ptr() if beta is not None else None, act.data_ptr(),
                 B, H, H, cin, GROUPS, 1, 0, int(swish), cin, st)
            out = self._buf('out_' + slot, (M, Np), torch.float32)
            # round 4: the skip connection and the NEXT GroupNorm's statistics come out of the epilogue (128-column tiles;
            # MDT_VAE_FUSE=0 = the round-3 form: 256-column tiles where they divide, separate add / statistics passes)
            cout = self._cout[name]
            cpg = cout // GROUPS if cout % GROUPS == 0 else 0
            stats = None
            fuse = FUSE_EPILOGUE and cout <= FUSE_MAX_COUT
            if fuse and want_stats and cout == Np and cpg >= 4 and (cpg & (cpg - 1)) == 0 and (Ho * Ho) % 128 == 0:
                stats = self._buf('sums_' + slot, (B, GROUPS, 2), torch.float32)
                stats.zero_()
            fres = res if fuse else None
            call('mdt_conv3x3_nhwc', act.data_ptr(), B, H, cin, up, wmat.data_ptr(), bias.data_ptr(),
                 fres.data_ptr() if fres is not None else None, out.data_ptr(), Np, Np,
                 stats.data_ptr() if stats is not None else None, GROUPS, st)
            if res is not None and fres is None:
                out = self._add(res, out, slot)
            return out, stats
        col = self._buf('col', (M, Kp), torch.bfloat16)
        call('mdt_gn_im2col', x.data_ptr(), sums.data_ptr() if sums is not None else None,
             gamma.data_ptr() if gamma is not None else None, beta.data_ptr() if beta is not None else None, col.data_ptr(),
             B, H, H, cin, GROUPS, k, up, int(swish), Kp, st)
        out = self._buf('out_' + slot, (M, Np), torch.float32)
        ops.gemm_nt(col, wmat, bias, ops.EPI_F32, outf=out)
        if res is not None:
            out = self._add(res, out, slot)
        return out, None

    def _add(self, a, b, slot):
        c = self._buf('sum_' + slot, tuple(a.shape), torch.float32)
        call('mdt_add_f32', a.data_ptr(), b.data_ptr(), c.data_ptr(), a.numel(), ops.stream_ptr())
        return c

    def _res(self, x, B, H, cin, cout, name, slot, x_stats=None):
        """ResnetBlock (autoencoder.py:78-140): x + conv2(swish(norm2(conv1(swish(norm1(x)))))) (1x1 shortcut where the widths
        differ) -> (out, GroupNorm sums of out or None).  The add and both statistics ride on the convolutions' epilogues."""
        h, h_stats = self._conv(x, B, H, cin, name + '.conv1', norm=name + '.norm1', swish=True, slot='h1', in_stats=x_stats,
                                want_stats=True)
        if cin != cout:
            x, _ = self._conv(x, B, H, cin, name + '.nin_shortcut', k=1, slot='sc')
        return self._conv(h, B, H, cout, name + '.conv2', norm=name + '.norm2', swish=True, slot=slot, in_stats=h_stats, res=x,
                          want_stats=True)

    def _attn(self, x, B, H, c, name, slot):
        st = ops.stream_ptr()
        W = self._weights()
        T = H * H
        sums = self._buf('sums', (B, GROUPS, 2), torch.float32)
        call('mdt_gn_stats', x.data_ptr(), sums.data_ptr(), B, T, c, GROUPS, st)
        hn = self._buf('attn_hn', (B * T, c), torch.bfloat16)
        call('mdt_gn_im2col', x.data_ptr(), sums.data_ptr(), W[name + '.norm.weight'].data_ptr(), W[name + '.norm.bias'].data_ptr(),
             hn.data_ptr(), B, H, H, c, GROUPS, 1, 0, 0, c, st)
        q = self._buf('attn_q', (B * T, c), torch.bfloat16)
        k = self._buf('attn_k', (B * T, c), torch.bfloat16)
        ops.gemm_nt(hn, self._packed[name + '.q'][0], self._packed[name + '.q'][1], ops.EPI_BF16, out=q)
        ops.gemm_nt(hn, self._packed[name + '.k'][0], self._packed[name + '.k'][1], ops.EPI_BF16, out=k)
        wv = self._packed[name + '.v'][0]
        o = self._buf('attn_o', (B * T, c), torch.bfloat16)
        vT = self._buf('attn_vT', (c, T), torch.bfloat16)
        S = self._buf('attn_S', (T, T), torch.float32)
        P = self._buf('attn_P', (T, T), torch.bfloat16)
        for b in range(B):
            rows = slice(b * T, (b + 1) * T)
            ops.gemm_nt(wv, hn[rows], None, ops.EPI_BF16, out=vT)               # v^T = Wv h^T  (bias folded into proj_out)
            ops.gemm_nt(q[rows], k[rows], None, ops.EPI_F32, outf=S)             # w_[i, j] = q_i . k_j   (autoencoder.py:186)
            call('mdt_softmax_rows', S.data_ptr(), P.data_ptr(), T, T, float(c) ** -0.5, st)  # :187-188
            ops.gemm_nt(P, vT, None, ops.EPI_BF16, out=o[rows])                  # h_[i, :] = sum_j P[i, j] v_j   (:191-193)
        wp, bp, _, _ = self._packed[name + '.proj_out']
        proj = self._buf('out_a', (B * T, c), torch.float32)
        ops.gemm_nt(o, wp, bp, ops.EPI_F32, outf=proj)
        return self._add(x, proj, slot)

    # ---- public surface -----------------------------------------------------------------------
    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:449-453: z / scale_factor -> post_quant_conv -> Decoder -> image [B, 3, 8R, 8R] fp32."""
        if not z.is_cuda:
            raise _lib.MaskDiTLibError('maskdit_amd.autoencoder: z is not on a HIP device; there is no CPU path')
        if next(self.parameters()).device != z.device:
            raise _lib.MaskDiTLibError('maskdit_amd.autoencoder: call .to(z.device) first')
        if self._packed is None:
            self._pack()
        z = z.to(torch.float32).contiguous()
        B, C, R, R2 = z.shape
        assert C == Z_CH and R == R2 and R % 8 == 0, f'latent shape {tuple(z.shape)}'
        if (R * R) % 128 or R * R > 4096:
            # the mid-block attention runs its T x T score GEMMs through mdt_gemm_nt (N % 128) and mdt_softmax_rows
            # (<= 4096 keys): R = 16, 32, 48, 64 (128 .. 512 px images; the shipped configs use 32 and 64)
            raise NotImplementedError(f'maskdit_amd.autoencoder: latent side {R} unsupported (R * R must be a multiple of 128, <= 4096)')
        st = ops.stream_ptr()
        W = self._weights()
        x = self._buf('x0', (B * R * R, Z_CH), torch.float32)
        call('mdt_vae_prologue', z.data_ptr(), W['post_quant_conv.weight'].data_ptr(), W['post_quant_conv.bias'].data_ptr(),
             x.data_ptr(), B, R * R, float(self.scale_factor), st)
        c = CH * CH_MULT[-1]
        H = R
        x, xs = self._conv(x, B, H, Z_CH, 'decoder.conv_in', slot='x1')
        x, xs = self._res(x, B, H, c, c, 'decoder.mid.block_1', 'p', xs)
        x = self._attn(x, B, H, c, 'decoder.mid.attn_1', 'q')
        x, xs = self._res(x, B, H, c, c, 'decoder.mid.block_2', 'p')
        flip = 1  # block_2 left x in slot 'p': the first block of the ladder writes slot 'q'
        for i_level in reversed(range(len(CH_MULT))):
            cout = CH * CH_MULT[i_level]
            for j in range(NUM_RES_BLOCKS + 1):
                x, xs = self._res(x, B, H, c, cout, f'decoder.up.{i_level}.block.{j}', 'pq'[flip], xs)
                flip ^= 1
                c = cout
            if i_level != 0:
                x, xs = self._conv(x, B, H, c, f'decoder.up.{i_level}.upsample.conv', up=1, slot='u', want_stats=True)
                H *= 2
        y, _ = self._conv(x, B, H, c, 'decoder.conv_out', norm='decoder.norm_out', swish=True, slot='a', in_stats=xs)
        img = torch.empty(B, OUT_CH, H, H, device=z.device, dtype=torch.float32)
        call('mdt_vae_epilogue', y.data_ptr(), y.shape[1], img.data_ptr(), B, H * H, OUT_CH, st)
        return img

    def encode(self, x):
        raise NotImplementedError('encoding (autoencoder.py:203-304) is outside the sampling path: training consumes '
                                  'pre-computed latent moments (train_utils/datasets.py:240-304)')

    def forward(self, inputs, fn):
        if fn == 'decode':
            return self.decode(inputs)
        return self.encode(inputs)

    def release_workspace(self):
        self._ws.clear()
        self._act_zeroed = None


def get_model(pretrained_path: Optional[str], scale_factor: float = 0.18215) -> FrozenAutoencoderKL:
    """autoencoder.py:468-474 (`pretrained_path=None`: zero weights, to be filled with load_state_dict)."""
    return FrozenAutoencoderKL(pretrained_path, scale_factor)
