"""CPU oracle of the KL-autoencoder DECODE path (test infrastructure only -- never imported by the product package).

Functional restatement (torch fp32) of the reference's `autoencoder.py`: FrozenAutoencoderKL.decode (:449-453) =
z / scale_factor -> post_quant_conv (1x1) -> Decoder.forward (:374-410): conv_in, mid (ResnetBlock, AttnBlock,
ResnetBlock), four up levels of three ResnetBlocks (+ nearest-2x Upsample with a 3x3 conv on levels 3, 2, 1),
norm_out + swish + conv_out.  ddconfig = get_model's (:455-468): ch 128, ch_mult (1, 2, 4, 4), num_res_blocks 2,
attn_resolutions [], z_channels 4, out_ch 3.

Pinned by tests/golden/vae_decode.npz: the output of the reference's own Decoder / post_quant_conv modules loaded with
`init_vae_params(seed)` (tests/golden/make_golden.py: gen_vae)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

CH, CH_MULT, NUM_RES_BLOCKS, Z_CH, OUT_CH = 128, (1, 2, 4, 4), 2, 4, 3


def vae_param_shapes():
    """Decode-side state-dict keys and shapes of the reference checkpoint layout."""
    shapes = {'post_quant_conv.weight': (Z_CH, Z_CH, 1, 1), 'post_quant_conv.bias': (Z_CH,)}

    def conv(name, cin, cout, k):
        shapes[f'{name}.weight'] = (cout, cin, k, k)
        shapes[f'{name}.bias'] = (cout,)

    def norm(name, c):
        shapes[f'{name}.weight'] = (c,)
        shapes[f'{name}.bias'] = (c,)

    def res(name, cin, cout):  # autoencoder.py:78-115
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
    for i_level in reversed(range(len(CH_MULT))):
        block_out = CH * CH_MULT[i_level]
        for j in range(NUM_RES_BLOCKS + 1):
            res(f'decoder.up.{i_level}.block.{j}', block_in, block_out)
            block_in = block_out
        if i_level != 0:
            conv(f'decoder.up.{i_level}.upsample.conv', block_in, block_in, 3)
    norm('decoder.norm_out', block_in)
    conv('decoder.conv_out', block_in, OUT_CH, 3)
    return shapes


def init_vae_params(seed: int = 0):
    """Deterministic synthetic weights (the published autoencoder_kl.pth is not available offline): convolutions
    N(0, 1.6 / fan_in) so that activations keep O(1) scale through the 30 layers, biases N(0, 0.05), GroupNorm
    gamma 1 + N(0, 0.1), beta N(0, 0.1) -- nothing is zero or one exactly, every term of the path matters."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shp in vae_param_shapes().items():
        if name.endswith('.weight') and len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            P[name] = torch.randn(shp, generator=g) * (1.6 / fan_in) ** 0.5
        elif '.norm' in name and name.endswith('.weight'):
            P[name] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif '.norm' in name:
            P[name] = 0.1 * torch.randn(shp, generator=g)
        else:
            P[name] = 0.05 * torch.randn(shp, generator=g)
    return P


def _gn(x, P, name):  # Normalize: GroupNorm(32, eps 1e-6, affine)  (autoencoder.py:35-36)
    return F.group_norm(x, 32, P[name + '.weight'], P[name + '.bias'], eps=1e-6)


def _swish(x):  # autoencoder.py:30-32
    return x * torch.sigmoid(x)


def _conv(x, P, name, pad):
    return F.conv2d(x, P[name + '.weight'], P[name + '.bias'], stride=1, padding=pad)


def _res(x, P, name):  # ResnetBlock.forward with temb = None (autoencoder.py:117-137)
    h = _conv(_swish(_gn(x, P, name + '.norm1')), P, name + '.conv1', 1)
    h = _conv(_swish(_gn(h, P, name + '.norm2')), P, name + '.conv2', 1)
    if name + '.nin_shortcut.weight' in P:
        x = _conv(x, P, name + '.nin_shortcut', 0)
    return x + h


def _attn(x, P, name):  # AttnBlock.forward (autoencoder.py:174-199)
    h = _gn(x, P, name + '.norm')
    q, k, v = (_conv(h, P, f'{name}.{n}', 0) for n in 'qkv')
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    h = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(h, P, name + '.proj_out', 0)


def vae_decode(P, z, scale_factor: float = 0.18215):
    """z [B, 4, R, R] -> image [B, 3, 8R, 8R]  (autoencoder.py:449-453, 374-410)."""
    z = F.conv2d(z / scale_factor, P['post_quant_conv.weight'], P['post_quant_conv.bias'])
    h = _conv(z, P, 'decoder.conv_in', 1)
    h = _res(h, P, 'decoder.mid.block_1')
    h = _attn(h, P, 'decoder.mid.attn_1')
    h = _res(h, P, 'decoder.mid.block_2')
    for i_level in reversed(range(len(CH_MULT))):
        for j in range(NUM_RES_BLOCKS + 1):
            h = _res(h, P, f'decoder.up.{i_level}.block.{j}')
        if i_level != 0:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')  # Upsample (autoencoder.py:48-52)
            h = _conv(h, P, f'decoder.up.{i_level}.upsample.conv', 1)
    return _conv(_swish(_gn(h, P, 'decoder.norm_out')), P, 'decoder.conv_out', 1)
