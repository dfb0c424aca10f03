"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which does not exist on the GPU
box).  It imports the reference's own `models.maskdit`, `train_utils.loss`,
`train_utils.helper`, `sample.edm_sampler`, `utils.StackedRandomGenerator` (with the
`_refshim` stand-ins for the uninstalled `timm` / `lmdb`), feeds them parameters drawn by
`oracle.maskdit_oracle.init_params` (so a fixture only has to store a seed, not weights)
and records inputs, random draws and outputs.  The reference ships no golden vectors of
its own: these files are the pin for the oracle.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixture tensors are small; parameter-sized results (gradients, updated weights) are stored
as per-tensor checksums (sum, |sum|, L2) plus 64 sampled entries per tensor.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('MASKDIT_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, '_refshim'))
sys.path.insert(0, REF)

from oracle import maskdit_oracle as O  # noqa: E402

import models.maskdit as ref_m  # noqa: E402  (reference)
from train_utils.loss import Losses  # noqa: E402  (reference)
from train_utils.helper import update_ema  # noqa: E402  (reference)
from sample import edm_sampler as ref_edm_sampler  # noqa: E402  (reference)
from utils import StackedRandomGenerator, sample as ref_sample  # noqa: E402  (reference)

torch.set_num_threads(8)


class Wrap(torch.nn.Module):
    """Stands in for DistributedDataParallel: train_utils/loss.py:47,52 dereference
    `net.module`, :56-58 `unwrap_model(net).model`."""

    def __init__(self, m):
        super().__init__()
        self.module = m

    @property
    def model(self):
        return self.module.model

    def forward(self, *a, **k):
        return self.module(*a, **k)


def build_ref(model_type, R, P):
    net = ref_m.Precond_models['edm'](img_resolution=R, img_channels=4, num_classes=1000,
                                      model_type=model_type, use_decoder=True, mae_loss_coef=0.1,
                                      pad_cls_token=False)
    missing = net.load_state_dict(P, strict=True)
    return net


from tests.golden.make_golden_idx import sample_idx  # noqa: E402


def checks(t):
    t64 = t.detach().double().flatten()
    idx = sample_idx(t64.numel())
    return np.array([t64.sum().item(), t64.abs().sum().item(), t64.norm().item()]), t64[idx].numpy()


def one_hot(idx, n=1000):
    y = torch.zeros(len(idx), n)
    y[torch.arange(len(idx)), idx] = 1
    return y


def gen_mask():
    out = {}
    for tag, (B, T, ratio, seed) in {'t256': (16, 256, 0.5, 7), 't1024': (4, 1024, 0.5, 8),
                                     't256_r75': (4, 256, 0.75, 9)}.items():
        torch.manual_seed(seed)
        md = ref_m.get_mask(B, T, ratio, 'cpu')
        torch.manual_seed(seed)
        noise = torch.rand(B, T)
        out[f'{tag}_noise'] = noise.numpy()
        out[f'{tag}_ratio'] = np.float64(ratio)
        out[f'{tag}_ids_keep'] = md['ids_keep'].numpy()
        out[f'{tag}_ids_restore'] = md['ids_restore'].numpy()
        out[f'{tag}_mask'] = md['mask'].numpy()
    np.savez_compressed(os.path.join(HERE, 'mask.npz'), **out)
    print('mask.npz written')


def gen_train(tag, model_type, R, B, seed, with_grads):
    cfg = O.make_cfg(model_type, img_resolution=R)
    P = O.init_params(cfg, seed=seed, dezero=True)
    net = build_ref(model_type, R, P)
    net.train()
    wrapped = Wrap(net)
    g = torch.Generator().manual_seed(seed + 100)
    images = 0.5 * torch.randn(B, 4, R, R, generator=g)
    cls = torch.randint(0, 1000, (B,), generator=g)
    keep = (torch.rand(B, 1, generator=g) >= 0.1).float()  # class dropout, train.py:208-209
    labels = one_hot(cls) * keep
    T = (R // 2) ** 2
    # replicate the reference's internal draws in order (loss.py:35,39; maskdit.py:102)
    torch.manual_seed(seed + 200)
    rnd_normal = torch.randn(B, 1, 1, 1)
    noise = torch.randn(B, 4, R, R)
    mask_noise = torch.rand(B, T)
    torch.manual_seed(seed + 200)
    loss_fn = Losses['edm']()
    loss = loss_fn(net=wrapped, images=images, labels=labels, mask_ratio=0.5, mae_loss_coef=0.1)
    out = dict(seed=np.int64(seed), B=np.int64(B), R=np.int64(R), images=images.numpy(), cls=cls.numpy(),
               keep=keep.numpy(), rnd_normal=rnd_normal.numpy(), noise=noise.numpy(),
               mask_noise=mask_noise.numpy(), loss=loss.detach().numpy())
    # D_yn (net output) for the same draws
    md = ref_m.get_mask  # noqa
    with torch.no_grad():
        mdict = {k: torch.from_numpy(v) for k, v in O.get_mask_from_noise(mask_noise.numpy(), 0.5).items()}
        sigma = (rnd_normal * 1.2 - 1.2).exp()
        D = net(images + noise * sigma, sigma, labels, mask_ratio=0.5, mask_dict=mdict)['x']
    out['D_yn'] = D.numpy()
    names = [k for k in P if k not in O.NON_TRAINABLE]
    out['param_names'] = np.array(names)
    pc = [checks(P[k]) for k in names]
    out['param_sums'] = np.stack([c[0] for c in pc])
    if with_grads:
        loss.mean().backward()
        sd = dict(net.named_parameters())
        gc = [checks(sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in names]
        out['grad_sums'] = np.stack([c[0] for c in gc])
        out['grad_samples'] = np.stack([c[1] for c in gc])
        # one optimizer step + EMA (train.py:141,223-230): torch AdamW == apex FusedAdam(adam_w_mode, wd 0)
        import copy
        ema = copy.deepcopy(net)
        opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.999),
                                eps=1e-8, weight_decay=0)
        opt.step()
        update_ema(ema, net, decay=0.9999)
        sd = dict(net.named_parameters())
        se = dict(ema.named_parameters())
        uc = [checks(sd[k]) for k in names]
        ec = [checks(se[k]) for k in names]
        out['upd_sums'] = np.stack([c[0] for c in uc])
        out['upd_samples'] = np.stack([c[1] for c in uc])
        out['ema_sums'] = np.stack([c[0] for c in ec])
        out['ema_samples'] = np.stack([c[1] for c in ec])
    np.savez_compressed(os.path.join(HERE, f'{tag}.npz'), **out)
    print(f'{tag}.npz written; loss[:4] =', loss.detach().numpy()[:4])


def gen_sampler(tag, model_type, R, seeds, num_steps, cfg_scale, seed, nocfg=True):
    cfg = O.make_cfg(model_type, img_resolution=R)
    P = O.init_params(cfg, seed=seed, dezero=True)
    net = build_ref(model_type, R, P)
    net.eval()
    rnd = StackedRandomGenerator('cpu', seeds)
    latents = rnd.randn([len(seeds), 4, R, R])
    cls = rnd.randint(1000, size=[len(seeds)])
    labels = torch.eye(1000)[cls]
    with torch.no_grad():
        z = ref_edm_sampler(net, latents.float(), labels.float(), randn_like=rnd.randn_like,
                            cfg_scale=cfg_scale, num_steps=num_steps)
        z_nocfg = ref_edm_sampler(net, latents.float(), labels.float(), randn_like=rnd.randn_like,
                                  cfg_scale=None, num_steps=num_steps) if nocfg else torch.zeros(0, dtype=torch.float64)
    np.savez_compressed(os.path.join(HERE, f'{tag}.npz'), seed=np.int64(seed), seeds=np.array(seeds),
                        latents=latents.numpy(), cls=cls.numpy(), num_steps=np.int64(num_steps),
                        cfg_scale=np.float64(cfg_scale), z=z.numpy(), z_nocfg=z_nocfg.numpy())
    print(f'{tag}.npz written; z std', z.std().item())


def gen_sampler_churn(tag='s2_sampler_churn', seeds=(200, 201, 202), num_steps=6, seed=9):
    """sample.py:51-53 with S_churn > 0: the stochastic branch (gamma > 0, x_hat = x_cur + noise) of the reference."""
    cfg = O.make_cfg('DiT-S/2', img_resolution=32)
    P = O.init_params(cfg, seed=seed, dezero=True)
    net = build_ref('DiT-S/2', 32, P)
    net.eval()
    seeds = list(seeds)
    rnd = StackedRandomGenerator('cpu', seeds)
    latents = rnd.randn([len(seeds), 4, 32, 32])
    cls = rnd.randint(1000, size=[len(seeds)])
    labels = torch.eye(1000)[cls]
    kw = dict(S_churn=10.0, S_min=0.05, S_max=50.0, S_noise=1.003)
    with torch.no_grad():
        z = ref_edm_sampler(net, latents.float(), labels.float(), randn_like=rnd.randn_like, cfg_scale=1.5,
                            num_steps=num_steps, **kw)
    np.savez_compressed(os.path.join(HERE, f'{tag}.npz'), seed=np.int64(seed), seeds=np.array(seeds), latents=latents.numpy(),
                        cls=cls.numpy(), num_steps=np.int64(num_steps), cfg_scale=np.float64(1.5), z=z.numpy(),
                        **{k: np.float64(v) for k, v in kw.items()})
    print(f'{tag}.npz written; z std', z.std().item())


def gen_moments():
    g = torch.Generator().manual_seed(5)
    mom = torch.cat([2.745 * torch.randn(4, 4, 32, 32, generator=g), torch.full((4, 4, 32, 32), -10.0)], 1)
    mom[0, 4:] = 25.0  # exercises the clamp (utils.py:61)
    torch.manual_seed(11)
    rn = torch.randn(4, 4, 32, 32)
    torch.manual_seed(11)
    z = ref_sample(mom)
    np.savez_compressed(os.path.join(HERE, 'moments.npz'), moments=mom.numpy(), randn=rn.numpy(), z=z.numpy())
    print('moments.npz written')


def gen_param_order():
    """`parameters()` order (names + shapes) of the reference module tree: optimizer state dicts are positional
    (train.py:141,153,264), so a drop-in must register its parameters in the same order."""
    import json
    out = {}
    for mt in ('DiT-S/2', 'DiT-XL/2'):
        net = ref_m.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type=mt,
                                          use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False)
        out[mt] = [[n, list(p.shape), bool(p.requires_grad)] for n, p in net.named_parameters()]
    with open(os.path.join(HERE, 'param_order.json'), 'w') as f:
        json.dump(out, f)
    print('param_order.json written')


def gen_vae():
    """Decode path of the reference's autoencoder.py (Decoder + post_quant_conv modules, ddconfig of get_model) loaded
    with the oracle's synthetic weights; the published autoencoder_kl.pth is not available offline.  Stores the image of
    ONE latent in full (fp16, 393 K values) + checksums of a second one."""
    import autoencoder as ref_ae  # noqa  (reference)
    from oracle import vae_oracle as VO
    ddconfig = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                    num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    dec = ref_ae.Decoder(**ddconfig)
    pq = torch.nn.Conv2d(4, 4, 1)
    P = VO.init_vae_params(seed=11)
    ref_keys = {'decoder.' + k: tuple(v.shape) for k, v in dec.state_dict().items()}
    ref_keys.update({'post_quant_conv.' + k: tuple(v.shape) for k, v in pq.state_dict().items()})
    assert ref_keys == VO.vae_param_shapes(), 'oracle key / shape table differs from the reference modules'
    dec.load_state_dict({k[len('decoder.'):]: v for k, v in P.items() if k.startswith('decoder.')}, strict=True)
    pq.load_state_dict({k[len('post_quant_conv.'):]: v for k, v in P.items() if k.startswith('post_quant_conv.')}, strict=True)
    dec.eval()
    g = torch.Generator().manual_seed(12)
    z = 0.5 * torch.randn(2, 4, 32, 32, generator=g)
    with torch.no_grad():
        img = dec(pq(z / 0.18215))  # FrozenAutoencoderKL.decode, autoencoder.py:449-453
    c = checks(img[1])
    np.savez_compressed(os.path.join(HERE, 'vae_decode.npz'), seed=np.int64(11), z=z.numpy(), img0=img[0].numpy().astype(np.float16),
                        img0_absmax=np.float64(img[0].abs().max().item()), img1_sums=c[0], img1_samples=c[1],
                        order=np.array(list(dec.state_dict().keys())))
    print('vae_decode.npz written; image std', img.std().item(), 'absmax', img.abs().max().item())


class _InjectedDraws:
    """Feeds the reference's OWN code (train_utils/loss.py:35,39, models/maskdit.py:102) prescribed random draws: while
    active, `torch.randn` / `torch.randn_like` / `torch.rand` hand out the queued tensors in call order (shape-checked)
    instead of drawing.  The arithmetic that consumes them is untouched reference code."""

    def __init__(self, *tensors):
        self.q = list(tensors)

    def _pop(self, shape):
        t = self.q.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.clone()

    def __enter__(self):
        self.saved = (torch.randn, torch.randn_like, torch.rand)
        torch.randn = lambda *size, **kw: self._pop(size[0] if len(size) == 1 and not isinstance(size[0], int) else size)
        torch.randn_like = lambda x, **kw: self._pop(x.shape)
        torch.rand = lambda *size, **kw: self._pop(size[0] if len(size) == 1 and not isinstance(size[0], int) else size)
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like, torch.rand = self.saved
        assert not self.q, 'the reference consumed fewer draws than were queued'


def gen_bs1024_grads(tag='xl2_bs1024_grads', seed=9, draw_seed=18, B=1024, S=16):
    """BASELINE configs[1] AT ITS OWN SIZE, the oracle end of the chain: the REFERENCE (not the oracle)
    runs train.py:216-220 -- `loss = loss_fn(net, images, labels, mask_ratio, mae_loss_coef); loss.mean().backward()` --
    over the 1024-sample batch of tests/test_40_full_batch_gpu.py::_bs1024_inputs(18) as 64 slices of 16 samples
    (gradients accumulate in `.grad`; the mean over 1024 is linear in the slices), with that test's draws injected into
    the reference's own `torch.randn` / `randn_like` / `rand` calls.  Stored: all 1024 per-sample losses; for EVERY
    parameter (sum, |sum|, L2) + 64 sampled entries (the `checks()` format); for the 16 named tensors of the test 4096
    sampled entries more, so that a relative L2 error can be estimated without the 2.7 GB of gradients.
    ~25 min on the build container's 8 cores."""
    import time
    cfg = O.make_cfg('DiT-XL/2', img_resolution=32)
    P = O.init_params(cfg, seed=seed, dezero=True)
    net = build_ref('DiT-XL/2', 32, P)
    net.train()
    wrapped = Wrap(net)
    T = 256
    g = torch.Generator().manual_seed(draw_seed)   # == _bs1024_inputs(draw_seed), statement for statement
    images = 0.5 * torch.randn(B, 4, 32, 32, generator=g)
    cls = torch.randint(0, 1000, (B,), generator=g)
    labels = torch.zeros(B, 1000)
    labels[torch.arange(B), cls] = 1
    labels *= (torch.rand(B, 1, generator=g) >= 0.1).float()
    rnd, noise = torch.randn(B, 1, 1, 1, generator=g), torch.randn(B, 4, 32, 32, generator=g)
    mnoise = torch.rand(B, T, generator=g)
    loss_fn = Losses['edm']()
    losses = []
    t0 = time.time()
    for lo in range(0, B, S):
        sl = slice(lo, lo + S)
        with _InjectedDraws(rnd[sl], noise[sl], mnoise[sl]):
            loss = loss_fn(net=wrapped, images=images[sl], labels=labels[sl], mask_ratio=0.5, mae_loss_coef=0.1)
        (loss.sum() / B).backward()
        losses.append(loss.detach())
        print(f'  slice {lo // S + 1}/{B // S}  {time.time() - t0:.0f} s', flush=True)
    names = [k for k in P if k not in O.NON_TRAINABLE]
    sd = dict(net.named_parameters())
    gc = [checks(sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in names]
    named = [str(k) for k in BS1024_NAMED]
    big = []
    for k in named:
        t64 = sd[k].grad.detach().double().flatten()
        big.append(t64[sample_idx(t64.numel(), k=4096)].numpy())
    np.savez_compressed(os.path.join(HERE, f'{tag}.npz'), seed=np.int64(seed), draw_seed=np.int64(draw_seed), B=np.int64(B),
                        loss=torch.cat(losses).numpy(), param_names=np.array(names),
                        grad_sums=np.stack([c[0] for c in gc]), grad_samples=np.stack([c[1] for c in gc]),
                        named=np.array(named), named_samples=np.stack(big))
    print(f'{tag}.npz written; loss mean {torch.cat(losses).mean().item():.6f}; {time.time() - t0:.0f} s')


# one gradient per GEMM site of a block + the tensors with their own backward kernels (the list of
# tests/test_40_full_batch_gpu.py::_NAMED_GRADS; the test asserts the two are equal)
BS1024_NAMED = ['model.blocks.0.attn.qkv.weight', 'model.blocks.13.attn.qkv.bias', 'model.blocks.13.attn.proj.weight',
                'model.blocks.27.mlp.fc1.weight', 'model.blocks.27.mlp.fc1.bias', 'model.blocks.5.mlp.fc2.weight',
                'model.blocks.5.adaLN_modulation.1.weight', 'model.blocks.20.adaLN_modulation.1.bias',
                'model.decoder_blocks.0.attn.qkv.weight', 'model.decoder_blocks.7.mlp.fc2.weight',
                'model.decoder_layer.linear.weight', 'model.mask_token', 'model.x_embedder.proj.weight',
                'model.final_layer.linear.weight', 'model.t_embedder.mlp.0.weight', 'model.y_embedder.embedding_table.weight']


JOBS = {
    'vae': gen_vae,
    'param_order': gen_param_order,
    'mask': gen_mask,
    'moments': gen_moments,
    's2_train': lambda: gen_train('s2_train', 'DiT-S/2', 32, 16, seed=0, with_grads=True),       # BASELINE config 1
    's2_512_fwd': lambda: gen_train('s2_512_fwd', 'DiT-S/2', 64, 2, seed=3, with_grads=False),   # T=1024 / L=512 shapes
    'xl2_fwd': lambda: gen_train('xl2_fwd', 'DiT-XL/2', 32, 2, seed=4, with_grads=False),        # hd=72 path
    's2_sampler': lambda: gen_sampler('s2_sampler', 'DiT-S/2', 32, [100, 101, 102, 103], 6, 1.5, seed=2),
    # round 2: gradients on the BASELINE configs themselves (configs[1]: XL/2 256; configs[3]: T=1024 / L=512)
    'xl2_train': lambda: gen_train('xl2_train', 'DiT-XL/2', 32, 2, seed=5, with_grads=True),
    's2_512_train': lambda: gen_train('s2_512_train', 'DiT-S/2', 64, 2, seed=6, with_grads=True),
    # round 3: configs[3] on the real model -- XL/2 at 512^2 latents (T = 1024, L = 512, hd 72), all gradients
    's2_sampler_churn': gen_sampler_churn,   # round 3: the S_churn > 0 branch (sample.py:51-53)
    'xl2_512_train': lambda: gen_train('xl2_512_train', 'DiT-XL/2', 64, 1, seed=8, with_grads=True),
    # configs[4]: XL/2, 50 Heun steps, cfg 1.5, the reference's fp32 network (sample.py:30-66)
    'xl2_sampler': lambda: gen_sampler('xl2_sampler', 'DiT-XL/2', 32, [0, 1], 50, 1.5, seed=7, nocfg=False),
    # round 5: configs[1] at its own size -- the reference over the 64 slices of the batch-1024 test (NOT part of the no-argument run: ~25 min)
    'xl2_bs1024_grads': gen_bs1024_grads,
}
SLOW_JOBS = {'xl2_bs1024_grads'}

if __name__ == '__main__':
    # python tests/golden/make_golden.py [job ...]   (no arguments: every fixture)
    for job in (sys.argv[1:] or [j for j in JOBS if j not in SLOW_JOBS]):
        JOBS[job]()
