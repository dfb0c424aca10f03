"""Pins the CPU oracle (oracle/maskdit_oracle.py) against fixtures produced by running the
reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import maskdit_oracle as O

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _tie_free(noise):
    s = np.sort(noise, axis=1)
    return (np.diff(s, axis=1) != 0).all(axis=1)


@pytest.mark.parametrize('tag', ['t256', 't1024', 't256_r75'])
def test_mask_matches_reference(golden_dir, tag):
    g = _load(golden_dir, 'mask.npz')
    noise, ratio = g[f'{tag}_noise'], float(g[f'{tag}_ratio'])
    md = O.get_mask_from_noise(noise, ratio)
    ok = _tie_free(noise)
    assert ok.sum() >= 1
    # bit-exact on tie-free rows (integer / index work)
    assert (md['ids_keep'][ok] == g[f'{tag}_ids_keep'][ok]).all()
    assert (md['ids_restore'][ok] == g[f'{tag}_ids_restore'][ok]).all()
    assert (md['mask'][ok] == g[f'{tag}_mask'][ok]).all()
    # permutation invariants on every row
    B, T = noise.shape
    ar = np.arange(T)[None].repeat(B, 0)
    assert (np.sort(md['ids_shuffle'], axis=1) == ar).all()
    assert (np.take_along_axis(md['ids_shuffle'], md['ids_restore'], axis=1) == ar).all()
    L = int(T * (1 - ratio))
    assert (md['mask'].sum(1) == T - L).all()
    assert (np.take_along_axis(md['mask'], md['ids_keep'], axis=1) == 0).all()


def test_sample_moments(golden_dir):
    g = _load(golden_dir, 'moments.npz')
    z = O.sample_moments(torch.from_numpy(g['moments']), torch.from_numpy(g['randn']))
    np.testing.assert_allclose(z.numpy(), g['z'], rtol=1e-6, atol=1e-6)


def _inputs(g, cfg):
    B = int(g['B'])
    labels = torch.zeros(B, 1000)
    labels[torch.arange(B), torch.from_numpy(g['cls'])] = 1
    labels = labels * torch.from_numpy(g['keep'])
    md = O.get_mask_from_noise(g['mask_noise'], 0.5)
    mask_dict = {k: torch.from_numpy(v) for k, v in md.items()}
    return (torch.from_numpy(g['images']), labels, torch.from_numpy(g['rnd_normal']),
            torch.from_numpy(g['noise']), mask_dict)


def _sums(t):
    t = t.double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), t.norm().item()])


def _check_param_recipe(P, g):
    names = [str(n) for n in g['param_names']]
    got = np.stack([_sums(P[k]) for k in names])
    np.testing.assert_allclose(got, g['param_sums'], rtol=1e-9, atol=1e-9)
    return names


@pytest.mark.parametrize('name,model,R', [('s2_512_fwd.npz', 'DiT-S/2', 64), ('xl2_fwd.npz', 'DiT-XL/2', 32)])
def test_forward_loss_matches_reference(golden_dir, name, model, R):
    g = _load(golden_dir, name)
    cfg = O.make_cfg(model, img_resolution=R)
    P = O.init_params(cfg, seed=int(g['seed']), dezero=True)
    _check_param_recipe(P, g)
    images, labels, rnd, noise, md = _inputs(g, cfg)
    with torch.no_grad():
        loss, D = O.edm_loss(P, cfg, images, labels, rnd, noise, md, mae_loss_coef=0.1)
    np.testing.assert_allclose(loss.numpy(), g['loss'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(D.numpy(), g['D_yn'], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('name,model,R', [('s2_train.npz', 'DiT-S/2', 32),        # BASELINE configs[0]
                                          ('xl2_train.npz', 'DiT-XL/2', 32),     # configs[1]: the benched model
                                          ('s2_512_train.npz', 'DiT-S/2', 64),   # configs[3] shapes: T = 1024, L = 512
                                          ('xl2_512_train.npz', 'DiT-XL/2', 64)])  # configs[3]: XL/2 at 512^2 latents
def test_train_step_matches_reference(golden_dir, name, model, R):
    """loss, EVERY parameter gradient (L2 norm + 64 sampled entries per tensor), one AdamW + EMA step --
    against what the reference itself produced (tests/golden/make_golden.py: gen_train)."""
    g = _load(golden_dir, name)
    cfg = O.make_cfg(model, img_resolution=R)
    P = O.init_params(cfg, seed=int(g['seed']), dezero=True)
    names = _check_param_recipe(P, g)
    images, labels, rnd, noise, md = _inputs(g, cfg)
    loss, D, grads = O.loss_and_grads(P, cfg, images, labels, rnd, noise, md, 0.1)
    np.testing.assert_allclose(loss.numpy(), g['loss'], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(D.numpy(), g['D_yn'], rtol=1e-4, atol=2e-5)
    from tests.golden.make_golden_idx import sample_idx
    for i, k in enumerate(names):
        gs = g['grad_sums'][i]
        got = _sums(grads[k])
        assert abs(got[2] - gs[2]) <= 2e-4 * gs[2] + 1e-9, (k, got, gs)
        idx = sample_idx(grads[k].numel())
        np.testing.assert_allclose(grads[k].double().flatten()[idx].numpy(), g['grad_samples'][i],
                                   rtol=2e-3, atol=2e-4 * gs[2] / max(1.0, grads[k].numel() ** 0.5) + 1e-9,
                                   err_msg=k)
    # optimizer + EMA
    for i, k in enumerate(names):
        p, m, v, e = P[k].clone(), torch.zeros_like(P[k]), torch.zeros_like(P[k]), P[k].clone()
        O.adamw_step(p, grads[k], m, v, step=1, lr=1e-4)
        O.ema_update(e, p, 0.9999)
        idx = sample_idx(p.numel())
        np.testing.assert_allclose(p.double().flatten()[idx].numpy(), g['upd_samples'][i], rtol=1e-6, atol=2e-7,
                                   err_msg=k)
        np.testing.assert_allclose(e.double().flatten()[idx].numpy(), g['ema_samples'][i], rtol=1e-6, atol=2e-7,
                                   err_msg=k)


def test_sampler_matches_reference(golden_dir):
    g = _load(golden_dir, 's2_sampler.npz')
    cfg = O.make_cfg('DiT-S/2', img_resolution=32)
    P = O.init_params(cfg, seed=int(g['seed']), dezero=True)
    labels = torch.eye(1000)[torch.from_numpy(g['cls'])]
    lat = torch.from_numpy(g['latents'])
    z = O.edm_sampler(P, cfg, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=int(g['num_steps']))
    assert z.dtype == torch.float64
    np.testing.assert_allclose(z.numpy(), g['z'], rtol=1e-4, atol=1e-4)
    z2 = O.edm_sampler(P, cfg, lat, labels, cfg_scale=None, num_steps=int(g['num_steps']))
    np.testing.assert_allclose(z2.numpy(), g['z_nocfg'], rtol=1e-4, atol=1e-4)


def test_sampler_churn_matches_reference(golden_dir):
    """The stochastic branch of sample.py:51-53 (S_churn > 0): noise from the reference's per-seed generators, whose
    state continues from the latent / label draws (generate.py order)."""
    from maskdit_amd.latents import StackedRandomGenerator
    g = _load(golden_dir, 's2_sampler_churn.npz')
    cfg = O.make_cfg('DiT-S/2', img_resolution=32)
    P = O.init_params(cfg, seed=int(g['seed']), dezero=True)
    rnd = StackedRandomGenerator('cpu', [int(s) for s in g['seeds']])
    lat = rnd.randn([len(g['seeds']), 4, 32, 32])
    cls = rnd.randint(1000, size=[len(g['seeds'])])
    assert np.array_equal(lat.numpy(), g['latents']) and np.array_equal(cls.numpy(), g['cls'])
    z = O.edm_sampler(P, cfg, lat, torch.eye(1000)[cls], cfg_scale=float(g['cfg_scale']), num_steps=int(g['num_steps']),
                      S_churn=float(g['S_churn']), S_min=float(g['S_min']), S_max=float(g['S_max']), S_noise=float(g['S_noise']),
                      randn_like=rnd.randn_like)
    np.testing.assert_allclose(z.numpy(), g['z'], rtol=1e-4, atol=1e-4)


def test_t_steps_schedule():
    t = O.edm_t_steps(50)
    assert t.dtype == torch.float64 and t.shape == (51,)
    assert abs(t[0].item() - 80.0) < 1e-9 and abs(t[49].item() - 0.002) < 1e-12 and t[50].item() == 0.0
    assert (t[:-1] > t[1:]).all()


@pytest.mark.skipif(os.environ.get('MASKDIT_SLOW') != '1', reason='~100 TFLOP of fp32 CPU work: MASKDIT_SLOW=1 to run')
def test_xl2_sampler_matches_reference(golden_dir):
    """BASELINE configs[4] on the oracle: XL/2, 50 Heun steps, cfg 1.5, 2 seeds, fp32 network / fp64 state."""
    g = _load(golden_dir, 'xl2_sampler.npz')
    cfg = O.make_cfg('DiT-XL/2', img_resolution=32)
    P = O.init_params(cfg, seed=int(g['seed']), dezero=True)
    labels = torch.eye(1000)[torch.from_numpy(g['cls'])]
    z = O.edm_sampler(P, cfg, torch.from_numpy(g['latents']), labels, cfg_scale=float(g['cfg_scale']),
                      num_steps=int(g['num_steps']))
    np.testing.assert_allclose(z.numpy(), g['z'], rtol=1e-4, atol=1e-4)


def test_vae_decode_oracle_matches_reference(golden_dir):
    """the decode path of the reference's autoencoder.py (its own Decoder / post_quant_conv modules with the
    oracle's synthetic weights; tests/golden/make_golden.py: gen_vae) against oracle/vae_oracle.py."""
    from oracle import vae_oracle as VO
    from tests.golden.make_golden_idx import sample_idx
    g = _load(golden_dir, 'vae_decode.npz')
    P = VO.init_vae_params(seed=int(g['seed']))
    with torch.no_grad():
        img = VO.vae_decode(P, torch.from_numpy(g['z']))
    assert img.shape == (2, 3, 256, 256)
    ref0 = g['img0'].astype(np.float32)  # stored as fp16: 1e-3 relative quantisation
    np.testing.assert_allclose(img[0].numpy(), ref0, rtol=2e-3, atol=2e-3 * float(g['img0_absmax']))
    got = _sums(img[1])
    np.testing.assert_allclose(got, g['img1_sums'], rtol=1e-4)
    idx = sample_idx(img[1].numel())
    np.testing.assert_allclose(img[1].double().flatten()[idx].numpy(), g['img1_samples'], rtol=1e-3, atol=1e-4)
