"""End-to-end parity of the HIP path (through the reference's Python surface, which sits on
the C ABI) against (a) the committed golden fixtures produced by running the reference and
(b) the CPU oracle on the same seeded inputs.

Tolerances (stated per check): masking / indices bit-exact; the network computes its GEMMs and
attention in bf16 with fp32 accumulation (BASELINE: "AMP bf16") while the oracle / reference
fixture is fp32, so latents, losses and gradients are compared at bf16-appropriate tolerances, set at
<= 3x the worst value measured on MI355X (round 2; round 1 allowed 20x):
  D_yn (denoised latents, one net evaluation) : max |err| <= TOL_D = 3e-3 * max |ref|    (measured 6e-4 .. 1e-3)
  per-sample loss                             : rel TOL_LOSS = 1e-3                      (measured 9e-5 .. 3e-4)
  parameter gradients  : per tensor ||g - g_ref||_2 <= TOL_GRAD = 1e-2 * ||g_ref||_2     (measured 3e-3; + tiny abs floor)
  multi-step sampler latents (bf16 network vs the reference's fp32 network, error compounds over the
  Heun steps)                                 : stated at the test
fp32-only kernels (optimizer, EMA, EDM algebra) : 1e-5 relative.
"""
import copy
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import maskdit_amd as M
    from oracle import maskdit_oracle as O

DEV = 'cuda'
TOL_D, TOL_LOSS, TOL_GRAD = 3e-3, 1e-3, 1e-2


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _build(model_type, R, seed, train=True, mae=0.1):
    cfg = O.make_cfg(model_type, img_resolution=R)
    P = O.init_params(cfg, seed=seed, dezero=True)
    net = M.Precond_models['edm'](img_resolution=R, img_channels=4, num_classes=1000, model_type=model_type,
                                  use_decoder=True, mae_loss_coef=mae, pad_cls_token=False).to(DEV)
    missing = net.load_state_dict(P, strict=True)
    net.train(train)
    return cfg, P, net


def _inputs(g):
    B = int(g['B'])
    labels = torch.zeros(B, 1000)
    labels[torch.arange(B), torch.from_numpy(g['cls'])] = 1
    labels = labels * torch.from_numpy(g['keep'])
    return (torch.from_numpy(g['images']), labels, torch.from_numpy(g['rnd_normal']), torch.from_numpy(g['noise']),
            torch.from_numpy(g['mask_noise']))


def _relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _run_loss(net, g, mask_ratio=0.5):
    images, labels, rnd, noise, mnoise = _inputs(g)
    B, T = mnoise.shape
    md = M.get_mask(B, T, mask_ratio, DEV, noise=mnoise.to(DEV))
    loss_fn = M.Losses['edm']()
    loss = loss_fn.with_draws(net, images.to(DEV), labels.to(DEV), rnd.to(DEV), noise.to(DEV), md, mae_loss_coef=0.1)
    return loss, md


@pytest.mark.parametrize('name,model,R', [('s2_train.npz', 'DiT-S/2', 32), ('s2_512_fwd.npz', 'DiT-S/2', 64),
                                          ('xl2_fwd.npz', 'DiT-XL/2', 32)])
def test_forward_loss_vs_reference_fixture(golden_dir, name, model, R):
    g = _load(golden_dir, name)
    cfg, P, net = _build(model, R, int(g['seed']))
    with torch.no_grad():
        loss, md = _run_loss(net, g)
    # masking is integer work: bit-exact against the oracle's stable argsort of the same noise
    ref_md = O.get_mask_from_noise(g['mask_noise'], 0.5)
    assert np.array_equal(md['ids_keep'].cpu().numpy(), ref_md['ids_keep'])
    assert np.array_equal(md['ids_restore'].cpu().numpy(), ref_md['ids_restore'])
    assert np.array_equal(md['mask'].cpu().numpy(), ref_md['mask'])
    D = net.engine().plan(int(g['B']), True, False, md['ids_keep'].shape[1]).buf['D']
    e = _relmax(D, torch.from_numpy(g['D_yn']))
    print(f'[{name}] D_yn rel-to-max err {e:.3e}')
    assert e <= TOL_D
    rl = ((loss.cpu() - torch.from_numpy(g['loss'])).abs() / torch.from_numpy(g['loss']).abs()).max().item()
    print(f'[{name}] loss rel err {rl:.3e}')
    assert rl <= TOL_LOSS


def test_s2_train_step_vs_oracle_and_fixture(golden_dir):
    """BASELINE config 1 (S/2, bs 16, mask 0.5): loss, every parameter gradient, one fused
    AdamW + EMA step."""
    g = _load(golden_dir, 's2_train.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']))
    ema = copy.deepcopy(net)
    for p in ema.parameters():
        p.requires_grad_(False)
    opt = M.FusedAdam(net.parameters(), lr=1e-4, adam_w_mode=True, weight_decay=0)
    assert opt._arena is net.engine(), 'optimizer must run the single-kernel arena path'
    opt.zero_grad(set_to_none=True)
    loss, md = _run_loss(net, g)
    loss.mean().backward()
    # oracle on the same inputs (fp32 CPU)
    images, labels, rnd, noise, mnoise = _inputs(g)
    mdict = {k: torch.from_numpy(v) for k, v in O.get_mask_from_noise(g['mask_noise'], 0.5).items()}
    loss_ref, D_ref, grads_ref = O.loss_and_grads(P, cfg, images, labels, rnd, noise, mdict, 0.1)
    assert torch.allclose(loss_ref, torch.from_numpy(g['loss']), rtol=1e-4, atol=1e-6)  # oracle == reference fixture
    rl = ((loss.detach().cpu() - loss_ref).abs() / loss_ref.abs()).max().item()
    print(f'loss rel err {rl:.3e}')
    assert rl <= TOL_LOSS
    worst = ('', 0.0)
    params = dict(net.named_parameters())
    for k, gr in grads_ref.items():
        got = params[k].grad
        assert got is not None, k
        num = (got.detach().cpu().double() - gr.double()).norm().item()
        den = gr.double().norm().item()
        rel = num / (den + 1e-12)
        if rel > worst[1]:
            worst = (k, rel)
        assert num <= TOL_GRAD * den + 1e-7, f'{k}: grad rel L2 err {rel:.3e} (|g| = {den:.3e})'
    print(f'worst grad rel L2 err {worst[1]:.3e} at {worst[0]}')
    # ---- optimizer + EMA: fused kernel vs the oracle applied to the HIP gradients (fp32, 1e-5)
    g_hip = {k: params[k].grad.detach().cpu().clone() for k in grads_ref}
    p_before = {k: params[k].detach().cpu().clone() for k in grads_ref}
    opt.fuse_ema(ema, 0.9999)
    opt.step()
    M.update_ema(ema, net, 0.9999)  # folded into the step: must be a no-op now
    ema_params = dict(ema.named_parameters())
    for k in grads_ref:
        p, m, v, e = p_before[k].clone(), torch.zeros_like(p_before[k]), torch.zeros_like(p_before[k]), p_before[k].clone()
        O.adamw_step(p, g_hip[k], m, v, step=1, lr=1e-4)
        O.ema_update(e, p, 0.9999)
        assert torch.allclose(params[k].detach().cpu(), p, rtol=1e-5, atol=1e-7), k
        assert torch.allclose(ema_params[k].detach().cpu(), e, rtol=1e-5, atol=1e-7), k
        assert torch.allclose(opt.state[params[k]]['exp_avg'].cpu(), m, rtol=1e-5, atol=1e-9), k
    # the reference fixture's updated weights: sign-of-gradient sized Adam step (lr) => compare loosely
    from tests.golden.make_golden_idx import sample_idx
    names = [str(n) for n in g['param_names']]
    bad = 0
    for i, k in enumerate(names):
        idx = sample_idx(params[k].numel())
        got = params[k].detach().cpu().double().flatten()[idx].numpy()
        bad += int((np.abs(got - g['upd_samples'][i]) > 2.05e-4).sum())  # |step| <= lr = 1e-4 each way
    assert bad == 0
    # a second EMA call (not folded) runs the standalone kernel
    e_before = {k: ema_params[k].detach().cpu().clone() for k in grads_ref}
    M.update_ema(ema, net, 0.99)
    for k in list(grads_ref)[:8]:
        ref = 0.99 * e_before[k] + 0.01 * params[k].detach().cpu()
        assert torch.allclose(ema_params[k].detach().cpu(), ref, rtol=1e-5, atol=1e-7), k


@pytest.mark.parametrize('name,model,R', [('xl2_train.npz', 'DiT-XL/2', 32),      # BASELINE configs[1]: the benched model
                                          ('s2_512_train.npz', 'DiT-S/2', 64),    # configs[3] shapes: T = 1024, L = 512
                                          ('xl2_512_train.npz', 'DiT-XL/2', 64)])  # configs[3] itself: XL/2 at 512^2 (hd 72 x L 512 in real blocks)
def test_backward_on_baseline_configs_vs_reference(golden_dir, name, model, R):
    """Gradient parity ON the BASELINE configurations (round 1 had forward-only fixtures there): XL/2 (hd 72, the
    NF = 3 GEMM tiles, D = 1152 norm / gate kernels inside real blocks) and 512^2 latents (T = 1024, L = 512
    attention fwd + bwd).  Loss and D_yn against the reference-generated fixture; EVERY parameter gradient against
    the fp32 oracle (itself pinned to the same fixture by tests/test_oracle_golden.py) and its L2 norm against the
    reference's own."""
    g = _load(golden_dir, name)
    cfg, P, net = _build(model, R, int(g['seed']))
    net.zero_grad(set_to_none=True)
    loss, md = _run_loss(net, g)
    loss.mean().backward()
    B = int(g['B'])
    D = net.engine().plan(B, True, True, md['ids_keep'].shape[1]).buf['D']
    e = _relmax(D, torch.from_numpy(g['D_yn']))
    rl = ((loss.detach().cpu() - torch.from_numpy(g['loss'])).abs() / torch.from_numpy(g['loss']).abs()).max().item()
    print(f'[{name}] D_yn rel-to-max err {e:.3e}, loss rel err {rl:.3e}')
    assert e <= TOL_D and rl <= TOL_LOSS
    images, labels, rnd, noise, _ = _inputs(g)
    mdict = {k: torch.from_numpy(v) for k, v in O.get_mask_from_noise(g['mask_noise'], 0.5).items()}
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    _, _, grads_ref = O.loss_and_grads(P, cfg, images, labels, rnd, noise, mdict, 0.1)
    params = dict(net.named_parameters())
    names = [str(n) for n in g['param_names']]
    worst = ('', 0.0)
    for i, k in enumerate(names):
        got = params[k].grad.detach().cpu().double()
        den = grads_ref[k].double().norm().item()
        num = (got - grads_ref[k].double()).norm().item()
        rel = num / (den + 1e-12)
        if rel > worst[1]:
            worst = (k, rel)
        assert num <= TOL_GRAD * den + 1e-7, f'{k}: grad rel L2 err {rel:.3e} (|g| = {den:.3e})'
        ref_norm = float(g['grad_sums'][i][2])  # the reference's own ||g||_2
        assert abs(got.norm().item() - ref_norm) <= TOL_GRAD * ref_norm + 1e-7, (k, got.norm().item(), ref_norm)
    print(f'[{name}] {len(names)} gradients, worst rel L2 err {worst[1]:.3e} at {worst[0]}')


def test_xl2_sampler_50_steps_vs_reference_fixture(golden_dir):
    """BASELINE configs[4] as benchmarked: XL/2, 50 Heun steps (99 network evaluations), cfg 1.5, hipGraph path --
    against the reference's own edm_sampler output (fp32 network, fp64 state; tests/golden/make_golden.py).
    This engine evaluates the network in bf16 (deviation from sample.py:56, which runs it in fp32):
    the test MEASURES that drift after 99 evaluations and bounds it."""
    g = _load(golden_dir, 'xl2_sampler.npz')
    cfg, P, net = _build('DiT-XL/2', 32, int(g['seed']), train=False)
    labels = torch.eye(1000)[torch.from_numpy(g['cls'])].to(DEV)
    lat = torch.from_numpy(g['latents']).to(DEV)
    z = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=int(g['num_steps']))
    ref = torch.from_numpy(g['z'])
    e = _relmax(z, ref)
    rms = ((z.cpu() - ref).norm() / ref.norm()).item()
    print(f'XL/2 50-step sampler (bf16 net) vs reference (fp32 net): rel-to-max err {e:.3e}, rel L2 err {rms:.3e}')
    assert z.dtype == torch.float64 and bool(torch.isfinite(z).all())
    assert e <= 3e-3 and rms <= 3e-3  # measured 7.1e-4 / 8.0e-4 on MI355X (99 bf16 network evaluations)


TOL_F32 = 5e-6  # fp32-faithful path vs the reference's fp32 network: summation order only (measured on MI355X: 1.2e-7 after 99 XL/2 evaluations, <= 1.5e-6 on the S/2 fixtures)


def test_xl2_sampler_50_steps_vs_reference_fixture_fp32(golden_dir):
    """BASELINE configs[4] at the REFERENCE'S OWN PRECISION: sample.py:56 evaluates the network in
    fp32 (`net(x_hat.float(), ...)`, no autocast in generate.py).  `precision='fp32'` runs the exact-fp32 plan
    (csrc/f32path.hip: fp32 master weights, fp32 activations, v_mfma_f32_32x32x2_f32) through the same two captured graphs;
    after 99 network evaluations it must agree with the reference's edm_sampler output to fp32 rounding -- two orders of
    magnitude below the bf16 network's drift (7e-4 in the test above)."""
    g = _load(golden_dir, 'xl2_sampler.npz')
    cfg, P, net = _build('DiT-XL/2', 32, int(g['seed']), train=False)
    labels = torch.eye(1000)[torch.from_numpy(g['cls'])].to(DEV)
    lat = torch.from_numpy(g['latents']).to(DEV)
    z = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=int(g['num_steps']), precision='fp32')
    ref = torch.from_numpy(g['z'])
    e = _relmax(z, ref)
    rms = ((z.cpu() - ref).norm() / ref.norm()).item()
    print(f'XL/2 50-step sampler (fp32 net) vs reference (fp32 net): rel-to-max err {e:.3e}, rel L2 err {rms:.3e}')
    assert z.dtype == torch.float64 and bool(torch.isfinite(z).all())
    assert e <= TOL_F32 and rms <= TOL_F32
    # the bf16 plan of the same network is untouched by the fp32 one (separate plan-cache entries and graphs)
    zb = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=int(g['num_steps']))
    assert 1e-5 < _relmax(zb, ref) <= 3e-3


def test_eval_forward_cfg_and_sampler_fp32_vs_oracle(golden_dir):
    """The fp32-faithful plan behind every inference call form: plain eval forward, forward_with_cfg, the graph and the
    direct-launch sampler, the churn branch -- each against the fp32 oracle / reference fixture at TOL_F32."""
    cfg, P, net = _build('DiT-S/2', 32, seed=5, train=False)
    net.set_eval_precision('fp32')
    gcpu = torch.Generator().manual_seed(1)
    x = torch.randn(3, 4, 32, 32, generator=gcpu) * 3
    sigma = torch.tensor([0.3, 2.0, 40.0])
    y = torch.zeros(3, 1000)
    y[torch.arange(3), torch.tensor([1, 500, 999])] = 1
    with torch.no_grad():
        D = net(x.to(DEV), sigma.to(DEV), y.to(DEV))['x']
        ref = O.precond_forward(P, cfg, x, sigma, y, training=False)
        e1 = _relmax(D, ref)
        D2 = net(x.to(DEV), torch.tensor(2.5, dtype=torch.float64, device=DEV), y.to(DEV), 1.5)['x']
        ref2 = O.precond_forward(P, cfg, x, torch.tensor(2.5), y, cfg_scale=1.5, training=False)
        e2 = _relmax(D2, ref2)
        print(f'fp32 eval forward vs oracle: {e1:.2e}; with cfg: {e2:.2e}')
        assert e1 <= TOL_F32 and e2 <= TOL_F32
        net.set_eval_precision('bf16')
        Db = net(x.to(DEV), sigma.to(DEV), y.to(DEV))['x']
        assert 1e-5 < _relmax(Db, ref) <= TOL_D, 'the bf16 plan must still be the bf16 plan'
    with pytest.raises(ValueError):
        net.set_eval_precision('fp16')
    g = _load(golden_dir, 's2_sampler.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']), train=False)
    labels = torch.eye(1000)[torch.from_numpy(g['cls'])].to(DEV)
    lat = torch.from_numpy(g['latents']).to(DEV)
    n = int(g['num_steps'])
    z = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=n, precision='fp32')
    e = _relmax(z, torch.from_numpy(g['z']))
    z_direct = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=n, precision='fp32', use_graph=False)
    z2 = M.edm_sampler(net, lat, labels, cfg_scale=None, num_steps=n, precision='fp32')
    e2 = _relmax(z2, torch.from_numpy(g['z_nocfg']))
    print(f'fp32 sampler vs reference fixture: cfg {e:.2e}, no cfg {e2:.2e}')
    assert e <= TOL_F32 and e2 <= TOL_F32 and torch.equal(z, z_direct)
    g = _load(golden_dir, 's2_sampler_churn.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']), train=False)
    rnd = M.StackedRandomGenerator('cpu', [int(s) for s in g['seeds']])
    lat = rnd.randn([len(g['seeds']), 4, 32, 32])
    cls = rnd.randint(1000, size=[len(g['seeds'])])
    z = M.edm_sampler(net, lat.to(DEV), torch.eye(1000)[cls].to(DEV), cfg_scale=float(g['cfg_scale']), num_steps=int(g['num_steps']),
                      randn_like=lambda t: rnd.randn(list(t.shape), dtype=t.dtype).to(t.device), S_churn=float(g['S_churn']),
                      S_min=float(g['S_min']), S_max=float(g['S_max']), S_noise=float(g['S_noise']), precision='fp32')
    e3 = _relmax(z, torch.from_numpy(g['z']))
    print(f'fp32 sampler with churn vs reference fixture: {e3:.2e}')
    assert e3 <= TOL_F32 and net.eval_precision == 'bf16'
    with pytest.raises(NotImplementedError):
        net.engine().plan(8, True, True, 128, 'fp32')  # fp32 TRAINING is not provided (train.py --no_amp)


def test_eval_forward_fp32_at_1024_tokens_vs_oracle():
    """The fp32 plan at 512^2-latent token counts (T = 1024: BASELINE configs[3] shapes): attention takes the three-launch
    form (q k^T -> row softmax -> p v through a scores workspace) instead of the LDS-resident fused kernel."""
    cfg, P, net = _build('DiT-S/2', 64, seed=11, train=False)
    net.set_eval_precision('fp32')
    gcpu = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 64, 64, generator=gcpu) * 2
    sigma = torch.tensor([0.5, 7.0])
    y = torch.zeros(2, 1000)
    y[torch.arange(2), torch.tensor([3, 777])] = 1
    with torch.no_grad():
        D = net(x.to(DEV), sigma.to(DEV), y.to(DEV))['x']
        ref = O.precond_forward(P, cfg, x, sigma, y, training=False)
    e = _relmax(D, ref)
    print(f'fp32 eval forward, T = 1024: {e:.2e}')
    assert e <= TOL_F32
    assert any(k.startswith('scores_') for k in net.engine().plan(2, False, False, None, 'fp32').buf), 'expected the scores workspace'


@pytest.mark.parametrize('model_type', ['DiT-B/2', 'DiT-L/2', 'DiT-H/2'])
def test_other_model_sizes_train_and_eval_vs_oracle(model_type):
    """SURVEY 8a row a21 (the size configs of models/maskdit.py:649-715 at patch 2) END TO END on the GPU, against the CPU
    oracle on the same seeded inputs: B/2 (D 768, 12 heads of 64), L/2 (D 1024, 16 x 64), H/2 (D 1280, 16 x 80: the head
    dimension only this size has; 32 blocks) -- masked training loss + every gradient, bf16 and fp32-faithful eval forward
    (H/2's fp32 attention takes the three-launch form: hd 80 is outside the fused kernel's domain)."""
    cfg, P, net = _build(model_type, 32, seed=13)
    B = 4
    gcpu = torch.Generator().manual_seed(7)
    images = 0.5 * torch.randn(B, 4, 32, 32, generator=gcpu)
    labels = torch.zeros(B, 1000)
    labels[torch.arange(B), torch.randint(0, 1000, (B,), generator=gcpu)] = 1
    rnd, noise = torch.randn(B, 1, 1, 1, generator=gcpu), torch.randn(B, 4, 32, 32, generator=gcpu)
    mnoise = torch.rand(B, 256, generator=gcpu)
    md = M.get_mask(B, 256, 0.5, DEV, noise=mnoise.to(DEV))
    net.zero_grad(set_to_none=True)
    loss = M.Losses['edm']().with_draws(net, images.to(DEV), labels.to(DEV), rnd.to(DEV), noise.to(DEV), md, mae_loss_coef=0.1)
    loss.mean().backward()
    mdict = {k: torch.from_numpy(v) for k, v in O.get_mask_from_noise(mnoise.numpy(), 0.5).items()}
    assert torch.equal(md['ids_restore'].cpu(), mdict['ids_restore'])
    loss_ref, _, grads_ref = O.loss_and_grads(P, cfg, images, labels, rnd, noise, mdict, 0.1)
    rl = ((loss.detach().cpu() - loss_ref).abs() / loss_ref.abs()).max().item()
    assert rl <= TOL_LOSS, f'{model_type}: loss rel err {rl:.3e}'
    params = dict(net.named_parameters())
    worst = ('', 0.0)
    for k, gr in grads_ref.items():
        got = params[k].grad
        assert got is not None, k
        num = (got.detach().cpu().double() - gr.double()).norm().item()
        den = gr.double().norm().item()
        if num / (den + 1e-12) > worst[1]:
            worst = (k, num / (den + 1e-12))
        assert num <= TOL_GRAD * den + 1e-7, f'{model_type} {k}: grad rel L2 err {num / (den + 1e-12):.3e}'
    net.eval()
    x = torch.randn(2, 4, 32, 32, generator=gcpu) * 2
    sigma = torch.tensor([0.4, 9.0])
    with torch.no_grad():
        ref = O.precond_forward(P, cfg, x, sigma, labels[:2], training=False)
        eb = _relmax(net(x.to(DEV), sigma.to(DEV), labels[:2].to(DEV))['x'], ref)
        net.set_eval_precision('fp32')
        ef = _relmax(net(x.to(DEV), sigma.to(DEV), labels[:2].to(DEV))['x'], ref)
    print(f'{model_type}: loss rel err {rl:.2e}, worst grad rel L2 {worst[1]:.2e} at {worst[0]}, eval forward bf16 {eb:.2e} / fp32 {ef:.2e}')
    assert eb <= TOL_D and ef <= TOL_F32


@pytest.mark.parametrize('model_type', ['DiT-S/2', 'DiT-XL/2'])
def test_kmajor_weight_shadows_are_exact_transposes(model_type):
    """mdt_transpose_bf16_batched (the K-major bf16 shadows the data-gradient GEMMs read; round 6: 16-byte fast path): every
    entry of the engine's transpose table -- the stacked adaLN weight, every qkv / proj / fc1 / fc2 / decoder-layer /
    t-embedder matrix -- must be the exact transpose of its N-major shadow."""
    cfg, P, net = _build(model_type, 32, seed=2)
    eng = net.engine()
    eng.refresh_shadows()
    torch.cuda.synchronize()
    n = 0
    for src, dst, rows, cols in eng.lay.t_entries:
        a = eng.W16[src:src + rows * cols].view(rows, cols)
        b = eng.WT16[dst:dst + rows * cols].view(cols, rows)
        assert torch.equal(b.view(torch.int16), a.t().contiguous().view(torch.int16)), (src, dst, rows, cols)
        assert float(a.float().abs().max()) > 0
        n += 1
    assert n >= 4 * (cfg['depth'] + 8) + 2


def test_generic_net_autograd_path_matches_fused_loss(golden_dir):
    """The reference's own loss arithmetic (train_utils/loss.py:44-52) written in torch on top of
    net(...)['x'] must give the same loss and gradients as the fused EDMLoss."""
    import torch.nn.functional as F
    g = _load(golden_dir, 's2_train.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']))
    net.zero_grad(set_to_none=True)
    loss_fused, md = _run_loss(net, g)
    loss_fused.mean().backward()
    g_fused = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    images, labels, rnd, noise, _ = (t.to(DEV) for t in _inputs(g))
    sigma = (rnd * 1.2 - 1.2).exp()
    weight = (sigma ** 2 + 0.25) / (sigma * 0.5) ** 2
    y, n = images, noise * sigma
    out = net(y + n, sigma, labels, mask_ratio=0.5, mask_dict=md)
    D = out['x']
    loss = weight * (D - y) ** 2
    loss = F.avg_pool2d(loss.mean(dim=1), 2).flatten(1)
    unmask = 1 - out['mask']
    loss = (loss * unmask).sum(1) / unmask.sum(1)
    mae = O.mae_loss(cfg, (y + n).cpu(), D.cpu(), (1 - unmask).cpu()).to(DEV)  # differentiable torch ops
    loss = loss + 0.1 * mae
    assert torch.allclose(loss.detach(), loss_fused.detach(), rtol=1e-4, atol=1e-6)
    loss.mean().backward()
    for k, p in net.named_parameters():
        if p.requires_grad:
            num = (p.grad - g_fused[k]).norm().item()
            # same kernels both ways; fp32 atomic accumulation order differs run to run
            assert num <= 5e-3 * g_fused[k].norm().item() + 1e-8, k


def test_eval_forward_and_cfg_vs_oracle():
    cfg, P, net = _build('DiT-S/2', 32, seed=5, train=False)
    gcpu = torch.Generator().manual_seed(1)
    x = torch.randn(3, 4, 32, 32, generator=gcpu) * 3
    sigma = torch.tensor([0.3, 2.0, 40.0])
    y = torch.zeros(3, 1000)
    y[torch.arange(3), torch.tensor([1, 500, 999])] = 1
    with torch.no_grad():
        D = net(x.to(DEV), sigma.to(DEV), y.to(DEV))['x']
        ref = O.precond_forward(P, cfg, x, sigma, y, training=False)
        assert _relmax(D, ref) <= TOL_D
        # scalar sigma broadcast + CFG (sampling call form: positional cfg_scale, sample.py:56)
        D2 = net(x.to(DEV), torch.tensor(2.5, dtype=torch.float64, device=DEV), y.to(DEV), 1.5)['x']
        ref2 = O.precond_forward(P, cfg, x, torch.tensor(2.5), y, cfg_scale=1.5, training=False)
        assert _relmax(D2, ref2) <= TOL_D
        # eval mode + mask_ratio > 0: mask returned, no masking applied (models/maskdit.py:482)
        out = net(x.to(DEV), sigma.to(DEV), y.to(DEV), mask_ratio=0.5)
        assert 'mask' in out and out['mask'].shape == (3, 256) and out['mask'].sum(1).eq(128).all()
        assert _relmax(out['x'], ref) <= TOL_D


def test_sampler_vs_reference_fixture(golden_dir):
    g = _load(golden_dir, 's2_sampler.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']), train=False)
    labels = torch.eye(1000)[torch.from_numpy(g['cls'])].to(DEV)
    lat = torch.from_numpy(g['latents']).to(DEV)
    n = int(g['num_steps'])
    z = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=n)
    assert z.dtype == torch.float64 and z.shape == lat.shape
    e = _relmax(z, torch.from_numpy(g['z']))
    print(f'sampler (cfg, graph) rel-to-max err {e:.3e}')
    assert e <= 4e-3  # 11 bf16 network evaluations compound (measured 1.1e-3)
    z_again = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=n)  # graph replay
    assert torch.equal(z, z_again)
    z_generic = M.edm_sampler(net, lat, labels, cfg_scale=float(g['cfg_scale']), num_steps=n, use_graph=False)
    # same network kernels; the fp64 state algebra runs in torch instead of the fused kernels, and
    # last-bit differences in the state are amplified by bf16 rounding inside the network
    assert _relmax(z_generic, z) <= 5e-3
    z2 = M.edm_sampler(net, lat, labels, cfg_scale=None, num_steps=n)
    e2 = _relmax(z2, torch.from_numpy(g['z_nocfg']))
    print(f'sampler (no cfg) rel-to-max err {e2:.3e}')
    assert e2 <= 4e-3


def test_sampler_stochastic_churn_vs_reference_fixture(golden_dir):
    """S_churn > 0 (sample.py:51-53; ADVICE r2: the branch used to raise): HIP network evaluations + torch fp64 state
    algebra against the reference's own output with the same per-seed noise streams."""
    g = _load(golden_dir, 's2_sampler_churn.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']), train=False)
    rnd = M.StackedRandomGenerator('cpu', [int(s) for s in g['seeds']])
    lat = rnd.randn([len(g['seeds']), 4, 32, 32])
    cls = rnd.randint(1000, size=[len(g['seeds'])])

    def randn_like(x):  # the fixture's generators live on the CPU
        return rnd.randn(list(x.shape), dtype=x.dtype).to(x.device)

    z = M.edm_sampler(net, lat.to(DEV), torch.eye(1000)[cls].to(DEV), cfg_scale=float(g['cfg_scale']), num_steps=int(g['num_steps']),
                      randn_like=randn_like, S_churn=float(g['S_churn']), S_min=float(g['S_min']), S_max=float(g['S_max']),
                      S_noise=float(g['S_noise']))
    assert z.dtype == torch.float64 and z.shape == lat.shape
    e = _relmax(z, torch.from_numpy(g['z']))
    print(f'sampler with churn: rel-to-max err {e:.3e}')
    assert e <= 4e-3  # 11 bf16 network evaluations, as test_sampler_vs_reference_fixture


def test_state_dict_roundtrip_and_rebinding():
    cfg, P, net = _build('DiT-S/2', 32, seed=6)
    sd = net.state_dict()
    assert set(sd) == set(P)
    for k in P:
        assert torch.equal(sd[k].cpu(), P[k]), k
    ema = copy.deepcopy(net)
    assert ema.engine() is not net.engine()
    for (k, a), (_, b) in zip(net.named_parameters(), ema.named_parameters()):
        assert torch.equal(a, b) and a.data_ptr() != b.data_ptr(), k
    # in-place edits through torch (load_state_dict / foreign optimizers) are seen by the engine
    x = torch.randn(2, 4, 32, 32, device=DEV)
    s = torch.tensor([1.0, 2.0], device=DEV)
    net.eval()
    with torch.no_grad():
        a = net(x, s)['x'].clone()
        net.model.final_layer.linear.weight.mul_(2.0)
        b = net(x, s)['x'].clone()
        assert not torch.allclose(a, b)
        # a GEMM weight (read through the bf16 / K-major SHADOWS, not the fp32 master) edited in place AFTER a
        # forward: `p.mul_` bumps only that Parameter's version counter, not the arena's
        net.model.blocks[0].attn.qkv.weight.mul_(1.5)
        c = net(x, s)['x'].clone()
        assert not torch.allclose(b, c), 'stale bf16 shadow: an in-place parameter edit was not seen'
        # load_state_dict after a forward (train.py:149 resume; generate.py:49) must take effect as well
        net.load_state_dict(P, strict=True)
        d = net(x, s)['x'].clone()
        ref_net = _build('DiT-S/2', 32, seed=6, train=False)[2]
        assert torch.equal(d, ref_net(x, s)['x']), 'load_state_dict after a forward left stale shadows'


def test_backward_of_an_overwritten_forward_raises(golden_dir):
    """Activations live in the plan's buffers: two forwards of the same shape followed by a backward through the
    FIRST one is legal autograd (two losses summed) but would differentiate the second forward's activations --
    the engine refuses instead of returning a wrong gradient."""
    g = _load(golden_dir, 's2_train.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']))
    l1, _ = _run_loss(net, g)
    l2, _ = _run_loss(net, g)
    with pytest.raises(RuntimeError, match='overwritten by a later forward'):
        (l1.mean() + l2.mean()).backward()
    net.zero_grad(set_to_none=True)
    l3, _ = _run_loss(net, g)  # the normal order still works afterwards
    l3.mean().backward()
    assert net.model.blocks[0].attn.qkv.weight.grad is not None


def test_gradient_arena_follows_autograd_semantics_per_parameter(golden_dir):
    """`.grad` of every parameter is a view into one arena and the HIP backward ACCUMULATES into it; what happens to a
    parameter between two backwards is decided PER PARAMETER, as autograd would (one sentinel tensor
    used to decide for the whole arena): grad dropped -> starts from zero, grad kept -> accumulates, a foreign tensor
    assigned as .grad -> accumulation continues from its value; and the fused optimizer skips parameters without a
    gradient like apex does (the one-kernel arena step only runs when every parameter has one)."""
    g = _load(golden_dir, 's2_train.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']))
    params = dict(net.named_parameters())
    drop = ['model.mask_token', 'model.blocks.3.mlp.fc1.weight']
    keep = ['model.blocks.3.mlp.fc2.weight', 'model.x_embedder.proj.weight', 'model.blocks.0.adaLN_modulation.1.bias']
    foreign = 'model.final_layer.linear.bias'
    net.zero_grad(set_to_none=True)
    _run_loss(net, g)[0].mean().backward()
    G1 = {k: params[k].grad.detach().clone() for k in drop + keep + [foreign]}
    for k in drop:
        params[k].grad = None
    params[foreign].grad = torch.full_like(params[foreign], 0.25)
    _run_loss(net, g)[0].mean().backward()
    for k in drop:
        assert _relmax(params[k].grad, G1[k]) <= 1e-5, f'{k}: a dropped gradient must restart from zero'
    for k in keep:
        assert _relmax(params[k].grad, 2 * G1[k]) <= 1e-5, f'{k}: a kept gradient must accumulate'
    assert _relmax(params[foreign].grad, G1[foreign] + 0.25) <= 1e-5, 'a foreign .grad must be accumulated onto'
    eng = net.engine()
    assert params[drop[1]].grad.data_ptr() == eng.view(eng.G, drop[1]).data_ptr(), '.grad must point back into the arena'
    # optimizer: parameters whose grad is None are skipped
    opt = M.FusedAdam(net.parameters(), lr=1e-3)
    before = {k: params[k].detach().clone() for k in drop + keep}
    params[drop[1]].grad = None
    opt.step()
    assert torch.equal(params[drop[1]], before[drop[1]]), 'a parameter without gradient must not move'
    assert not torch.equal(params[keep[0]], before[keep[0]]) and not torch.equal(params[drop[0]], before[drop[0]])
    # requires_grad toggled after the first backward (finetuning with frozen tensors): its gradient stays untouched
    params[keep[1]].requires_grad_(False)
    params[keep[1]].grad = None
    net.zero_grad(set_to_none=True)
    _run_loss(net, g)[0].mean().backward()
    assert params[keep[1]].grad is None and params[keep[0]].grad is not None


def test_fails_loudly_off_gpu():
    net = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2')
    with pytest.raises(M.MaskDiTLibError):
        net(torch.zeros(1, 4, 32, 32), torch.ones(1))


def test_sample_moments_and_class_dropout_vs_reference_fixture(golden_dir):
    g = _load(golden_dir, 'moments.npz')  # produced by the reference's utils.sample (seed 11)
    mom = torch.from_numpy(g['moments']).to(DEV)
    torch.manual_seed(11)
    rn_ref = torch.randn(4, 4, 32, 32, device=DEV)  # device philox stream differs from the CPU fixture's draw ...
    from maskdit_amd._lib import call
    z = torch.empty(4, 4, 32, 32, device=DEV)
    rn = torch.from_numpy(g['randn']).to(DEV)       # ... so feed the fixture's own randn to the kernel
    call('mdt_sample_moments', mom.data_ptr(), rn.data_ptr(), z.data_ptr(), 4, 4 * 32 * 32, 0.18215,
         torch.cuda.current_stream().cuda_stream)
    np.testing.assert_allclose(z.cpu().numpy(), g['z'], rtol=2e-6, atol=2e-6)
    torch.manual_seed(11)
    z2 = M.sample(mom)  # draws randn_like(mean) itself: same draw as torch.randn on the device
    ref2 = O.sample_moments(mom.cpu(), rn_ref.cpu())
    np.testing.assert_allclose(z2.cpu().numpy(), ref2.numpy(), rtol=2e-6, atol=2e-6)
    y = torch.eye(1000, device=DEV)[:64].contiguous()
    torch.manual_seed(3)
    u = torch.rand(64, 1, device=DEV)
    torch.manual_seed(3)
    M.class_dropout_(y, 0.3)
    assert torch.equal(y, torch.eye(1000, device=DEV)[:64] * (u >= 0.3).float())


def test_optimizer_and_model_checkpoint_roundtrip(tmp_path):
    """train.py:259-271 / 147-162: {"model","ema","opt"} saved with torch.save, reloaded into fresh objects,
    training continues identically (apex-style optimizer state layout: group['step'], exp_avg, exp_avg_sq)."""
    g = _load(os.path.join(os.path.dirname(__file__), 'golden'), 's2_train.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']))
    opt = M.FusedAdam(net.parameters(), lr=1e-3, adam_w_mode=True, weight_decay=0)

    def one_step(n, o):
        o.zero_grad(set_to_none=True)
        loss, _ = _run_loss(n, g)
        loss.mean().backward()
        o.step()

    one_step(net, opt)
    path = os.path.join(tmp_path, '0000001.pt')
    torch.save({'model': net.state_dict(), 'opt': opt.state_dict()}, path)
    sd = opt.state_dict()
    assert sd['param_groups'][0]['step'] == 1 and set(sd['state'][next(iter(sd['state']))]) >= {'exp_avg', 'exp_avg_sq'}
    ck = torch.load(path, map_location='cpu')
    net2 = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2',
                                   use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False).to(DEV)
    net2.load_state_dict(ck['model'], strict=True)
    net2.train()
    opt2 = M.FusedAdam(net2.parameters(), lr=1e-3, adam_w_mode=True, weight_decay=0)
    opt2.load_state_dict(ck['opt'])
    assert opt2._arena is net2.engine() and opt2.param_groups[0]['step'] == 1
    one_step(net, opt)
    one_step(net2, opt2)
    worst = 0.0
    for (k, a), (_, b) in zip(net.named_parameters(), net2.named_parameters()):
        if a.requires_grad:
            # identical state + identical inputs; only fp32 atomic accumulation order may differ
            d = (a - b).abs().max().item()
            worst = max(worst, d)
            assert d <= 2e-5, (k, d)
    print('resume: max param difference after one more step', worst)


@pytest.mark.parametrize('ratio', [0.3, 0.6, 0.9])
def test_arbitrary_mask_ratio_vs_oracle(golden_dir, ratio):
    """Mask-ratio schedules (train_utils/helper.py:9-27) produce kept-token counts that are not multiples of
    the 64-row tile (ratio 0.3 -> 179 of 256 kept, 0.6 -> 102, 0.9 -> 25): the encoder runs on the count rounded up, with
    the padding rows masked out of attention and cut out of the gradient.  Loss and every parameter
    gradient against the fp32 oracle at the exact count."""
    g = _load(golden_dir, 's2_train.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']))
    images, labels, rnd, noise, mnoise = _inputs(g)
    B, T = mnoise.shape
    md = M.get_mask(B, T, ratio, DEV, noise=mnoise.to(DEV))
    Lv = int(T * (1 - ratio))
    assert md['ids_keep'].shape[1] == Lv and Lv % 64 != 0
    net.zero_grad(set_to_none=True)
    loss = M.Losses['edm']().with_draws(net, images.to(DEV), labels.to(DEV), rnd.to(DEV), noise.to(DEV), md, mae_loss_coef=0.1)
    loss.mean().backward()
    ref_md = O.get_mask_from_noise(g['mask_noise'], ratio)
    assert np.array_equal(md['ids_keep'].cpu().numpy(), ref_md['ids_keep'])
    mdict = {k: torch.from_numpy(v) for k, v in ref_md.items()}
    loss_ref, _, grads_ref = O.loss_and_grads(P, cfg, images, labels, rnd, noise, mdict, 0.1)
    rl = ((loss.detach().cpu() - loss_ref).abs() / loss_ref.abs()).max().item()
    assert rl <= TOL_LOSS, rl
    params = dict(net.named_parameters())
    worst = ('', 0.0)
    for k, gr in grads_ref.items():
        num = (params[k].grad.detach().cpu().double() - gr.double()).norm().item()
        den = gr.double().norm().item()
        if num / (den + 1e-12) > worst[1]:
            worst = (k, num / (den + 1e-12))
        assert num <= TOL_GRAD * den + 1e-7, f'{k}: grad rel L2 err {num / (den + 1e-12):.3e}'
    print(f'ratio {ratio}: kept {Lv}, loss rel err {rl:.2e}, worst grad rel err {worst[1]:.2e} ({worst[0]})')


def test_one_plan_serves_changing_kept_counts_and_plans_are_cached(golden_dir):
    """(a) PassPlan.set_valid() -- ONE plan re-used with a different kept-token count inside the same 64-row
    bucket (a mask-ratio schedule does this every step): 179 then 170 then 179 of 256 through the same net, each loss and
    every gradient against the fp32 oracle at the exact count; (b) the plan cache is LRU under a memory budget: an
    unmasked evaluation and a second training shape in between must NOT evict the training plan (round 2 dropped every
    training plan whenever another training shape was requested, losing anything patched onto it)."""
    g = _load(golden_dir, 's2_train.npz')
    cfg, P, net = _build('DiT-S/2', 32, int(g['seed']))
    images, labels, rnd, noise, mnoise = _inputs(g)
    B, T = mnoise.shape
    eng = net.engine()
    plan_ids = []
    for Lv in (179, 170, 179):
        ratio = 1.0 - (Lv + 0.5) / T
        assert int(T * (1 - ratio)) == Lv
        md = M.get_mask(B, T, ratio, DEV, noise=mnoise.to(DEV))
        net.zero_grad(set_to_none=True)
        loss = M.Losses['edm']().with_draws(net, images.to(DEV), labels.to(DEV), rnd.to(DEV), noise.to(DEV), md, mae_loss_coef=0.1)
        loss.mean().backward()
        pl = eng.plan(B, True, True, Lv)
        plan_ids.append(id(pl))
        pl.fwd.marker = 'patched'  # stands for bench.py's GemmTimer.wrap instrumentation
        assert pl.Lv == Lv and pl.L == 192
        mdict = {k: torch.from_numpy(v) for k, v in O.get_mask_from_noise(g['mask_noise'], ratio).items()}
        loss_ref, _, grads_ref = O.loss_and_grads(P, cfg, images, labels, rnd, noise, mdict, 0.1)
        rl = ((loss.detach().cpu() - loss_ref).abs() / loss_ref.abs()).max().item()
        assert rl <= TOL_LOSS, (Lv, rl)
        params = dict(net.named_parameters())
        for k, gr in grads_ref.items():
            num = (params[k].grad.detach().cpu().double() - gr.double()).norm().item()
            assert num <= TOL_GRAD * gr.double().norm().item() + 1e-7, f'kept {Lv}: {k}'
        # other shapes in between: eval forward (unmasked) and a training plan of another bucket
        net.eval()
        with torch.no_grad():
            net(images[:4].to(DEV), torch.ones(4, device=DEV), labels[:4].to(DEV))
        net.train()
        md2 = M.get_mask(B, T, 0.5, DEV, noise=mnoise.to(DEV))
        with torch.no_grad():
            M.Losses['edm']().with_draws(net, images.to(DEV), labels.to(DEV), rnd.to(DEV), noise.to(DEV), md2, mae_loss_coef=0.1)
    assert len(set(plan_ids)) == 1, 'the 192-row training plan was rebuilt instead of re-used'
    assert getattr(eng.plan(B, True, True, 179).fwd, 'marker', None) == 'patched'
    assert len(eng._plans) >= 3
