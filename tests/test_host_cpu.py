"""CPU checks of the host logic around the HIP path: arena layout vs the oracle's (reference)
state-dict shapes, slab coverage for the DP all-reduce, schedules, config loader, module
surface (state-dict keys, attributes the reference callers read), loud failure without a GPU."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import maskdit_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('model,R', [('DiT-S/2', 32), ('DiT-XL/2', 32), ('DiT-XL/2', 64), ('DiT-B/2', 32), ('DiT-L/2', 32)])
def test_arena_layout_matches_reference_state_dict(model, R):
    from maskdit_amd.engine import Layout, make_spec, param_table
    sp = make_spec(model, R, 4, 1000)
    cfg = O.make_cfg(model, img_resolution=R)
    ref = {k: v for k, v in O.param_shapes(cfg).items() if k not in O.NON_TRAINABLE}
    tab = dict(param_table(sp))
    assert tab == ref  # same keys, same shapes as the reference checkpoint layout
    lay = Layout(sp)
    spans = sorted((lay.off[k], lay.off[k] + int(np.prod(v))) for k, v in tab.items())
    assert all(a % 8 == 0 for a, _ in spans)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))  # disjoint
    assert spans[-1][1] <= lay.n
    # adaLN weights are stacked densely: one GEMM produces every modulation vector
    assert lay.ada_b - lay.ada_w == sp.n_mod * sp.D
    # DP slabs tile the arena exactly
    cov = np.zeros(lay.n, dtype=np.int32)
    for lo, hi in lay.slabs.values():
        cov[lo:hi] += 1
    assert (cov == 1).all()
    if model == 'DiT-XL/2' and R == 32:
        assert sum(int(np.prod(v)) for v in tab.values()) == 730_115_216


def test_module_surface_cpu():
    import maskdit_amd as M
    net = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2',
                                  use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False)
    cfg = O.make_cfg('DiT-S/2')
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == O.param_shapes(cfg)
    P = O.init_params(cfg)
    assert torch.allclose(net.model.pos_embed, P['model.pos_embed']) and torch.allclose(net.model.decoder_pos_embed, P['model.decoder_pos_embed'])
    assert not net.model.pos_embed.requires_grad and net.model.mask_token.requires_grad
    assert net.model.patch_size == 2 and net.model.out_channels == 4 and net.model.extras == 0 and net.model.cls_token is None
    assert (net.img_resolution, net.img_channels, net.num_classes, net.sigma_min, net.sigma_max) == (32, 4, 1000, 0, float('inf'))
    assert float(net.model.final_layer.linear.weight.abs().max()) == 0.0  # adaLN-zero style init (maskdit.py:380-383)
    net.load_state_dict(P, strict=True)
    with pytest.raises(M.MaskDiTLibError):
        net(torch.zeros(1, 4, 32, 32), torch.ones(1))
    with pytest.raises(M.MaskDiTLibError):
        M.Losses['edm']()(net, torch.zeros(1, 4, 32, 32), torch.zeros(1, 1000))
    with pytest.raises(NotImplementedError):
        M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2', pad_cls_token=True)
    # inference precision switch (round 6): default bf16, validated, and it is a host-side attribute (no GPU needed to set it)
    assert net.eval_precision == 'bf16' and net.set_eval_precision('fp32') is net and net.eval_precision == 'fp32'
    with pytest.raises(ValueError):
        net.set_eval_precision('tf32')
    with pytest.raises(M.MaskDiTLibError):   # ... and the fp32 path fails as loudly off-GPU as the bf16 one
        net(torch.zeros(1, 4, 32, 32), torch.ones(1))


def test_schedules_and_config():
    from maskdit_amd.schedule import get_mask_ratio_fn, get_one_hot, load_config, lr_rampup_factor
    xs = np.linspace(0, 1, 7)
    for k in range(2, 7):  # train_utils/helper.py:10-19
        f = get_mask_ratio_fn(f'cosine{k}', 0.75, 0.25)
        np.testing.assert_allclose([f(x) for x in xs], 0.5 * np.cos(np.pi * xs / 2) ** k + 0.25, rtol=1e-12)
    np.testing.assert_allclose([get_mask_ratio_fn('exp', 0.5, 0.1)(x) for x in xs], 0.4 * np.exp(-7 * xs) + 0.1)
    np.testing.assert_allclose([get_mask_ratio_fn('linear', 0.5, 0.1)(x) for x in xs], 0.4 * xs + 0.1)
    assert get_mask_ratio_fn('constant', 0.5)(0.3) == 0.5
    with pytest.raises(ValueError):
        get_mask_ratio_fn('cos4')  # the typo in configs/finetune/*.yaml of the reference stays an error
    assert lr_rampup_factor(0, 1024, 0) == 0.0 and lr_rampup_factor(1, 1024, 0) == 1.0
    assert math.isclose(lr_rampup_factor(10, 1024, 100), 10 * 1024 / 100e3)
    oh = get_one_hot(torch.tensor([3, 0]), 5)
    assert oh.tolist() == [[0, 0, 0, 1, 0], [1, 0, 0, 0, 0]]
    cfg = load_config(os.path.join(ROOT, 'configs', 'xl2-256-synthetic.yaml'))
    assert cfg.model.model_type == 'DiT-XL/2' and cfg.train.batchsize == 128 and cfg.model.mae_loss_coef == 0.1


def test_gemm_nt8_wait_counts():
    """The compile-time vmcnt operands of gemm_nt8 (maskdit_amd/csrc/gemm_nt8.hip: wait_count) restated:
    simulate the steady-state issue sequence and check that the awaited load is exactly the oldest of
    the W+1 most recent ones (loads retire in order)."""
    def c_issue(p, nf, rpp):
        return 1 + (rpp if p < nf else 0)

    def wait_count(p, nf, rpp):
        w = rpp if ((p + 2) & 3) < nf else 0
        for d in range(5, -1, -1):
            w += c_issue((p - d) % 4, nf, rpp)
        if p == 2:
            w = min(w, (4 - nf) + c_issue(0, nf, rpp) + c_issue(1, nf, rpp) + c_issue(2, nf, rpp))
        return w

    for nf, rpp in ((2, 1), (3, 1), (4, 1), (2, 2), (3, 2)):  # rpp = B rounds per phase (1: 8 waves, 2: 4 waves)
        seq = []  # (kind, tile, slot)
        for g in range(0, 40):
            t, p = divmod(g, 4)
            seq.append(('A', t + 2, p))
            if p < nf:
                seq += [('B', t + 2, p * rpp + r) for r in range(rpp)]
            if g < 12:
                continue
            # end of phase g: next phase prefetches A slot (p+2)&3 of tile t (p<=1) or t+1, and at p==2 all B of t+1
            tile = t if p <= 1 else t + 1
            awaited = [('A', tile, (p + 2) & 3)] + ([('B', t + 1, j) for j in range(nf * rpp)] if p == 2 else [])
            w = wait_count(p, nf, rpp)
            landed = set(seq[:len(seq) - w])
            assert all(a in landed for a in awaited), (nf, g, w)
            # and the wait is not needlessly strict: with W+1 outstanding some awaited load would be in flight
            assert any(a not in set(seq[:len(seq) - w - 1]) for a in awaited), (nf, g, w)


def test_gemm_nt8_drain_counts():
    """Tail of the same pipeline (maskdit_amd/csrc/gemm_nt8.hip: drain_count): in the last pair of
    K-tiles nothing is issued any more; the operand must still cover the load the next phase needs."""
    def c_issue(p, nf, rpp):
        return 1 + (rpp if p < nf else 0)

    def drain_count(d, nf, rpp):
        if d >= 6:
            return 0
        w = rpp if ((d + 2) & 3) < nf else 0
        for e in range(d - 5, 0):
            w += c_issue(e % 4, nf, rpp)
        if d == 2:
            w = min(w, 4 - nf)
        return w

    for nf, rpp in ((2, 1), (3, 1), (4, 1), (2, 2), (3, 2)):
        seq = []
        T = 6  # K-tiles; the last pair (tiles 4, 5) is the drain body
        for g in range(0, 4 * (T - 2)):  # steady phases issue tile t+2
            t, p = divmod(g, 4)
            seq.append(('A', t + 2, p))
            if p < nf:
                seq += [('B', t + 2, p * rpp + r) for r in range(rpp)]
        for d in range(6):
            t, p = divmod(4 * (T - 2) + d, 4)
            tile = t if p <= 1 else t + 1
            awaited = [('A', tile, (p + 2) & 3)] + ([('B', t + 1, j) for j in range(nf * rpp)] if p == 2 else [])
            awaited = [a for a in awaited if a[1] < T]
            w = drain_count(d, nf, rpp)
            landed = set(seq[:len(seq) - w]) if w else set(seq)
            assert all(a in landed for a in awaited), (nf, d, w)


def test_bench_cpu_baseline_is_bounded():
    """bench.py's CPU leg (the oracle's training step in a child process) returns a rate within its
    wall-clock budget, and gives up cleanly -- instead of stalling the benchmark -- when it cannot."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t0 = time.time()
    r = bench.cpu_baseline(4, 'DiT-S/2', 32, budget_s=120)
    assert r['kind'] == 'port' and r['unit'] == 'img/s' and r['value'] and r['value'] > 0 and time.time() - t0 < 125
    assert '3 timed steps' in r['sample'] and 'warm-up' in r['sample'] and r['cores'] >= 1
    assert r['sampler']['unit'] == 'samples/s' and r['sampler']['value'] > 0                # the CPU sampler leg
    t0 = time.time()
    r2 = bench.cpu_baseline(16, 'DiT-XL/2', 32, budget_s=3)
    assert r2['value'] is None and time.time() - t0 < 15


@pytest.mark.parametrize('model', ['DiT-S/2', 'DiT-XL/2'])
def test_parameter_registration_order_is_the_references(golden_dir, model):
    """Optimizer state dicts are positional (train.py:141 gives `model.parameters()` to FusedAdam, :153 / :264 load /
    save its state by index): `parameters()` must enumerate in the reference module tree's order.  Fixture =
    `named_parameters()` of the reference itself (tests/golden/make_golden.py: param_order)."""
    import json
    import maskdit_amd as M
    with open(os.path.join(golden_dir, 'param_order.json')) as f:
        ref = json.load(f)[model]
    if model == 'DiT-XL/2':  # names / shapes / flags only: skip the 2.9 GB of parameter storage
        from maskdit_amd.engine import make_spec, param_table
        from maskdit_amd.precond import reference_param_order
        sp = make_spec(model, 32, 4, 1000)
        shapes = {n: list(s) for n, s in param_table(sp)}
        shapes['model.pos_embed'], shapes['model.decoder_pos_embed'] = [1, sp.T, sp.D], [1, sp.T, sp.Dd]
        got = [['model.' + n, shapes['model.' + n], n not in ('pos_embed', 'decoder_pos_embed')] for n in reference_param_order(sp)]
    else:
        net = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type=model,
                                      use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False)
        got = [[n, list(p.shape), bool(p.requires_grad)] for n, p in net.named_parameters()]
    assert got == ref


def test_optimizer_state_of_a_reference_checkpoint_maps_by_position(golden_dir):
    """A reference `opt` state dict (apex layout: param_groups[0]['params'] = 0..N-1 in `model.parameters()` order,
    state[i] = {exp_avg, exp_avg_sq}; frozen parameters have no entry) restores onto the parameters of the same NAME."""
    import json
    import maskdit_amd as M
    with open(os.path.join(golden_dir, 'param_order.json')) as f:
        ref = json.load(f)['DiT-S/2']
    state = {i: {'exp_avg': torch.full(shape, float(i)), 'exp_avg_sq': torch.full(shape, i + 0.5)}
             for i, (name, shape, rg) in enumerate(ref) if rg}
    sd = {'state': state, 'param_groups': [{'lr': 3e-4, 'bias_correction': True, 'betas': (0.9, 0.999), 'eps': 1e-8,
                                           'weight_decay': 0, 'step': 7, 'params': list(range(len(ref)))}]}
    net = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2',
                                  use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False)
    opt = M.FusedAdam(net.parameters(), lr=1e-4)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]['step'] == 7 and opt.param_groups[0]['lr'] == 3e-4
    index = {name: i for i, (name, _, _) in enumerate(ref)}
    for name, p in net.named_parameters():
        if p.requires_grad:
            st = opt.state[p]
            assert float(st['exp_avg'].flatten()[0]) == float(index[name]) and st['exp_avg'].shape == p.shape, name
            assert float(st['exp_avg_sq'].flatten()[-1]) == index[name] + 0.5, name


@pytest.mark.parametrize('fixture', ['s2_sampler.npz', 'xl2_sampler.npz'])
def test_stacked_random_generator_matches_reference(golden_dir, fixture):
    """The per-seed latents / class draws of the sampling entry point: fixtures hold what the reference's own
    utils.StackedRandomGenerator('cpu', seeds) produced (tests/golden/make_golden.py: gen_sampler)."""
    import maskdit_amd as M
    g = np.load(os.path.join(golden_dir, fixture))
    seeds = [int(s) for s in g['seeds']]
    rnd = M.StackedRandomGenerator('cpu', seeds)
    lat = rnd.randn([len(seeds), 4, 32, 32])
    cls = rnd.randint(1000, size=[len(seeds)])
    assert np.array_equal(lat.numpy(), g['latents']) and np.array_equal(cls.numpy(), g['cls'])
    # a sample depends on its seed only, not on its batch-mates
    solo = M.StackedRandomGenerator('cpu', seeds[-1:])
    assert torch.equal(solo.randn([1, 4, 32, 32])[0], lat[-1])
    assert rnd.randn_like(lat).shape == lat.shape
    with pytest.raises(AssertionError):
        rnd.randn([len(seeds) + 1, 4])


def test_seed_batches_shard_like_the_reference():
    """sample.py:233-235 restated with its own arithmetic: tensor_split into ceil(n / (max_bs * world)) * world chunks,
    rank-strided; every seed exactly once, no batch above the maximum."""
    import maskdit_amd as M
    for n, mb, world in [(64, 64, 1), (50000, 50, 8), (100, 64, 8), (7, 4, 2), (1, 64, 4)]:
        seeds = list(range(1000, 1000 + n))
        got = [M.seed_batches(seeds, mb, r, world) for r in range(world)]
        num = ((n - 1) // (mb * world) + 1) * world
        ref = [b.tolist() for b in torch.as_tensor(seeds).tensor_split(num)]
        for r in range(world):
            assert got[r] == ref[r::world]
        flat = sorted(s for rb in got for b in rb for s in b)
        assert flat == seeds and max(len(b) for rb in got for b in rb) <= mb


def test_reference_yaml_configs_parse_and_are_supported():
    """Every training / test YAML the reference ships parses with maskdit_amd.schedule.load_config and names a
    supported model / flag set.  Reads /root/reference (build container only; skipped elsewhere)."""
    import glob
    from maskdit_amd.engine import make_spec
    from maskdit_amd.schedule import get_mask_ratio_fn, load_config
    files = sorted(glob.glob('/root/reference/configs/**/*.yaml', recursive=True))
    if not files:
        pytest.skip('/root/reference is not present on this machine')
    for f in files:
        cfg = load_config(f)
        mc = cfg.model
        sp = make_spec(mc.model_type, mc.in_size, mc.in_channels, mc.num_classes, mc.use_decoder, mc.get('mae_loss_coef', 0.1))
        assert sp.T in (256, 1024) and not mc.pad_cls_token and mc.get('ext_feature_dim', 0) == 0, f
        if 'train' in cfg:
            assert cfg.train.batchsize > 0 and cfg.train.lr > 0 and cfg.log.ckpt_every > 0, f
            name = mc.get('mask_ratio_fn', 'constant')
            if name == 'cos4':  # configs/finetune/imagenet256-latent-cos.yaml: not a name helper.py:9-27 accepts either
                with pytest.raises(ValueError):
                    get_mask_ratio_fn(name, mc.mask_ratio, mc.get('mask_ratio_min', 0))
            else:
                get_mask_ratio_fn(name, mc.mask_ratio, mc.get('mask_ratio_min', 0))(0.5)


def test_vae_decoder_surface_cpu(golden_dir):
    """Decode-side drop-in of autoencoder.py: reference state-dict keys / shapes in the reference's module order, the full
    checkpoint (with encoder.* / quant_conv.* entries) loads, missing decode keys are an error, no CPU compute path."""
    import maskdit_amd as M
    from maskdit_amd import autoencoder as AE
    from oracle import vae_oracle as VO
    g = np.load(os.path.join(golden_dir, 'vae_decode.npz'))
    ref_order = ['post_quant_conv.weight', 'post_quant_conv.bias'] + ['decoder.' + str(k) for k in g['order']]
    table = AE.decoder_param_table()
    assert [n for n, _ in table] == ref_order  # `parameters()` order of the reference Decoder (self.up.insert(0, ...))
    assert {n: tuple(s) for n, s in table} == VO.vae_param_shapes()
    vae = AE.get_model(None)
    P = VO.init_vae_params(seed=3)
    full = dict(P)
    full['encoder.conv_in.weight'] = torch.zeros(128, 3, 3, 3)   # present in autoencoder_kl.pth, not on the decode path
    full['quant_conv.weight'] = torch.zeros(8, 8, 1, 1)
    vae.load_state_dict(full)
    assert torch.equal(vae.state_dict()['decoder.mid.attn_1.q.weight'], P['decoder.mid.attn_1.q.weight'])
    broken = {k: v for k, v in P.items() if k != 'decoder.conv_out.bias'}
    with pytest.raises(RuntimeError):
        vae.load_state_dict(broken)
    with pytest.raises(M.MaskDiTLibError):
        vae.decode(torch.zeros(1, 4, 32, 32))
    with pytest.raises(NotImplementedError):
        vae.encode(torch.zeros(1, 3, 256, 256))


def test_get_latest_ckpt_and_logger(tmp_path, capsys):
    """train.py helpers mirroring utils.py:22-34 (highest-numbered <step>.pt, None when there is none) and
    utils.py:169-225 (stdout / stderr tee into the experiment's log.txt)."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location('mdt_train_entry', os.path.join(ROOT, 'train.py'))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    d = tmp_path / 'checkpoints'
    assert tr.get_latest_ckpt(str(d)) is None          # no directory yet: a fresh run, not an error
    d.mkdir()
    assert tr.get_latest_ckpt(str(d)) is None
    for name in ('0000100.pt', '0002000.pt', '0000500.pt', 'notes.txt', 'best.pt', '0009999.pt.tmp'):
        (d / name).write_bytes(b'')
    assert tr.get_latest_ckpt(str(d)) == os.path.join(str(d), '0002000.pt')
    log = tmp_path / 'log.txt'
    lg = tr.Logger(str(log))
    try:
        print('step 1 loss 0.5')
        print('a warning', file=sys.stderr)
    finally:
        lg.close()
    assert sys.stdout is not lg and sys.stderr is not lg, 'close() must restore the streams'
    text = log.read_text()
    assert 'step 1 loss 0.5' in text and 'a warning' in text
    lg2 = tr.Logger(str(log))  # append mode: a resumed run keeps the earlier lines
    try:
        print('resumed')
    finally:
        lg2.close()
    assert log.read_text().startswith(text) and 'resumed' in log.read_text()
    # the tee still answers what libraries ask a text stream (tqdm / faulthandler / warnings), closing twice is fine
    lg3 = tr.Logger(str(log))
    try:
        assert sys.stdout.isatty() == lg3.stdout.isatty() and sys.stderr.encoding == lg3.stdout.encoding
    finally:
        lg3.close()
        lg3.close()
    # ... and checkpoints are written atomically / a truncated newest file is skipped on auto-resume
    import torch
    tr.save_ckpt_atomic({'step': 100}, str(d / '0000100.pt'))
    tr.save_ckpt_atomic({'step': 500}, str(d / '0000500.pt'))
    assert not (d / '0000500.pt.tmp').exists()
    msgs = []
    path, ck = tr.load_newest_valid_ckpt(str(d), log=msgs.append)   # 0002000.pt is an empty (truncated) file
    assert path == os.path.join(str(d), '0000500.pt') and ck == {'step': 500}
    assert len(msgs) == 1 and '0002000.pt' in msgs[0]
    assert tr.list_ckpts(str(d))[0].endswith('0002000.pt') and len(tr.list_ckpts(str(d))) == 3
    # a *.pt.tmp that is still being written (a previous incarnation inside save_ckpt_atomic) is left alone;
    # only one that has been quiet for min_age_s is a leftover
    fresh, old = d / '0003000.pt.tmp', d / '0000050.pt.tmp'
    fresh.write_bytes(b'x')
    old.write_bytes(b'x')
    os.utime(str(old), (1.0e9, 1.0e9))
    msgs = []
    tr.remove_stale_tmp(str(d), log=msgs.append)
    assert fresh.exists() and not old.exists() and (d / '0009999.pt.tmp').exists()
    assert any('left alone' in m and '0003000' in m for m in msgs) and any('removed' in m and '0000050' in m for m in msgs)
    tr.remove_stale_tmp(str(d), log=msgs.append, min_age_s=0.0)
    assert not fresh.exists() and not (d / '0009999.pt.tmp').exists()


def test_zero1_slab_ownership_tiles_every_slab():
    """ZeRO-1 ownership (maskdit_amd/ddp.py slab_pieces, maskdit_amd/zero.py owned_pieces): for every world size the
    owned ranges of all ranks tile each slab exactly once, equal pieces are 8-element aligned (16-byte bf16 / 32-byte
    fp32 boundaries for the collectives) and the remainder (< 8 W elements) stays with the last rank."""
    from maskdit_amd.ddp import slab_pieces
    from maskdit_amd.zero import owned_pieces
    from maskdit_amd.engine import Layout, make_spec
    slabs = Layout(make_spec('DiT-XL/2', 32, 4, 1000)).slabs
    for world in (1, 2, 3, 4, 8, 16):
        for name, (lo, hi) in slabs.items():
            q, pieces, tail = slab_pieces(lo, hi, world)
            assert q % 8 == 0 and len(pieces) == world
            assert all(e - a == q for a, e in pieces) and pieces[0][0] == lo
            assert tail == (lo + world * q, hi) and 0 <= hi - tail[0] < 8 * world
        n = max(hi for _, hi in slabs.values())
        cover = np.zeros(n, dtype=np.int32)
        for r in range(world):
            mine = owned_pieces(slabs, world, r)
            assert all(e > a for a, e in mine)
            for a, e in mine:
                cover[a:e] += 1
        lo_all = min(lo for lo, _ in slabs.values())
        assert lo_all == 0 and (cover == 1).all(), f'world {world}: ranks must own every element exactly once'
        sizes = [sum(e - a for a, e in owned_pieces(slabs, world, r)) for r in range(world)]
        assert max(sizes) - min(sizes) < 8 * world * len(slabs), 'shards are balanced to within the per-slab remainders'


def test_isa_wait_audit():
    """tools/check_waits.py over the ISA hipcc emits for the product flags: every hand-counted
    `s_waitcnt vmcnt(N)` / bare `s_barrier` of attn_bwd_dma, gemm_nt8<*> (8- and 4-wave forms, the cross-tile prefetch
    hand-over, the implicit-GEMM convolution), gemm_tn8<*> and the LayerNorm / 128x128 GEMM kernels is consistent with the
    instruction stream around it on ALL control-flow paths -- and the same audit FAILS on the round-3 attention wait
    (`-DMDT_REGRESS_R3_ATTN_WAIT`: the peeled first item reaches the buffer hand-over with six LDS-DMA pieces and four
    register loads in flight)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('check_waits', os.path.join(ROOT, 'tools', 'check_waits.py'))
    cw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cw)
    assert cw.main([]) == 0, 'the wait audit found a problem in the product build: run `python tools/check_waits.py`'
    rep = cw.Report()
    cw.compile_asm('attention.hip', ['MDT_REGRESS_R3_ATTN_WAIT'])
    cw.check_file('attention.hip', ['MDT_REGRESS_R3_ATTN_WAIT'], rep)
    assert len(rep.errors) == 1 and 'attn_bwd_dma_kernel<72>' in rep.errors[0] and 'no_dma' in rep.errors[0], rep.errors
    assert 'DDDDDDLLLL' in rep.errors[0]  # six LDS-DMA pieces + lse + three O fragments behind the 3 guaranteed pieces


def test_attention_lds_image_is_conflict_free():
    """tools/attn_lds_conflicts.py: the hd-72 tile image of the single-pass attention kernels (SpCfg<72>, round 4) has no
    bank conflict for either fragment-read pattern under gfx950's lane groups, while the 144-byte rows of rounds 2-3 took
    1.8x / 2x their LDS cycles (the 0.39-0.47 conflict fraction of profiles/r3_pmc_counters.txt)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('attn_lds_conflicts', os.path.join(ROOT, 'tools', 'attn_lds_conflicts.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    (r, ri), (t, ti), (r0, _), (t0, _) = m.main()
    assert r == ri and t == ti, 'the split image must be conflict-free'
    assert r0 > 1.7 * ri and t0 > 1.9 * ti, 'the enumeration should reproduce the measured conflicts of the old image'


def test_plan_cache_evicts_least_recently_used_across_engines():
    """maskdit_amd/engine.py `_evict_lru_plan` (two live engines -- train.py's net and ema -- used to budget and
    evict on their own, so the second one could die with OutOfMemoryError while the first one's idle 244 GB training plan
    stayed cached): the victim is the least-recently-used cached plan of ANY live engine on the device, its eviction hooks
    (the sampler's captured graphs) run, and engines on other devices are left alone.  Host logic only -- stand-in objects."""
    import types
    from maskdit_amd import engine as E

    class FakePlan:
        def __init__(self, last_use, log, name):
            self.last_use, self.name = last_use, name
            self.evict_hooks = [lambda: log.append(name)]

    class FakeEngine:
        def __init__(self, device):
            self.device = torch.device(device)
            self._plans = {}

        __hash__ = object.__hash__

    log = []
    a, b, other = FakeEngine('cuda:0'), FakeEngine('cuda:0'), FakeEngine('cuda:1')
    a._plans = {'train1024': FakePlan(3, log, 'a.train1024'), 'eval128': FakePlan(9, log, 'a.eval128')}
    b._plans = {'eval128': FakePlan(5, log, 'b.eval128')}
    other._plans = {'x': FakePlan(1, log, 'other.x')}       # older than everything, but on another device
    saved = list(E.LIVE_ENGINES)
    try:
        for e in saved:
            E.LIVE_ENGINES.discard(e)
        for e in (a, b, other):
            E.LIVE_ENGINES.add(e)
        assert E._evict_lru_plan('cuda:0') and log == ['a.train1024'] and 'train1024' not in a._plans
        assert E._evict_lru_plan(torch.device('cuda', 0)) and log[-1] == 'b.eval128' and not b._plans
        assert E._evict_lru_plan('cuda:0') and log[-1] == 'a.eval128'
        assert not E._evict_lru_plan('cuda:0'), 'nothing left on cuda:0'
        assert list(other._plans) == ['x'], 'plans on other devices must not be touched'
    finally:
        for e in (a, b, other):
            E.LIVE_ENGINES.discard(e)
        for e in saved:
            E.LIVE_ENGINES.add(e)


def test_nt8o_counter_protocol_model():
    """csrc/gemm_nt8o.hip synchronises its MMA / loader / epilogue waves through monotonic LDS counters instead of
    s_barrier.  tools/nt8o_protocol_model.py restates each role's walk (same order of waits, adds, fragment reads and LDS-DMA
    issues, same use counters and thresholds) and explores EVERY interleaving of the roles and of the asynchronous, per-loader
    in-order DMA completions for small parameters: no deadlock, no fragment read before all loaders' pieces of that fill have
    landed or after the stage is being overwritten, no refill before every MMA wave has finished the previous fill, no tile
    picked up by an epilogue wave before every MMA wave's y stores of it were acknowledged -- across tile boundaries, for
    K-tile counts that are and are not multiples of the ring depth, 1-3 loaders, with and without epilogue waves.  The FIRST
    form of the kernel (one summed ydone counter) must fail the check: it does, at the workgroup's last tile."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('nt8o_protocol_model', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools',
                                                                                     'nt8o_protocol_model.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for cfg in (dict(W=2, NL=2, NE=1, nk=4, tiles=2), dict(W=2, NL=2, NE=1, nk=5, tiles=2), dict(W=2, NL=1, NE=0, nk=7, tiles=2),
                dict(W=2, NL=2, NE=2, nk=4, tiles=2), dict(W=1, NL=3, NE=1, nk=4, tiles=3)):
        assert mod.check(**cfg) > 1000
    with pytest.raises(AssertionError, match='acknowledged its y stores'):
        mod.check(W=2, NL=2, NE=1, nk=4, tiles=2, summed_ydone=True)
