"""Data-parallel path on the real engine: two ranks train the same S/2 model on two halves of a batch through
maskdit_amd.DataParallel; the averaged slab-wise gradients must equal (a) the fp32 ORACLE's gradients of the full
batch (bf16-compute tolerance 1e-2 per tensor) and (b) the single-process HIP gradients of the full batch
(5e-3: same kernels, different tile counts / atomic order), and the replicas must stay bit-identical after the
optimizer step.  On a one-GPU box both ranks share cuda:0 over a gloo rendezvous (RCCL needs one device per
rank); with >= 2 visible GPUs the same worker also runs over the 'nccl' backend (= RCCL, the production path:
ReduceOp.AVG on arena views, collectives on RCCL's stream overlapping the backward kernels)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, backend='gloo'):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    devno = rank if backend == 'nccl' else 0
    torch.cuda.set_device(devno)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        import maskdit_amd as M
        from oracle import maskdit_oracle as O
        dev = f'cuda:{devno}'
        cfg = O.make_cfg('DiT-S/2', img_resolution=32)
        P = O.init_params(cfg, seed=rank, dezero=True)  # different replicas: construction must broadcast rank 0's
        net = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2',
                                      use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False).to(dev)
        net.load_state_dict(P)
        net.train()
        dp = M.DataParallel(net)
        P0 = O.init_params(cfg, seed=0, dezero=True)
        for k, p in net.named_parameters():
            assert torch.equal(p.detach().cpu(), P0[k]), f'{k} was not broadcast from rank 0'
        B = 16
        g = torch.Generator().manual_seed(5)
        images = 0.5 * torch.randn(B, 4, 32, 32, generator=g)
        labels = torch.zeros(B, 1000)
        labels[torch.arange(B), torch.randint(0, 1000, (B,), generator=g)] = 1
        rnd, noise = torch.randn(B, 1, 1, 1, generator=g), torch.randn(B, 4, 32, 32, generator=g)
        mnoise = torch.rand(B, 256, generator=g)
        loss_fn = M.Losses['edm']()
        opt = M.FusedAdam(net.parameters(), lr=1e-3)

        def run(sl, model):
            md = M.get_mask(sl.stop - sl.start, 256, 0.5, dev, noise=mnoise[sl].to(dev))
            return loss_fn.with_draws(model, images[sl].to(dev), labels[sl].to(dev), rnd[sl].to(dev), noise[sl].to(dev), md, 0.1)

        half = slice(rank * B // 2, (rank + 1) * B // 2)
        # accumulation micro-step without sync, then the synchronising one
        opt.zero_grad(set_to_none=True)
        with dp.no_sync():
            (run(half, dp).mean() * 0.0).backward()  # contributes zeros: exercises no_sync + accumulate
        run(half, dp).mean().backward()
        dp.finish_grad_sync()
        g_dp = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
        if rank == 0:  # (a) the averaged gradient IS the full-batch gradient of the fp32 oracle
            mdict = {k: torch.from_numpy(v) for k, v in O.get_mask_from_noise(mnoise.numpy(), 0.5).items()}
            _, _, g_ref = O.loss_and_grads(P0, cfg, images, labels, rnd, noise, mdict, 0.1)
            for k, gr in g_ref.items():
                num = (g_dp[k].cpu().double() - gr.double()).norm().item()
                den = gr.double().norm().item()
                assert num <= 1e-2 * den + 1e-7, f'{k}: DP gradient differs from the oracle ({num / (den + 1e-12):.3e})'
        # single-process reference on the full batch (same weights), no DP hook
        net.engine().grad_slab_hook = None
        opt.zero_grad(set_to_none=True)
        run(slice(0, B), net).mean().backward()
        worst = 0.0
        for k, p in net.named_parameters():
            if p.requires_grad:
                num = (p.grad - g_dp[k]).norm().item()
                den = p.grad.norm().item()
                worst = max(worst, num / (den + 1e-12))
                assert num <= 5e-3 * den + 1e-7, f'{k}: DP gradient differs from the full-batch gradient ({num / (den + 1e-12):.3e})'
        # restore the averaged grads, step, and compare replicas bit for bit
        for k, p in net.named_parameters():
            if p.requires_grad:
                p.grad.copy_(g_dp[k])
        opt.step()
        flat = net.engine().P.detach().clone()
        other = flat.clone()
        dist.broadcast(other, src=0)
        assert torch.equal(flat, other), 'replicas diverged after the optimizer step'
        q.put((rank, 'ok', worst))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL: ' + traceback.format_exc(), 0.0))
    finally:
        dist.destroy_process_group()


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _backends():
    return ['gloo'] + (['nccl'] if torch.cuda.is_available() and torch.cuda.device_count() >= 2 else [])


@pytest.mark.timeout(600)
@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
def test_two_rank_data_parallel(backend):
    if backend not in _backends():
        pytest.skip('the nccl (RCCL) variant needs >= 2 visible GPUs')
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] == 'ok' for r in res), res
    print('worst DP-vs-full-batch grad rel err', max(r[2] for r in res))


def _rccl_forms_worker(port, q):
    """ONE rank, backend nccl (= RCCL): the only RCCL execution a 1-GPU box allows (two ranks on one device are refused:
    "Duplicate GPU detected").  It cannot show link behaviour, but it does run the exact CALL FORMS of the data-parallel path
    through RCCL's argument checks and kernels: asynchronous AVG all-reduce on an arena view, in-place reduce_scatter_tensor
    whose output is a view of its input, the bf16 staging wire, in-place all_gather_into_tensor into the arena."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    try:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        from maskdit_amd.ddp import GradSlabReducer, slab_pieces
        flat = torch.randn(1 << 16, device='cuda')
        ref = flat.clone()
        r = GradSlabReducer()
        assert r.backend == 'nccl' and r.world == 1
        r.attach(flat)
        r._launch(128, 40000)                                   # all_reduce(AVG), async
        qq, pieces, tail = slab_pieces(40000, 65000, 1)
        r._launch(40000, 40000 + qq, scatter_q=qq)             # in-place reduce_scatter_tensor(AVG)
        r.finish()
        torch.cuda.synchronize()
        assert r.forms_agreed and r.use_avg and r.use_scatter and not r.plain_only, 'RCCL refused a call form at attach()'
        assert torch.equal(flat, ref) and r.wire_bytes == (40000 - 128 + qq) * 4
        rb = GradSlabReducer(wire_dtype=torch.bfloat16)         # the staging arena is only allocated for world > 1: give it one
        rb.attach(flat)
        rb.stage = torch.empty(flat.numel(), device='cuda', dtype=torch.bfloat16)
        rb._launch(0, 4096)
        rb.finish()
        torch.cuda.synchronize()
        assert torch.equal(flat[:4096], ref[:4096].to(torch.bfloat16).float()) and torch.equal(flat[4096:], ref[4096:])
        w = dist.all_gather_into_tensor(flat[8192:8192 + 1024], flat[8192:8192 + 1024], async_op=True)  # zero.py's parameter gather form
        w.wait()
        torch.cuda.synchronize()
        assert torch.equal(flat[8192:9216], ref[8192:9216])
        q.put('ok NCCL ' + '.'.join(map(str, torch.cuda.nccl.version())))
    except Exception:  # noqa: BLE001
        import traceback
        q.put('FAIL: ' + traceback.format_exc())
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_call_forms_on_a_single_rank_group():
    """No multi-GPU node has been available in rounds 1-5, so RCCL had never executed a line of this repo's exchange.  A
    world-size-1 nccl group on the one GPU runs every collective FORM the path uses (see the worker) -- an argument-level
    refusal (ReduceOp.AVG, in-place views, bf16) would show here instead of on the first 8-GPU run.  What a single rank
    CANNOT show: with world = 1 reduce-scatter / all-gather are near no-ops, so the in-place ALIASING semantics for world > 1
    (output = a view of the input) remain unverified on RCCL; the gloo tests cover the ownership arithmetic only."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_forms_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(60)
    assert res.startswith('ok'), res
    print('single-rank RCCL group:', res)


def _zero_worker(rank, world, port, q):
    """ZeRO-1 (maskdit_amd/zero.py): reduce-scattered gradient slabs (rank r owns piece r of every slab) + sharded
    AdamW/EMA + parameter all-gather must give the unsharded result."""
    sys.path.insert(0, ROOT)
    import copy
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import maskdit_amd as M
        from oracle import maskdit_oracle as O
        dev = 'cuda:0'
        cfg = O.make_cfg('DiT-S/2', img_resolution=32)
        P0 = O.init_params(cfg, seed=0, dezero=True)

        def build():
            n = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2',
                                        use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False).to(dev)
            n.load_state_dict(P0)
            return n.train()

        B = 16
        g = torch.Generator().manual_seed(5)
        images = 0.5 * torch.randn(2, B, 4, 32, 32, generator=g)
        labels = torch.zeros(2, B, 1000)
        for s in range(2):
            labels[s, torch.arange(B), torch.randint(0, 1000, (B,), generator=g)] = 1
        rnd, noise = torch.randn(2, B, 1, 1, 1, generator=g), torch.randn(2, B, 4, 32, 32, generator=g)
        mnoise = torch.rand(2, B, 256, generator=g)
        loss_fn = M.Losses['edm']()

        def run(model, s, sl):
            md = M.get_mask(sl.stop - sl.start, 256, 0.5, dev, noise=mnoise[s, sl].to(dev))
            return loss_fn.with_draws(model, images[s, sl].to(dev), labels[s, sl].to(dev), rnd[s, sl].to(dev), noise[s, sl].to(dev), md, 0.1)

        # ---- sharded run next to an unsharded reference that is fed the SAME averaged gradient: plain FusedAdam on a
        # second replica whose gradient arena is filled from the owners' reduced ranges.  Same kernel per element =>
        # bit-identical parameters, EMA and moments.  (Recomputing the gradient from the full batch instead would
        # differ by fp32-atomics noise, which Adam's m / (sqrt(v) + eps) amplifies to O(lr) where |g| is tiny.)
        net = build()
        ema = copy.deepcopy(net).eval()
        dp = M.DataParallel(net)
        opt = M.ShardedFusedAdam(net.parameters(), data_parallel=dp, lr=1e-3)
        assert opt._m.numel() <= net.engine().lay.n // world + 8 * world * len(net.engine().lay.slabs), 'moments are not sharded'
        opt.fuse_ema(ema, 0.99)
        ref = build()
        ref_ema = copy.deepcopy(ref).eval()
        ropt = M.FusedAdam(ref.parameters(), lr=1e-3)
        ropt.fuse_ema(ref_ema, 0.99)
        half = slice(rank * B // 2, (rank + 1) * B // 2)
        slabs = net.engine().lay.slabs
        for s in range(2):
            opt.zero_grad(set_to_none=True)
            run(dp, s, half).mean().backward()
            dp.finish_grad_sync()
            # the averaged gradient, assembled from the owners (each rank holds the reduced values of its own range only)
            G = net.engine().G
            Gfull = G.detach().clone()
            opt._gather(Gfull, slabs)  # every owner's reduced pieces (piece r of every slab belongs to rank r)
            for lo, hi in opt._pieces:
                assert torch.equal(Gfull[lo:hi], G[lo:hi])
            if s == 0:  # sanity of the reduce-to-owner gradient itself: the full-batch gradient within atomics noise
                ref2 = build()
                run(ref2, 0, slice(0, B)).mean().backward()
                gf = ref2.engine().G
                relg = ((Gfull - gf).norm() / gf.norm()).item()
                assert relg <= 5e-3, f'reduce-to-owner gradient differs from the full-batch gradient ({relg:.3e})'
                del ref2, gf
            ropt.zero_grad(set_to_none=True)
            ref._prepare_grad_arena()
            ref.engine().G.copy_(Gfull)
            opt.step()
            M.update_ema(ema, net, 0.99)
            ropt.step()
            M.update_ema(ref_ema, ref, 0.99)
        try:
            opt.state_dict()
            raise AssertionError('state_dict() on a sharded group without consolidate() must raise, not dead-lock')
        except RuntimeError:
            pass
        opt.consolidate()  # collective: moments gathered, EMA arena current everywhere
        sd = opt.state_dict() if rank == 0 else None  # the reference's rank-0-only save (train.py:259-264)
        sd = opt.state_dict()
        rsd = ropt.state_dict()
        worst = relg
        for (k, a), (_, b) in zip(net.named_parameters(), ref.named_parameters()):
            assert torch.equal(a, b), f'{k}: sharded step differs from the unsharded one ({(a - b).abs().max().item():.3e})'
        for (k, a), (_, b) in zip(ema.named_parameters(), ref_ema.named_parameters()):
            assert torch.equal(a, b), f'EMA {k} differs after sync_ema'
        assert sd['param_groups'][0]['step'] == rsd['param_groups'][0]['step'] == 2
        for i in rsd['state']:
            assert torch.equal(sd['state'][i]['exp_avg'], rsd['state'][i]['exp_avg']), f'moment {i}'
            assert torch.equal(sd['state'][i]['exp_avg_sq'], rsd['state'][i]['exp_avg_sq']), f'moment {i}'
        for arena in (net.engine().P, ema.engine().P):
            other = arena.detach().clone()
            dist.broadcast(other, src=0)
            assert torch.equal(arena, other), 'replicas diverged'
        # a sharded checkpoint loads back into the sharded optimizer (and keeps the apex layout)
        opt.load_state_dict(sd)
        opt.consolidate()
        sd2 = opt.state_dict()
        for i in sd['state']:
            assert torch.equal(sd['state'][i]['exp_avg_sq'], sd2['state'][i]['exp_avg_sq'])
        q.put((rank, 'ok', worst))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, 'FAIL: ' + traceback.format_exc(), 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_zero1_matches_unsharded():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
    for r in res:
        if r[1] != 'ok':
            print(r[1])
    assert all(r[1] == 'ok' for r in res), [r[1][-600:] for r in res]
    print('ZeRO-1: reduce-to-owner gradient vs full-batch gradient, rel L2', max(r[2] for r in res))
