#!/usr/bin/env python
"""train.py -- the reference's training entry point (train.py:35-336) on the MI355X engine.

    python train.py --config configs/xl2-256-synthetic.yaml [--results_dir results] [--ckpt_path x.pt]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --config ...

Same YAML schema, same step body (train.py:200-230: sample -> class dropout -> micro-batches ->
loss -> backward -> lr warm-up -> optimizer step -> EMA), same resume semantics (train.py:147-162,184-188:
`--ckpt_path` loads model / ema / (strict only) optimizer, the step counter continues from the file name, the
mask-ratio schedule and the stop condition count from the resume point, and the EMA is initialised from the model
ONLY on a fresh start), same checkpoint dict ({"model","ema","opt","args"} as `{step:07d}.pt`), same throughput log
line (steps/sec after a device sync every `log_every` steps).

Data (`data.category`): `synthetic` = latent moments drawn on the device; `wds` / `lmdb` = the reference's latent
shards / LMDB through maskdit_amd.data (pinned-memory prefetch, H2D on a copy stream).  One process per GPU;
gradient averaging = maskdit_amd.DataParallel (RCCL), no accelerate / apex / omegaconf needed."""
from __future__ import annotations

import argparse
import copy
import os
import re
import sys
import time

import torch
import torch.distributed as dist

import maskdit_amd as M
from maskdit_amd.schedule import get_mask_ratio_fn, get_one_hot, load_config, lr_rampup_factor


def str2bool(v):
    return str(v).lower() in ('1', 'true', 'yes', 'y', 't')


def strip_compile_prefix(sd):
    """generate.py:46-48: checkpoints written from a torch.compile'd module carry `_orig_mod.` in their keys."""
    return {k.replace('_orig_mod.', ''): v for k, v in sd.items()}


def list_ckpts(ckpt_dir):
    """Every `<step>.pt` of a checkpoint directory, newest first (files still being written end in `.tmp` and never match)."""
    steps = []
    if os.path.isdir(ckpt_dir):
        for f in os.listdir(ckpt_dir):
            m = re.fullmatch(r'(\d+)\.pt', f)
            if m:
                steps.append((int(m.group(1)), f))
    return [os.path.join(ckpt_dir, f) for _, f in sorted(steps, reverse=True)]


def get_latest_ckpt(ckpt_dir):
    """utils.py:22-34: the highest-numbered `<step>.pt` of a checkpoint directory, or None."""
    c = list_ckpts(ckpt_dir)
    return c[0] if c else None


def save_ckpt_atomic(obj, path):
    """Write `<path>.tmp`, then rename: a job killed in the middle of a save leaves the previous checkpoint as the newest
    complete one instead of a truncated `<step>.pt` that every auto-resume would then crash on."""
    tmp = path + '.tmp'
    with open(tmp, 'wb') as f:
        torch.save(obj, f)
        f.flush()
        os.fsync(f.fileno())  # the DATA is on stable storage before the name appears (power loss, lagging network file systems)
    os.replace(tmp, path)
    try:  # ... and so is the directory entry
        fd = os.open(os.path.dirname(os.path.abspath(path)), os.O_RDONLY)
        try:
            os.fsync(fd)
        finally:
            os.close(fd)
    except OSError:
        pass  # (file systems that cannot fsync a directory)


def remove_stale_tmp(ckpt_dir, log=print, min_age_s=600.0):
    """`<step>.pt.tmp` files are what a killed save leaves behind: never a checkpoint, only disk space.  Only
    files that have not been written to for `min_age_s` go: during a preemption grace period, or on a lagging shared file
    system, the PREVIOUS incarnation of the job may still be inside save_ckpt_atomic, and unlinking the file it is writing
    would make its os.replace publish nothing."""
    import time
    if os.path.isdir(ckpt_dir):
        now = time.time()
        for f in os.listdir(ckpt_dir):
            if re.fullmatch(r'\d+\.pt\.tmp', f):
                full = os.path.join(ckpt_dir, f)
                try:
                    if now - os.path.getmtime(full) < min_age_s:
                        log(f'left alone (written {now - os.path.getmtime(full):.0f} s ago, possibly a save in progress): {f}')
                        continue
                    os.remove(full)
                    log(f'removed the leftover of an interrupted save: {f}')
                except OSError:
                    pass


def load_ckpt_with_retry(path, tries=5, wait_s=2.0):
    """torch.load for the ranks that did not choose the checkpoint: on a lagging shared file system the file rank 0 has
    just validated may not be complete from here yet -- retry a few times before giving up."""
    import time
    for k in range(tries):
        try:
            return torch.load(path, map_location='cpu', weights_only=False)
        except Exception:  # noqa: BLE001
            if k + 1 == tries:
                raise
            time.sleep(wait_s)


def load_newest_valid_ckpt(ckpt_dir, log=print):
    """(path, loaded dict) of the newest checkpoint that torch.load accepts; unreadable ones (a truncated file from a kill
    that predates the atomic save, a half-synced network file system) are skipped with a message.  (None, None) if none."""
    for path in list_ckpts(ckpt_dir):
        try:
            return path, torch.load(path, map_location='cpu', weights_only=False)
        except Exception as e:  # noqa: BLE001 -- any unpickling / zip error means "not a complete checkpoint"
            log(f'auto-resume: skipping unreadable checkpoint {path} ({type(e).__name__}: {e})')
    return None, None


class Logger:
    """utils.py:169-225: tee of stdout / stderr into `<experiment>/log.txt` (rank 0), flushed on every write."""

    def __init__(self, file_name, file_mode='a+'):
        self.file = open(file_name, file_mode)
        self.stdout, self.stderr = sys.stdout, sys.stderr
        sys.stdout = sys.stderr = self

    def write(self, text):
        if text:
            self.file.write(text)
            self.stdout.write(text)
            self.flush()

    def flush(self):
        self.file.flush()
        self.stdout.flush()

    def __getattr__(self, name):
        # everything else a text stream offers (isatty / fileno / encoding / errors ...: tqdm, faulthandler, warnings and
        # torch.distributed ask for them) comes from the real stdout
        return getattr(self.stdout, name)

    def close(self):
        if self.file.closed:
            return
        self.flush()
        if sys.stdout is self:
            sys.stdout = self.stdout
        if sys.stderr is self:
            sys.stderr = self.stderr
        self.file.close()


def evaluate_in_loop(args, cfg, ema, dev, rank, world, exp_dir, step):
    """train.py:274-286 (`--enable_eval`): after a checkpoint, every rank generates its share of `eval_seeds` from the
    EMA weights with the hipGraph sampler (generate_with_net, sample.py:230-296) into
    `<experiment>/fid/edm-steps<N>-ckpt<step>_cfg<s>/`.  The FID number itself needs the Inception network pickle,
    which is not available offline: when `--ref_path` points at reference statistics AND an evaluator is
    importable it is computed, otherwise the samples are left for an offline evaluator and the hook reports their
    latent statistics."""
    import numpy as np
    outdir = os.path.join(exp_dir, 'fid', f'edm-steps{args.num_steps}-ckpt{step}_cfg{args.cfg_scale}')
    os.makedirs(outdir, exist_ok=True)
    was_training = ema.training
    ema.eval()
    t0, n, stats = time.time(), 0, torch.zeros(2, device=dev, dtype=torch.float64)
    for seeds in M.seed_batches(list(range(args.eval_seeds)), args.max_batch_size, rank, world):
        if not seeds:
            continue
        rnd = M.StackedRandomGenerator(dev, seeds)
        lat = rnd.randn([len(seeds), ema.img_channels, ema.img_resolution, ema.img_resolution], device=dev)
        lab = torch.eye(ema.num_classes, device=dev)[rnd.randint(ema.num_classes, size=[len(seeds)], device=dev)]
        z = M.edm_sampler(ema, lat, lab, cfg_scale=args.cfg_scale, randn_like=rnd.randn_like, num_steps=args.num_steps)
        stats += torch.stack([z.sum(), (z * z).sum()])
        for sd, zi in zip(seeds, z.cpu().numpy()):
            np.save(os.path.join(outdir, f'{sd:06d}.npy'), zi)
        n += len(seeds)
    ema.train(was_training)
    cnt = torch.tensor([float(n)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(stats)
        dist.all_reduce(cnt)
        dist.barrier()
    numel = cnt.item() * ema.img_channels * ema.img_resolution ** 2
    mean = stats[0].item() / max(numel, 1)
    std = max(stats[1].item() / max(numel, 1) - mean * mean, 0.0) ** 0.5
    fid = None
    if args.ref_path:
        try:
            from fid import calc  # the reference's evaluator (fid.py), when the user has it + the Inception pickle
            fid = calc(outdir, args.ref_path, args.eval_seeds, args.global_seed, 64)
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f'eval: FID evaluator unavailable ({type(e).__name__}: {e}); samples kept in {outdir}', flush=True)
    if rank == 0:
        print(f'eval @ step {step}: {int(cnt.item())} samples, {args.num_steps} steps, cfg={args.cfg_scale}, latent mean {mean:+.4f} std {std:.4f}, '
              f'time {time.time() - t0:.1f} s' + (f', fid: {fid}' if fid is not None else ''), flush=True)
    return {'outdir': outdir, 'n': int(cnt.item()), 'mean': mean, 'std': std, 'fid': fid}


def parse(argv=None):
    ap = argparse.ArgumentParser('training parameters')
    ap.add_argument('--config', required=True)
    ap.add_argument('--results_dir', default='results')
    ap.add_argument('--exp_name', default='run')
    ap.add_argument('--ckpt_path', default=None)
    ap.add_argument('--use_strict_load', type=str2bool, default=True)   # train.py:309: False = finetune (weights only, non-strict)
    ap.add_argument('--global_seed', type=int, default=0)
    ap.add_argument('--max_num_steps', type=int, default=None)
    ap.add_argument('--data_path', default=None, help='override data.root (wds: shard dir / glob; lmdb: dataset dir)')
    ap.add_argument('--use_ckpt_path', type=str2bool, default=True)     # train.py:87: with --ckpt_path, keep writing into ITS experiment dir
    ap.add_argument('--auto_resume', type=str2bool, default=True)       # train.py:98-99: no --ckpt_path -> newest checkpoint of the exp dir
    # in-loop evaluation (train.py:274-286, args of train.py:318-331)
    ap.add_argument('--enable_eval', action='store_true')
    ap.add_argument('--eval_seeds', type=int, default=64, help='samples generated per evaluation (reference: seeds 0-49999)')
    ap.add_argument('--max_batch_size', type=int, default=64)
    ap.add_argument('--num_steps', type=int, default=40)
    ap.add_argument('--cfg_scale', type=float, default=None)
    ap.add_argument('--ref_path', default=None)
    # train.py:305 of the reference: --no_amp = fp32 / TF32 training.  This engine's TRAINING kernels compute in bf16 with an fp32
    # residual stream / fp32 master weights (the reference under autocast); an fp32 backward does not exist.  The flag is
    # accepted so that a reference command line fails with a statement of what IS provided, not with an argparse usage error.
    ap.add_argument('--no_amp', action='store_true', help='not provided: fp32 exists for inference only (generate.py --precision fp32)')
    args = ap.parse_args(argv)
    if args.no_amp:
        ap.error('--no_amp (fp32 training) is not provided by maskdit_amd: training runs bf16 MFMA operands with fp32 accumulation, '
                 "an fp32 residual stream and fp32 master weights; the fp32-faithful path covers inference (generate.py --precision fp32, "
                 "edm_sampler(precision='fp32'))")
    return args


def make_batches(cfg, args, dev, rank, world, B):
    """Iterator of (moments [B, 2C, R, R] f32, labels int64 [B]) ON THE DEVICE."""
    mc = cfg.model
    R, C = mc.in_size, mc.in_channels
    default = getattr(args, 'default_category', 'synthetic')
    cat = cfg.data.get('category', default) if 'data' in cfg else default
    root = args.data_path or (cfg.data.get('root') if 'data' in cfg else None)
    if cat == 'synthetic':
        gen = torch.Generator(device=dev).manual_seed(1 + rank)

        def it():
            while True:
                mom = torch.cat([2.745 * torch.randn(B, C, R, R, device=dev, generator=gen), torch.full((B, C, R, R), -10.0, device=dev)], 1)
                yield mom, torch.randint(0, mc.num_classes, (B,), device=dev, generator=gen)
        return it()
    from maskdit_amd import data as D
    if cat in ('wds', 'webdataset', 'imagenet_wds'):
        src = D.WdsTarLatents(root, B, rank=rank, world=world, seed=args.global_seed, epochs=None)
    elif cat in ('lmdb', 'imagenet_latent', 'imagenet_lmdb'):
        src = D.LmdbLatents(root, B, R, rank=rank, world=world, seed=args.global_seed, epochs=None)
    else:
        raise ValueError(f'data.category {cat!r}: expected synthetic | wds | lmdb')
    return D.LatentPrefetcher(src, dev)


def train_loop(args):
    """train.py:56-291.  The tee of stdout / stderr into log.txt is undone on EVERY exit path (an exception used to leave
    sys.stdout hijacked and log.txt open -- in pytest that leaked into the following tests)."""
    state = {'logger': None}
    try:
        return _train_loop(args, state)
    finally:
        if state['logger'] is not None:
            state['logger'].close()


def _train_loop(args, state):
    cfg = load_config(args.config)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # test hook (as bench.py): MDT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 over gloo, so that the N > 1 loop (DataParallel,
    # ZeRO-1 consolidation before rank 0 saves, resume broadcast) can be exercised on a one-GPU box; RCCL needs a device per rank
    one_dev = os.environ.get('MDT_BENCH_ONE_DEVICE') == '1'
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if one_dev:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    torch.manual_seed(args.global_seed)  # same seed on every rank, as train.py:67-68

    mc, tc = cfg.model, cfg.train
    net = M.Precond_models[mc.precond](img_resolution=mc.in_size, img_channels=mc.in_channels, num_classes=mc.num_classes,
                                       model_type=mc.model_type, use_decoder=mc.use_decoder, mae_loss_coef=mc.mae_loss_coef,
                                       pad_cls_token=mc.pad_cls_token, ext_feature_dim=mc.get('ext_feature_dim', 0)).to(dev)
    ema = copy.deepcopy(net).eval()
    for p in ema.parameters():
        p.requires_grad_(False)
    # experiment directory (train.py:85-103): with --ckpt_path (and use_ckpt_path) keep writing into that checkpoint's
    # experiment; otherwise <results_dir>/<exp_name>, resuming from its newest checkpoint if there is one
    resumed = False
    preloaded = None
    early_log = []  # messages of the resume scan: printed once the Logger tees into log.txt (they used to precede it)
    if args.ckpt_path and args.use_ckpt_path and os.path.basename(os.path.dirname(os.path.abspath(args.ckpt_path))) == 'checkpoints':
        exp_dir = os.path.dirname(os.path.dirname(os.path.abspath(args.ckpt_path)))
    else:
        exp_dir = os.path.join(args.results_dir, args.exp_name)
        if args.ckpt_path is None and args.auto_resume:
            # RANK 0 picks the checkpoint (the newest one that actually loads) and tells the others: every rank listing the
            # directory on its own can disagree on a lagging file system, and then resumes from different steps
            choice = [None]
            if rank == 0:
                remove_stale_tmp(os.path.join(exp_dir, 'checkpoints'), log=early_log.append)
                choice[0], preloaded = load_newest_valid_ckpt(os.path.join(exp_dir, 'checkpoints'), log=early_log.append)
            if world > 1:
                dist.broadcast_object_list(choice, src=0)
            args.ckpt_path = choice[0]
            resumed = args.ckpt_path is not None
    logger = None
    if rank == 0:
        os.makedirs(os.path.join(exp_dir, 'checkpoints'), exist_ok=True)
        logger = state['logger'] = Logger(os.path.join(exp_dir, 'log.txt'))
        print(f'Experiment directory created at {exp_dir}', flush=True)
        for msg in early_log:
            print(msg, flush=True)
        if resumed:
            print(f'resuming from the latest checkpoint {args.ckpt_path}', flush=True)
    # data parallelism first: the sharded optimizer (train.zero1) needs the wrapper's reducer
    wire = str(tc.get('grad_wire', 'fp32')).lower()
    model = M.DataParallel(net, grad_wire_dtype=torch.bfloat16 if wire in ('bf16', 'bfloat16') else None) if world > 1 else net
    zero1 = bool(tc.get('zero1', False)) and world > 1
    if zero1:  # optimizer state / optimizer + EMA stream sharded over the ranks (maskdit_amd/zero.py)
        opt = M.ShardedFusedAdam(net.parameters(), data_parallel=model, lr=tc.lr, adam_w_mode=True, weight_decay=0)
    else:
        opt = M.FusedAdam(net.parameters(), lr=tc.lr, adam_w_mode=True, weight_decay=0)
    step0 = 0
    if args.ckpt_path:  # train.py:147-162
        if world > 1:
            dist.barrier()  # rank 0 has read the whole file: the others start reading a file that is known to be complete
        ck = preloaded if preloaded is not None else load_ckpt_with_retry(args.ckpt_path)
        preloaded = None
        net.load_state_dict(strip_compile_prefix(ck['model']), strict=args.use_strict_load)
        ema.load_state_dict(strip_compile_prefix(ck['ema']), strict=args.use_strict_load)
        if args.use_strict_load and 'opt' in ck:
            opt.load_state_dict(ck['opt'])
        base = os.path.basename(args.ckpt_path).split('.pt')[0]
        step0 = int(base) if base.isdigit() else 0
        del ck
    else:
        M.update_ema(ema, net, decay=0)  # train.py:184-188: only a FRESH run copies the model into the EMA
    opt.fuse_ema(ema, 0.9999)
    if world > 1 and args.ckpt_path:  # the checkpoint was loaded after DataParallel's construction-time broadcast
        dist.broadcast(net.engine().P, src=0)
        dist.broadcast(ema.engine().P, src=0)
        net.engine().shadows_dirty = ema.engine().shadows_dirty = True
    net.train()
    loss_fn = M.Losses['edm']()
    mask_ratio_fn = get_mask_ratio_fn(mc.get('mask_ratio_fn', 'constant'), mc.mask_ratio, mc.get('mask_ratio_min', 0))
    accum = tc.get('grad_accum', 1)
    mb = tc.batchsize
    global_batch = mb * accum * world
    max_steps = args.max_num_steps or tc.get('max_num_steps', 100)
    if rank == 0:
        print(f'{mc.model_type} params {sum(p.numel() for p in net.parameters()):,}  global batch {global_batch} '
              f'({world} GPU x {mb} x accum {accum})  steps {step0} -> {step0 + max_steps}'
              + ('  [ZeRO-1]' if zero1 else '') + (f'  [gradient wire {wire}]' if world > 1 else ''), flush=True)

    batches = make_batches(cfg, args, dev, rank, world, mb * accum)
    step, log_steps, running = step0, 0, torch.zeros((), device=dev)
    last_loss = last_eval = None
    t0 = time.time()
    for mom, cls in batches:
        x = M.sample(mom)                                      # train.py:203
        y = get_one_hot(cls, mc.num_classes)                   # train_wds.py:266 (LMDB labels arrive as class indices too)
        opt.zero_grad(set_to_none=True)                        # train.py:206
        ratio = mask_ratio_fn((step - step0) / max_steps)      # train.py:207: progress counts from the resume point
        if mc.class_dropout_prob > 0:                          # train.py:208 (no draw at probability 0)
            M.class_dropout_(y, mc.class_dropout_prob)
        for a in range(accum):
            xs, ys = x[a * mb:(a + 1) * mb], y[a * mb:(a + 1) * mb].contiguous()
            sync = a == accum - 1
            if world > 1 and not sync:
                with model.no_sync():
                    loss = loss_fn(model, xs, ys, mask_ratio=ratio, mae_loss_coef=mc.mae_loss_coef)
                    (loss.mean() / accum).backward()
            else:
                loss = loss_fn(model, xs, ys, mask_ratio=ratio, mae_loss_coef=mc.mae_loss_coef)
                (loss.mean() / accum).backward()
            running += loss.detach().mean() / accum            # train.py:226,232: every accumulation round counts
        if world > 1:
            model.finish_grad_sync()
        for g in opt.param_groups:                             # train.py:223-225
            g['lr'] = tc.lr * lr_rampup_factor(step, global_batch, tc.get('lr_rampup_kimg', 0))
        opt.step()
        M.update_ema(ema, net)                                 # train.py:230 (folded into opt.step())
        step += 1
        log_steps += 1
        if step % cfg.log.log_every == 0:
            torch.cuda.synchronize()
            sps = log_steps / (time.time() - t0)
            avg = running / log_steps
            if world > 1:
                dist.all_reduce(avg, op=dist.ReduceOp.SUM)
                avg = avg / world
            last_loss = avg.item()
            if rank == 0:
                print(f'(step={step:07d}) Train Loss: {last_loss:.4f}, Train Steps/Sec: {sps:.2f}, '
                      f'img/s: {sps * global_batch:.1f}, mem: {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB', flush=True)
            running.zero_()
            log_steps, t0 = 0, time.time()
        if step % cfg.log.ckpt_every == 0 and step > step0:            # train.py:259-271
            if zero1:
                opt.consolidate()  # COLLECTIVE: moments + EMA gathered so that rank 0 alone can save
            if rank == 0:
                save_ckpt_atomic({'model': net.state_dict(), 'ema': ema.state_dict(), 'opt': opt.state_dict(), 'args': vars(args)},
                                 os.path.join(exp_dir, 'checkpoints', f'{step:07d}.pt'))
                print(f'Saved checkpoint to {os.path.join(exp_dir, "checkpoints", f"{step:07d}.pt")}', flush=True)
            if world > 1:
                dist.barrier()
            if args.enable_eval:                                       # train.py:274-286
                if zero1:
                    opt.sync_ema()
                last_eval = evaluate_in_loop(args, cfg, ema, dev, rank, world, exp_dir, step)
                t0, log_steps = time.time(), 0
                running.zero_()
        if step >= step0 + max_steps:                          # train.py:236: max_num_steps MORE steps
            break
    if hasattr(batches, 'close'):
        batches.close()
    if world > 1:
        dist.barrier()
    if zero1:
        opt.sync_ema()
    return {'net': net, 'ema': ema, 'opt': opt, 'step': step, 'loss': last_loss, 'exp_dir': exp_dir, 'eval': last_eval}


def main():
    train_loop(parse())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
