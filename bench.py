#!/usr/bin/env python
"""bench.py -- MaskDiT-XL/2 training-step throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N = 1: plain process
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   # N > 1 (driver)

One "step" = one optimizer step of the reference's training loop body (train.py:200-230) on
the workload BASELINE.json quotes the metric on (configs[1]): MaskDiT-XL/2, ImageNet-256
latents [B,4,32,32], mask_ratio 0.5, mae_loss_coef 0.1, class-dropout 0.1, bf16 GEMM/attention
compute with fp32 master weights, GLOBAL batch 1024 -- as `accum` micro-batches per GPU --
forward + backward + DP gradient all-reduce + fused AdamW + EMA.  Synthetic latents / labels
are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

The timed step is train.py:200-230 in full: `sample(moments)` -> class-dropout -> loss forward/backward ->
(DP gradient mean) -> AdamW -> EMA; the latent MOMENTS [B,8,R,R] and one-hot labels are resident in HBM.

Extra objects in the line:
  roofline     : the dominant kernel (gemm_nt8_kernel, bf16 MFMA): algorithmic FLOPs of its
                 launches / their summed duration, both measured live with HIP events recorded
                 on the launch stream around every gemm_nt launch of the timed steps.
                 roofline.encoder = the north-star quantity: XL/2 ENCODER forward + backward (patch-embed ..
                 28 blocks .. and their backward, incl. the conditioning path) timed with HIP events at the plan
                 positions engine.PassPlan.marks names; achieved = 350.1 GFLOP/sample / that time.
                 roofline.traffic = HBM bytes per gemm_nt8 launch from a rocprofv3 --pmc pass of THIS command at
                 THIS micro-batch (profiles/pmc_gemm_nt.json records the command; null when it does not match).
  cpu_baseline : the CPU oracle (oracle/maskdit_oracle.py, a restatement of the reference
                 path pinned to reference-generated fixtures) timed on this host's cores on a
                 bounded sample (XL/2, batch 16, fwd+bwd+AdamW+EMA: >= 3 warm-up + 3 timed steps; thread count
                 chosen among 32/16/64 by the warm-up steps and stated; more threads are slower on the GPU box) + the 50-step sampler (batch 4,
                 6 steps timed and scaled by 99/11 network evaluations), rank 0, N = 1 only.

`python bench.py --gpus N` WITHOUT torch.distributed.run re-executes itself under it (one rank per GPU).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--model', default='DiT-XL/2')
    ap.add_argument('--resolution', type=int, default=32, help='latent resolution (32 = ImageNet-256, 64 = ImageNet-512)')
    ap.add_argument('--global-batch', type=int, default=1024)
    ap.add_argument('--micro-batch', type=int, default=0,
                    help='samples per forward/backward pass per GPU (0 = the largest power-of-two split of the per-GPU '
                         'batch whose saved activations fit in free HBM: 1024 in one pass on a 288 GB MI355X)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip the per-launch HIP events (roofline = null)')
    ap.add_argument('--cpu-batch', type=int, default=16)
    ap.add_argument('--no-sampler', action='store_true', help='skip the EDM sampler leg (BASELINE configs[4])')
    ap.add_argument('--no-sampler-fp32', action='store_true', help='skip the fp32-faithful repetition of the sampler leg (~30 s)')
    ap.add_argument('--sampler-batch', type=int, default=64)
    ap.add_argument('--sampler-steps', type=int, default=50)
    ap.add_argument('--zero1', action='store_true', help='N > 1: ZeRO-1 (maskdit_amd.ShardedFusedAdam: reduce-scattered gradient '
                                                          'slabs, sharded AdamW + EMA, parameter all-gather)')
    ap.add_argument('--grad-wire', default='fp32', choices=['fp32', 'bf16'], help='N > 1: dtype of the gradient slabs on the links')
    return ap.parse_args()


class GemmTimer:
    """HIP events around every gemm_nt launch of a plan (recorded on the launch stream)."""

    def __init__(self, lib):
        self.lib = lib
        self.pool = []
        self.used = 0
        self.records = []  # (start, stop, flops)

    def _ev(self):
        if self.used == len(self.pool):
            e = C.c_void_p()
            self.lib.mdt_event_create(C.byref(e))
            self.pool.append(e)
        e = self.pool[self.used]
        self.used += 1
        return e

    def wrap(self, plan, span=None):
        """Replace plan.run by an instrumented replay.  span = (first launch index, end index, label): one more
        event pair around that range of the plan (the encoder)."""
        from maskdit_amd import _lib
        timer = self
        calls = plan.calls
        if not hasattr(self, 'spans'):
            self.spans = []

        def run(stream):
            for i, (fn, args, name) in enumerate(calls):
                if span is not None and i == span[0]:
                    s0 = timer._ev()
                    timer.lib.mdt_event_record(s0, stream)
                if span is not None and i == span[1]:
                    e0 = timer._ev()
                    timer.lib.mdt_event_record(e0, stream)
                    timer.spans.append((s0, e0, span[2]))
                if fn is None:
                    args()
                    continue
                if name == 'mdt_gemm_nt':
                    a = args[0]._obj
                    s, e = timer._ev(), timer._ev()
                    timer.lib.mdt_event_record(s, stream)
                    rc = fn(*args, stream)
                    timer.lib.mdt_event_record(e, stream)
                    timer.records.append((s, e, 2.0 * a.M * a.N * a.K))
                else:
                    rc = fn(*args, stream)
                if rc != 0:
                    raise _lib.MaskDiTLibError(f'{name} failed ({rc}): {_lib.lib().mdt_last_error().decode()}')
            if span is not None and span[1] >= len(calls):
                e0 = timer._ev()
                timer.lib.mdt_event_record(e0, stream)
                timer.spans.append((s0, e0, span[2]))

        plan.run = run

    def span_ms(self):
        ms = C.c_float()
        out = {}
        for s, e, label in getattr(self, 'spans', []):
            self.lib.mdt_event_elapsed_ms(s, e, C.byref(ms))
            out[label] = out.get(label, 0.0) + ms.value
        return out

    def summarise(self):
        ms = C.c_float()
        tot_ms, tot_fl = 0.0, 0.0
        for s, e, fl in self.records:
            self.lib.mdt_event_elapsed_ms(s, e, C.byref(ms))
            tot_ms += ms.value
            tot_fl += fl
        n = len(self.records)
        return n, tot_ms, tot_fl


_CPU_WORKER = r"""
import json, sys, time
sys.path.insert(0, sys.argv[1])
batch, model, R, out = int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
cands = [int(c) for c in sys.argv[6].split(',')]
import torch as T
from oracle import maskdit_oracle as O
cfg = O.make_cfg(model, img_resolution=R)
P = O.init_params(cfg, seed=0, dezero=True)
names = [k for k in P if k not in O.NON_TRAINABLE]
Mm = {k: T.zeros_like(P[k]) for k in names}
V = {k: T.zeros_like(P[k]) for k in names}
EMA = {k: P[k].clone() for k in names}
g = T.Generator().manual_seed(0)
Tk = (R // cfg['patch']) ** 2
res = {'warm': [], 'timed': [], 'threads': None, 'sampler': None}
def one(it):
    images = 0.5 * T.randn(batch, 4, R, R, generator=g)
    labels = T.zeros(batch, 1000)
    labels[T.arange(batch), T.randint(0, 1000, (batch,), generator=g)] = 1
    labels *= (T.rand(batch, 1, generator=g) >= 0.1).float()
    rnd, noise = T.randn(batch, 1, 1, 1, generator=g), T.randn(batch, 4, R, R, generator=g)
    mnoise = T.rand(batch, Tk, generator=g)
    t0 = time.perf_counter()
    O.train_step(P, Mm, V, EMA, cfg, images, labels, rnd, noise, mnoise, 0.5, 0.1, step=it + 1)
    return time.perf_counter() - t0
it = 0
T.set_num_threads(cands[0]); one(it); it += 1          # cold step (page-in, allocator): not recorded
for c in cands:                                       # warm-up steps double as the thread-count scan
    T.set_num_threads(c)
    res['warm'].append((c, one(it))); it += 1
    json.dump(res, open(out, 'w'))
while len(res['warm']) < 3:
    res['warm'].append((cands[0], one(it))); it += 1
best = min(res['warm'], key=lambda cw: cw[1])[0]
T.set_num_threads(best)
res['threads'] = best
for _ in range(3):
    res['timed'].append(one(it)); it += 1
    json.dump(res, open(out, 'w'))
# sampler leg (BASELINE configs[4] on the CPU): batch 4, cfg 1.5, 6 Heun steps = 11 network evaluations
sb, ns = 4, 6
lat = T.randn(sb, 4, R, R, generator=g)
lab = T.eye(1000)[T.randint(0, 1000, (sb,), generator=g)]
with T.no_grad():
    t0 = time.perf_counter()
    O.edm_sampler(P, cfg, lat, lab, cfg_scale=1.5, num_steps=ns)
    res['sampler'] = {'seconds': time.perf_counter() - t0, 'batch': sb, 'steps': ns, 'evals': 2 * ns - 1}
json.dump(res, open(out, 'w'))
"""


def cpu_baseline(batch, model, R, budget_s=240.0):
    """Bounded CPU leg: the oracle's training step (fwd + bwd + AdamW + EMA;
    oracle/maskdit_oracle.py) in a child process with a wall-clock budget, so that a slow / oversubscribed host can
    never stall the benchmark: 1 cold step, >= 3 warm-up steps that also pick the thread count (a 16-sample fp32
    batch does not scale to the 256 hardware threads of the GPU box -- measured there: 32 threads 7.1 s/step, 64 threads
    11.1 s, 128 threads 22.7 s -- so the candidates 32 / 16 / 64 are each timed once and the fastest is used and reported), 3 timed steps, then 6 sampler steps."""
    import subprocess
    import tempfile
    cores = os.cpu_count() or 1
    cands = sorted({min(cores, c) for c in (32, 16, 64)}, key=lambda c: (c != min(cores, 32), c))  # 32 first (cold step)
    out = tempfile.NamedTemporaryFile(prefix='mdt_cpu_', suffix='.json', delete=False).name
    env = dict(os.environ, HIP_VISIBLE_DEVICES='')
    env.pop('OMP_NUM_THREADS', None)
    proc = subprocess.Popen([sys.executable, '-c', _CPU_WORKER, ROOT, str(batch), model, str(R), out, ','.join(map(str, cands))],
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env)
    try:
        proc.wait(timeout=budget_s)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
    try:
        res = json.load(open(out))
    except Exception:
        res = {'warm': [], 'timed': [], 'threads': None, 'sampler': None}
    finally:
        try:
            os.unlink(out)
        except OSError:
            pass
    scan = ', '.join(f'{c} thr {t:.2f} s' for c, t in res['warm'])
    if res['timed']:
        best = min(res['timed'])
        threads = res['threads']
        note = f'1 cold + {len(res["warm"])} warm-up (thread scan: {scan}) + {len(res["timed"])} timed steps, best {best:.2f} s/step'
    elif res['warm']:
        threads, best = min(res['warm'], key=lambda cw: cw[1])
        note = f'only {len(res["warm"])} warm-up steps fit the {budget_s:.0f} s budget ({scan})'
    else:
        return {'value': None, 'unit': 'img/s', 'cores': cands[0], 'kind': 'port',
                'sample': f'{model} latent {R}x{R}, batch {batch}: no step finished within {budget_s:.0f} s'}
    line = {'value': round(batch / best, 3), 'unit': 'img/s', 'cores': threads, 'host_cores': cores, 'kind': 'port',
            'sample': f'{model} latent {R}x{R}, batch {batch}, mask 0.5, fp32 CPU oracle train step (fwd+bwd+AdamW+EMA), {note}'}
    sm = res.get('sampler')
    if sm:
        full = sm['seconds'] * 99.0 / sm['evals']
        line['sampler'] = {'value': round(sm['batch'] / full, 5), 'unit': 'samples/s', 'cores': threads,
                           'sample': f'oracle edm_sampler {model}, batch {sm["batch"]}, cfg 1.5: {sm["steps"]} Heun steps '
                                     f'({sm["evals"]} network evaluations) took {sm["seconds"]:.1f} s, scaled by 99/{sm["evals"]} '
                                     f'to the 50-step schedule'}
    return line


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` (N > 1) from a plain shell: re-execute under torch.distributed.run, one rank per
    GPU of this node (what the driver's launch line does)."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    argv = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, argv)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            respawn_under_torchrun(args.gpus)  # does not return
        args.gpus = world
    import torch.distributed as dist
    # test hook: MDT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 over gloo, so that the N > 1 code path
    # can be exercised on a single-GPU box (RCCL needs one device per rank); never set by the driver
    one_dev = os.environ.get('MDT_BENCH_ONE_DEVICE') == '1'
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if one_dev:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)

    import maskdit_amd as M
    from maskdit_amd import _lib
    if not os.path.exists(_lib.LIB_PATH) and local_rank == 0:
        _lib.build()  # normally prebuilt by __graft_entry__.build(); compiling is not a fallback path
    if world > 1:
        dist.barrier()
    lib = _lib.lib()

    R = args.resolution
    per_gpu = args.global_batch // world
    assert per_gpu * world == args.global_batch, 'global batch must divide by the number of GPUs'
    mb = min(args.micro_batch, per_gpu) if args.micro_batch > 0 else per_gpu
    accum = per_gpu // mb
    assert accum * mb == per_gpu

    torch.manual_seed(0)  # train.py:67-68: same seed on every rank
    net = M.Precond_models['edm'](img_resolution=R, img_channels=4, num_classes=1000, model_type=args.model,
                                  use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False)
    # the reference zero-initialises every adaLN / output projection; re-draw them so that no GEMM
    # sees an all-zero operand (zero operands clock ~20 % higher and would flatter the number)
    with torch.no_grad():
        for p in net.parameters():
            if p.requires_grad and float(p.abs().max()) == 0.0:
                p.normal_(std=0.02)
    net = net.to(dev).train()
    import copy
    ema = copy.deepcopy(net).eval()
    for p in ema.parameters():
        p.requires_grad_(False)
    model = M.DataParallel(net, grad_wire_dtype=torch.bfloat16 if args.grad_wire == 'bf16' else None) if world > 1 else net
    if args.zero1 and world > 1:
        opt = M.ShardedFusedAdam(net.parameters(), data_parallel=model, lr=1e-4, adam_w_mode=True, weight_decay=0)
    else:
        opt = M.FusedAdam(net.parameters(), lr=1e-4, adam_w_mode=True, weight_decay=0)
    opt.fuse_ema(ema, 0.9999)
    loss_fn = M.Losses['edm']()
    if args.micro_batch <= 0:
        # saved activations of one training pass: bf16/fp32 tensors listed in DESIGN.md section 2
        sp = net.spec
        per_sample = (sp.depth * 46080 * (sp.T // 2) * sp.D // 1152 + sp.ddepth * 20480 * sp.T) * 1.02  # measured: 244 GB in use at 1024 incl. 30 GB of arenas
        free = torch.cuda.mem_get_info(dev)[0]
        while mb > 64 and per_sample * mb + 20e9 > free:
            mb //= 2
        accum = per_gpu // mb

    # synthetic data of the dataset's shape, resident in HBM: latent MOMENTS [B, 8, R, R] whose sample()
    # has std = sigma_data (mean 2.745 * randn scaled by 0.18215, logvar -10), one-hot labels
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    mom_all = torch.cat([2.745 * torch.randn(per_gpu, 4, R, R, device=dev, generator=gen),
                         torch.full((per_gpu, 4, R, R), -10.0, device=dev)], 1)
    cls = torch.randint(0, 1000, (per_gpu,), device=dev, generator=gen)
    y_clean = torch.zeros(per_gpu, 1000, device=dev)
    y_clean[torch.arange(per_gpu, device=dev), cls] = 1
    y_all = torch.empty_like(y_clean)
    loss_acc = torch.zeros((), device=dev)

    def step():
        x_all = M.sample(mom_all)                 # train.py:203
        opt.zero_grad(set_to_none=True)           # train.py:206
        y_all.copy_(y_clean)                      # (the loader hands over fresh labels every step)
        M.class_dropout_(y_all, 0.1)              # train.py:208-209
        for a in range(accum):
            xs, ys = x_all[a * mb:(a + 1) * mb], y_all[a * mb:(a + 1) * mb]
            last = a == accum - 1
            if world > 1 and not last:
                with model.no_sync():
                    loss = loss_fn(model, xs, ys, mask_ratio=0.5, mae_loss_coef=0.1)
                    (loss.mean() / accum).backward()
            else:
                loss = loss_fn(model, xs, ys, mask_ratio=0.5, mae_loss_coef=0.1)
                (loss.mean() / accum).backward()
            loss_acc.add_(loss.detach().mean() / accum)
        if world > 1:
            model.finish_grad_sync()
        opt.step()
        M.update_ema(ema, net, 0.9999)  # folded into opt.step() (fuse_ema): no extra pass

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    timer = None
    if not args.no_kernel_events:
        timer = GemmTimer(lib)
        pl = net.engine().plan(mb, True, True, None)
        timer.wrap(pl.fwd, span=(0, pl.marks['enc_fwd_end'], 'enc_fwd'))
        timer.wrap(pl.bwd, span=(pl.marks['enc_bwd_begin'], len(pl.bwd.calls), 'enc_bwd'))
    loss_acc.zero_()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    mean_loss = float(loss_acc.item()) / max(args.steps, 1)
    if not (mean_loss == mean_loss and abs(mean_loss) < 1e6):
        raise SystemExit(f'bench: non-finite / absurd training loss {mean_loss}')

    roof = None
    if timer is not None:
        n, tot_ms, tot_fl = timer.summarise()
        if n and tot_ms > 0:
            ach = tot_fl / (tot_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'gemm_nt8_kernel (all mdt_gemm_nt launches)', 'achieved': round(ach, 1), 'peak': MFMA_BF16_PEAK_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                    'launches': n, 'avg_launch_us': round(tot_ms * 1e3 / n, 2),
                    'avg_flops_per_launch': round(tot_fl / n / 1e9, 3), 'flops_unit': 'GFLOP',
                    'share_of_step_time': round(tot_ms * 1e-3 / dt, 4)}
            pmc = os.path.join(ROOT, 'profiles', 'pmc_gemm_nt.json')
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    # only a PMC pass of THIS workload counts (profiles/pmc_gemm_nt.json states its command)
                    # ... and only one taken on THIS build of the kernels (source hash of maskdit_amd/csrc + include)
                    if (rec.get('model'), rec.get('resolution'), rec.get('micro_batch')) == (args.model, R, mb):
                        if rec.get('source_hash') == _lib.source_hash():
                            roof['traffic'] = rec.get('hbm_bytes_per_launch')
                            roof['traffic_source'] = rec.get('command')
                        else:
                            roof['traffic_source'] = (f"profiles/pmc_gemm_nt.json was taken on another build (source hash "
                                                      f"{rec.get('source_hash')} != {_lib.source_hash()}): not reported")
                except Exception:
                    pass
            sp_ms = timer.span_ms()
            if (args.model, R) == ('DiT-XL/2', 32) and 'enc_fwd' in sp_ms and 'enc_bwd' in sp_ms:
                enc_ms = (sp_ms['enc_fwd'] + sp_ms['enc_bwd']) / args.steps        # per optimizer step (all micro-batches)
                enc_tf = 350.1e9 * per_gpu / (enc_ms * 1e-3) / 1e12               # 350.1 GFLOP / sample fwd + bwd
                roof['encoder'] = {'what': 'XL/2 encoder fwd+bwd (patch-embed, 28 blocks, conditioning path), HIP events',
                                   'ms': round(enc_ms, 2), 'fwd_ms': round(sp_ms['enc_fwd'] / args.steps, 2),
                                   'bwd_ms': round(sp_ms['enc_bwd'] / args.steps, 2), 'achieved': round(enc_tf, 1),
                                   'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(enc_tf / MFMA_BF16_PEAK_TFLOPS, 4),
                                   'target_frac': 0.60}

    # ---- EDM sampler leg (BASELINE configs[4]): XL/2, 50 Heun steps (99 network evaluations of the
    # CFG-doubled batch), cfg_scale 1.5, batch 64, hipGraph-captured; latents only (no VAE).  N = 1 only.
    sampler = None
    if rank == 0 and world == 1 and not args.no_sampler:
        try:
            net.engine().release_plans()  # training activations (~50 GB per plan) are not needed any more
            torch.cuda.empty_cache()
            ema.eval()
            sb = args.sampler_batch
            gs = torch.Generator(device=dev).manual_seed(7)
            lat = torch.randn(sb, 4, R, R, device=dev, generator=gs)
            lab = torch.eye(1000, device=dev)[torch.randint(0, 1000, (sb,), device=dev, generator=gs)]
            M.edm_sampler(ema, lat, lab, cfg_scale=1.5, num_steps=4)  # capture + warm-up
            torch.cuda.synchronize()
            ts = time.perf_counter()
            z = M.edm_sampler(ema, lat, lab, cfg_scale=1.5, num_steps=args.sampler_steps)
            torch.cuda.synchronize()
            te = time.perf_counter() - ts
            ok = bool(torch.isfinite(z).all())
            evals = 2 * args.sampler_steps - 1
            sampler = {'metric': f'EDM samples/sec {args.model} {args.sampler_steps}-step Heun cfg=1.5 bs={sb}', 'value': round(sb / te, 3),
                       'unit': 'samples/s', 'seconds': round(te, 3), 'net_evals': evals, 'finite': ok, 'hipgraph': True,
                       'model_tflops_per_s': round(sb * evals * 2 * 251.6e9 / te / 1e12, 1) if (args.model, R) == ('DiT-XL/2', 32) else None}
            # ... and the same workload at the REFERENCE'S OWN PRECISION: sample.py:56 evaluates the network in fp32 (no
            # autocast in generate.py); precision='fp32' = fp32 weights / activations / v_mfma_f32_32x32x2_f32 (157 TF peak)
            if not args.no_sampler_fp32:
                try:
                    M.edm_sampler(ema, lat, lab, cfg_scale=1.5, num_steps=2, precision='fp32')  # capture + warm-up
                    torch.cuda.synchronize()
                    ts = time.perf_counter()
                    z32 = M.edm_sampler(ema, lat, lab, cfg_scale=1.5, num_steps=args.sampler_steps, precision='fp32')
                    torch.cuda.synchronize()
                    t32 = time.perf_counter() - ts
                    sampler['fp32_value'] = round(sb / t32, 3)
                    sampler['fp32'] = {'seconds': round(t32, 3), 'finite': bool(torch.isfinite(z32).all()), 'hipgraph': True,
                                       'arithmetic': 'exact fp32 (fp32 master weights, fp32 activations, v_mfma_f32_32x32x2_f32)',
                                       'peak_tflops': 157.3,
                                       'model_tflops_per_s': round(sb * evals * 2 * 251.6e9 / t32 / 1e12, 1) if (args.model, R) == ('DiT-XL/2', 32) else None,
                                       'frac_of_fp32_matrix_peak': round(sb * evals * 2 * 251.6e9 / t32 / 1e12 / 157.3, 4) if (args.model, R) == ('DiT-XL/2', 32) else None,
                                       'bf16_vs_fp32_rel_to_max': round(float((z - z32).abs().max() / z32.abs().max()), 6)}
                    from maskdit_amd import sampler as _smp
                    _smp.release_graphs()
                    ema.engine().release_plans()
                    torch.cuda.empty_cache()
                except Exception as e:  # noqa: BLE001
                    sampler['fp32'] = {'error': repr(e)}
            # the step that follows the sampler in generate.py (sample.py:248,273-284): VAE decode of the latents
            # (random-init decoder of the reference architecture; ~0.62 TFLOP per 256^2 image)
            try:
                from maskdit_amd import autoencoder
                vae = autoencoder.get_model(None)
                with torch.no_grad():
                    for _, p in vae.named_weights():
                        if p.dim() == 4:
                            p.normal_(std=(1.6 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
                        elif p.dim() == 1:
                            p.normal_(std=0.1).add_(1.0 if p.shape[0] >= 128 else 0.0)
                vae = vae.to(dev)
                zf = z.float()
                vae.decode(zf)  # warm-up (packs the weights, sizes the workspace)
                torch.cuda.synchronize()
                tv = time.perf_counter()
                img = vae.decode(zf)
                torch.cuda.synchronize()
                tv = time.perf_counter() - tv
                sampler['vae_decode'] = {'seconds': round(tv, 4), 'images_per_s': round(sb / tv, 1), 'finite': bool(torch.isfinite(img).all()),
                                         'samples_per_s_incl_decode': round(sb / (te + tv), 3)}
                vae.release_workspace()
            except Exception as e:  # noqa: BLE001
                sampler['vae_decode'] = {'error': repr(e)}
        except Exception as e:
            sampler = {'value': None, 'error': repr(e)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.cpu_batch, args.model, R)
        except Exception as e:  # the oracle is a checker; its absence must not void the GPU number
            cpu = {'value': None, 'error': repr(e)}

    if rank == 0:
        value = args.global_batch * args.steps / dt
        line = {
            'metric': 'training img/sec MaskDiT-XL/2 256 mask=0.5 bs=1024' if (args.model, R, args.global_batch) == ('DiT-XL/2', 32, 1024)
            else f'training img/sec {args.model} latent{R} mask=0.5 bs={args.global_batch}',
            'value': round(value, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': f'{args.model} ImageNet{R * 8}-latent [{R}x{R}x4], mask_ratio=0.5, mae_loss_coef=0.1, '
                                   f'sample(moments)+class-dropout+fwd+bwd+grad-allreduce+AdamW+EMA, random-init (de-zeroed) weights',
                       'global_batch': args.global_batch, 'per_gpu_batch': per_gpu, 'micro_batch': mb, 'accum': accum,
                       'tokens_per_sample': (R // 2) ** 2, 'kept_tokens': (R // 2) ** 2 // 2,
                       'parallelism': f'dp{world}' + ('+zero1' if args.zero1 and world > 1 else '') + ('+bf16grads' if args.grad_wire == 'bf16' and world > 1 else '')},
            'model_tflops_per_s': round(value * 392.7e9 / 1e12, 1) if (args.model, R) == ('DiT-XL/2', 32) else None,
            'mean_loss': round(mean_loss, 5),
            'roofline': roof, 'sampler': sampler, 'cpu_baseline': cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
