#!/usr/bin/env python
"""generate.py -- the reference's sampling entry point (generate.py:18-91, sample.py:230-296) on
the MI355X engine: per-seed StackedRandomGenerator latents / labels (utils.py:119-133), the
hipGraph-captured EDM Heun sampler with classifier-free guidance, latents written as .npy.

    python generate.py --config configs/xl2-256-synthetic.yaml --seeds 0-63 --num_steps 50 --cfg_scale 1.5 \
        [--ckpt_path 2000000.pt] [--outdir samples] [--precision fp32]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 generate.py ...   # seeds sharded by rank

Seeds are split over the ranks exactly as sample.py:233-235 does (no exchange step: replicas only).
`--pretrained_path autoencoder_kl.pth` adds the reference's decode step (sample.py:248,273-296): the latents go through
maskdit_amd.autoencoder (HIP) and are written as uint8 images (`.png` when PIL is importable, else `.npy`); without it the
fp64 latents `z` are the product (the published VAE weights are not available offline)."""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

import maskdit_amd as M
from maskdit_amd.schedule import load_config


def parse_seeds(s):
    out = []
    for part in s.split(','):
        if '-' in part:
            a, b = part.split('-')
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def load_weights(net, path, key='ema'):
    """generate.py:44-49: the `ema` weights of a training checkpoint; keys written from a torch.compile'd module carry
    an `_orig_mod.` prefix."""
    ck = torch.load(path, map_location='cpu', weights_only=False)
    sd = ck[key] if key in ck else ck
    net.load_state_dict({k.replace('_orig_mod.', ''): v for k, v in sd.items()})


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', required=True)
    ap.add_argument('--ckpt_path', default=None)
    ap.add_argument('--outdir', default='samples')
    ap.add_argument('--seeds', type=parse_seeds, default=list(range(64)))
    ap.add_argument('--max_batch_size', type=int, default=64)
    ap.add_argument('--num_steps', type=int, default=50)
    ap.add_argument('--cfg_scale', type=float, default=None)
    ap.add_argument('--class_idx', type=int, default=None)
    ap.add_argument('--subdirs', action='store_true', help='one sub-directory per 1000 seeds (sample.py:288)')
    ap.add_argument('--pretrained_path', default=None, help='autoencoder_kl.pth: decode the latents to images (sample.py:248)')
    ap.add_argument('--precision', choices=['bf16', 'fp32'], default='bf16',
                    help="arithmetic of the network evaluations: 'fp32' = exact fp32 weights / activations / matrix instructions, what "
                         "the reference's own sampler runs (sample.py:56; ~1/9 of the bf16 throughput); 'bf16' = the training kernels")
    args = ap.parse_args(argv)
    cfg = load_config(args.config)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    mc = cfg.model
    net = M.Precond_models[mc.precond](img_resolution=mc.in_size, img_channels=mc.in_channels, num_classes=mc.num_classes,
                                       model_type=mc.model_type, use_decoder=mc.use_decoder, mae_loss_coef=mc.mae_loss_coef,
                                       pad_cls_token=mc.pad_cls_token).to(dev).eval()
    if args.ckpt_path:
        load_weights(net, args.ckpt_path)
    vae = None
    if args.pretrained_path:
        from maskdit_amd import autoencoder
        vae = autoencoder.get_model(args.pretrained_path).to(dev)
    os.makedirs(args.outdir, exist_ok=True)
    t0, n = time.time(), 0
    for seeds in M.seed_batches(args.seeds, args.max_batch_size, rank, world):
        if not seeds:
            continue
        rnd = M.StackedRandomGenerator(dev, seeds)
        latents = rnd.randn([len(seeds), net.img_channels, net.img_resolution, net.img_resolution], device=dev)
        labels = torch.eye(net.num_classes, device=dev)[rnd.randint(net.num_classes, size=[len(seeds)], device=dev)]
        if args.class_idx is not None:
            labels[:, :] = 0
            labels[:, args.class_idx] = 1
        z = M.edm_sampler(net, latents, labels, cfg_scale=args.cfg_scale, randn_like=rnd.randn_like, num_steps=args.num_steps,
                          precision=args.precision)
        images = None
        if vae is not None:  # sample.py:282-286: decode, [-1, 1] -> uint8 HWC
            images = vae.decode(z.float()).add_(1).mul_(127.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        for j, (s, zi) in enumerate(zip(seeds, z.cpu().numpy())):
            d = os.path.join(args.outdir, f'{s - s % 1000:06d}') if args.subdirs else args.outdir
            os.makedirs(d, exist_ok=True)
            if images is None:
                np.save(os.path.join(d, f'{s:06d}.npy'), zi)
                continue
            try:
                import PIL.Image
                PIL.Image.fromarray(images[j], 'RGB').save(os.path.join(d, f'{s:06d}.png'))
            except ImportError:
                np.save(os.path.join(d, f'{s:06d}.npy'), images[j])
        n += len(seeds)
    torch.cuda.synchronize()
    print(f'[rank {rank}/{world}] {n} latents, {args.num_steps} steps, cfg={args.cfg_scale}: {n / (time.time() - t0):.2f} samples/s', flush=True)
    return n


if __name__ == '__main__':
    main()
