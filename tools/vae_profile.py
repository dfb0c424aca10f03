"""VAE decode timing / profiling (run on the GPU box):  python tools/vae_profile.py [batch] [decodes]
Random-init decoder of the reference architecture (as bench.py's sampler leg), 32x32x4 latents -> 256x256x3 images.
MDT_VAE_FUSE=0 selects the round-3 form (separate skip-connection add and GroupNorm statistics passes) for A/B runs;
under `rocprofv3 --kernel-trace` the per-kernel table of profiles/r*_kernel_stats_vae.txt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskdit_amd import autoencoder  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    vae = autoencoder.get_model(None)
    with torch.no_grad():
        for _, p in vae.named_weights():
            if p.dim() == 4:
                p.normal_(std=(1.6 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
            elif p.dim() == 1:
                p.normal_(std=0.1).add_(1.0 if p.shape[0] >= 128 else 0.0)
    vae = vae.to(dev)
    z = torch.randn(B, 4, 32, 32, device=dev) * 0.5
    img = vae.decode(z)  # warm-up: packs the weights, sizes the workspace
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        img = vae.decode(z)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'VAE decode batch {B} (fused epilogue: {autoencoder.FUSE_EPILOGUE}): {ms:.2f} ms = {B / ms * 1e3:.0f} img/s, finite {bool(torch.isfinite(img).all())}, '
          f'checksum {img.double().abs().mean().item():.6f}')


if __name__ == '__main__':
    main()
