"""One EDM sampler run of bench.py's sampler leg, for `rocprofv3 --kernel-trace -- python tools/sampler_profile.py`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maskdit_amd as M  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16'   # 'fp32' = the fp32-faithful plan (csrc/f32path.hip)
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    net = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-XL/2', use_decoder=True,
                                  mae_loss_coef=0.1, pad_cls_token=False).to(dev)
    net.eval()
    sb = 64
    lat = torch.randn(sb, 4, 32, 32, device=dev)
    lab = torch.eye(1000, device=dev)[torch.randint(0, 1000, (sb,), device=dev)]
    M.edm_sampler(net, lat, lab, cfg_scale=1.5, num_steps=4 if precision == 'bf16' else 2, precision=precision)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    z = M.edm_sampler(net, lat, lab, cfg_scale=1.5, num_steps=steps, precision=precision)
    e1.record()
    torch.cuda.synchronize()
    print(f'{precision}: {steps} steps, {2 * steps - 1} network evaluations of batch {2 * sb}: {e0.elapsed_time(e1):.1f} ms, finite {bool(torch.isfinite(z).all())}')


if __name__ == '__main__':
    main()
