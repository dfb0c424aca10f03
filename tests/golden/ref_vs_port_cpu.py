"""BUILD CONTAINER ONLY (needs /root/reference): the reference's OWN training step timed beside the oracle port that
bench.py's `cpu_baseline` leg times on the GPU box (`cpu_baseline.kind: "port"` -- the reference cannot travel there).
Same inputs, same thread count, same step: XL/2, 256^2 latents, batch 16, mask 0.5, mae 0.1 --
    reference : train_utils/loss.py EDMLoss -> models/maskdit.py -> loss.mean().backward() -> torch AdamW (== apex FusedAdam in
                adam_w_mode, wd 0) -> train_utils/helper.update_ema                (train.py:216-230)
    port      : oracle.maskdit_oracle.train_step (the function bench.py times)
and the two losses are compared, so the record also shows the port computes the reference's numbers.
    python tests/golden/ref_vs_port_cpu.py [--threads 8] [--steps 3] > profiles/r5_cpu_reference_vs_port.txt
(lives next to make_golden.py: like it, it is test infrastructure that imports the reference and the oracle; nothing under tools/ or
the product package does.)"""
import argparse
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import make_golden as G  # noqa: E402  (imports the reference through the timm / lmdb shims)
from oracle import maskdit_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    B, R, model = a.batch, 32, 'DiT-XL/2'
    cfg = O.make_cfg(model, img_resolution=R)
    P = O.init_params(cfg, seed=0, dezero=True)
    T = (R // cfg['patch']) ** 2
    g = torch.Generator().manual_seed(0)

    def draws():
        images = 0.5 * torch.randn(B, 4, R, R, generator=g)
        labels = torch.zeros(B, 1000)
        labels[torch.arange(B), torch.randint(0, 1000, (B,), generator=g)] = 1
        labels *= (torch.rand(B, 1, generator=g) >= 0.1).float()
        return images, labels, torch.randn(B, 1, 1, 1, generator=g), torch.randn(B, 4, R, R, generator=g), torch.rand(B, T, generator=g)

    batches = [draws() for _ in range(a.steps + 1)]
    # ---- the reference
    net = G.build_ref(model, R, P)
    net.train()
    wrapped = G.Wrap(net)
    ema = copy.deepcopy(net)
    opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    loss_fn = G.Losses['edm']()
    t_ref, l_ref = [], []
    for images, labels, rnd, noise, mnoise in batches:
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        with G._InjectedDraws(rnd, noise, mnoise):
            loss = loss_fn(net=wrapped, images=images, labels=labels, mask_ratio=0.5, mae_loss_coef=0.1)
        loss.mean().backward()
        opt.step()
        G.update_ema(ema, net, decay=0.9999)
        t_ref.append(time.perf_counter() - t0)
        l_ref.append(loss.detach().mean().item())
    del net, ema, opt, wrapped
    # ---- the port
    names = [k for k in P if k not in O.NON_TRAINABLE]
    Pp = {k: v.clone() for k, v in P.items()}
    Mm = {k: torch.zeros_like(Pp[k]) for k in names}
    V = {k: torch.zeros_like(Pp[k]) for k in names}
    EMA = {k: Pp[k].clone() for k in names}
    t_port, l_port = [], []
    for it, (images, labels, rnd, noise, mnoise) in enumerate(batches):
        t0 = time.perf_counter()
        out = O.train_step(Pp, Mm, V, EMA, cfg, images, labels, rnd, noise, mnoise, 0.5, 0.1, step=it + 1)
        t_port.append(time.perf_counter() - t0)
        l = out[0] if isinstance(out, (tuple, list)) else out
        l_port.append(float(torch.as_tensor(l).mean()))
    print(f'# reference vs oracle port on the build container: {model}, batch {B}, {R}x{R} latents, mask 0.5, mae 0.1, {a.threads} threads of '
          f'{os.cpu_count()} ({torch.__version__}); step 0 is cold and not averaged')
    print('step   reference s   port s   reference loss   port loss')
    for k in range(len(batches)):
        print(f'{k:4d}   {t_ref[k]:11.2f}   {t_port[k]:6.2f}   {l_ref[k]:14.6f}   {l_port[k]:9.6f}')
    r = sum(t_ref[1:]) / max(1, len(t_ref) - 1)
    q = sum(t_port[1:]) / max(1, len(t_port) - 1)
    print(f'mean of the warm steps: reference {r:.2f} s = {B / r:.2f} img/s; port {q:.2f} s = {B / q:.2f} img/s; port / reference time {q / r:.3f}')
    print(f'max |loss difference| over the steps (same weights only at step 0; both then apply their own AdamW step): '
          f'{max(abs(x - y) for x, y in zip(l_ref, l_port)):.3e}')


if __name__ == '__main__':
    main()
