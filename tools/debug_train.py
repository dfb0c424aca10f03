import sys, copy, torch
sys.path.insert(0, '/root/repo')
import maskdit_amd as M
from maskdit_amd.schedule import get_one_hot
dev = 'cuda'
torch.manual_seed(0)
net = M.Precond_models['edm'](img_resolution=32, img_channels=4, num_classes=1000, model_type='DiT-S/2', use_decoder=True, mae_loss_coef=0.1, pad_cls_token=False).to(dev)
ema = copy.deepcopy(net).eval()
for p in ema.parameters():
    p.requires_grad_(False)
opt = M.FusedAdam(net.parameters(), lr=1e-3, adam_w_mode=True, weight_decay=0)
M.update_ema(ema, net, decay=0)
opt.fuse_ema(ema, 0.9999)
net.train()
loss_fn = M.Losses['edm']()
k = 'model.blocks.0.attn.qkv.weight'
P = dict(net.named_parameters())[k]
E = dict(ema.named_parameters())[k]
prev = P.detach().clone()
for step in range(6):
    mom = torch.cat([2.745 * torch.randn(32, 4, 32, 32, device=dev), torch.full((32, 4, 32, 32), -10.0, device=dev)], 1)
    x = M.sample(mom)
    y = get_one_hot(torch.randint(0, 1000, (32,), device=dev), 1000)
    opt.zero_grad(set_to_none=True)
    M.class_dropout_(y, 0.1)
    for a in range(2):
        loss = loss_fn(net, x[a * 16:(a + 1) * 16], y[a * 16:(a + 1) * 16].contiguous(), mask_ratio=0.5, mae_loss_coef=0.1)
        (loss.mean() / 2).backward()
    for g in opt.param_groups:
        g['lr'] = 1e-3 * min(step * 32 / 1e-8, 1.0)
    opt.step()
    M.update_ema(ema, net)
    torch.cuda.synchronize()
    print(f'step {step}: lr {opt.param_groups[0]["lr"]:.1e} |dP| {(P.detach() - prev).abs().max().item():.3e} |EMA-P| {(E - P).abs().max().item():.3e} '
          f'|G| {P.grad.abs().max().item():.3e} arena {opt._arena is not None} ema_applied {net.engine().ema_applied}')
    prev = P.detach().clone()
    if step == 2:
        sd = {'model': net.state_dict(), 'ema': ema.state_dict()}
        torch.save(sd, '/tmp/ck3.pt')
        ck = torch.load('/tmp/ck3.pt', map_location='cpu')
        print('   ck3: model==live', torch.equal(ck['model'][k], P.detach().cpu()), ' ema==live ema', torch.equal(ck['ema'][k], E.detach().cpu()),
              ' model==ema', torch.equal(ck['model'][k], ck['ema'][k]))
