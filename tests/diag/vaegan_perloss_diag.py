"""Relative L2 error of every per-loss gradient of the VAE-GAN step (HIP modules vs the oracle in fp64 on CPU).
usage: python tests/diag/vaegan_perloss_diag.py [img] [z] [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu as O, ref_vaegan as G  # noqa: E402  (checker only)
import vae_play_amd as V  # noqa: E402

S, z, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 32, 4)
x, targets, eps, z_p = G.synthetic_batch(B, S, z)
p = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in G.init_vaegan_params(S, z, seed=0).items()}
O.require_grad(p)
_, o_losses = G.train_losses(p, x.double(), targets.double(), eps.double(), z_p.double(), S)
names = O.trainable_names(p)
net = V.VaeGan(S, z)
net.load_state_dict(G.init_vaegan_params(S, z, seed=0), strict=True)
net = net.cuda().train()
xd, td, ed, zd = x.cuda(), targets.cuda(), eps.cuda(), z_p.cuda()
x_tilde, dc, dl, mus, logvar, params = net(xd, eps=ed, z_p=zd)
nle, kl, mse, bo, bp, bs, l1 = V.VaeGan.loss(xd, x_tilde, dl[:B], dl[B:-B], dl[-B:], dc[:B], dc[B:-B], dc[-B:], mus, logvar, td, params)
lam = G.LAMBDA_MSE
losses = {"loss_recon": F.mse_loss(xd, x_tilde), "loss_encoder": torch.sum(kl) + torch.sum(mse)}
losses["loss_discriminator"] = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
losses["loss_decoder"] = torch.sum(lam * mse) - (1.0 - lam) * losses["loss_discriminator"]
losses["loss_aux"] = l1
ours = dict(net.named_parameters())
for k in ("loss_recon", "loss_encoder", "loss_discriminator", "loss_aux"):
    print(f"{k}: ours {losses[k].item():.8g} oracle64 {o_losses[k].item():.8g}")
    go = torch.autograd.grad(o_losses[k], [p[n] for n in names], retain_graph=True, allow_unused=True)
    gd = torch.autograd.grad(losses[k], [ours[n] for n in names], retain_graph=True, allow_unused=True)
    worst = []
    for n, a, b in zip(names, gd, go):
        if b is None or a is None or float(b.abs().max()) == 0.0:
            continue
        e = ((a.detach().cpu().double() - b).norm() / (b.norm() + 1e-300)).item()
        worst.append((e, n))
    worst.sort(reverse=True)
    print("   worst rel-l2:", ", ".join(f"{n} {e:.1e}" for e, n in worst[:6]))
    print("   best  rel-l2:", ", ".join(f"{n} {e:.1e}" for e, n in worst[-3:]))
    if os.environ.get("VP_DIAG_ALL") and k == "loss_recon":
        for e, n in sorted(worst, key=lambda t: names.index(t[1])):
            print(f"      {n:40s} {e:.1e}")
