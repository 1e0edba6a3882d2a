"""Where along the backward chain do HIP gradients leave the fp64 oracle?  Activation gradients (d loss / d block output) of
the decoder blocks, HIP autograd modules vs the oracle evaluated in fp64, plus the BatchNorm parameter gradients.
usage: python tests/diag/step_bwd_chain_diag.py [S] [z] [B] [precision]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu as O  # noqa: E402  (checker only)
import vae_play_amd as V  # noqa: E402

S, z, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 64, 4)
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
C = 3
L = O.iter_level_for(S)
x, eps = O.synthetic_batch(B, C, S, z)
p0 = O.init_params(C, z, L, seed=0)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


# fp64 oracle with retained intermediate gradients
p = {k: (v.double().clone() if v.dtype.is_floating_point else v.clone()) for k, v in p0.items()}
O.require_grad(p)
mu, logvar = O.encoder_forward(p, x.double(), L)
zz = O.reparameterize(mu, logvar, eps.double())
t = F.linear(zz, p["decoder.fc.0.weight"])
t = F.relu(O._bn(p, "decoder.fc.1", t, True)).view(B, -1, 8, 8)
acts64 = []
for i in range(L):
    t = O.decoder_block(p, f"decoder.conv.{i}", t)
    t.retain_grad()
    acts64.append(t)
logit = F.conv2d(t, p[f"decoder.conv.{L}.0.weight"], p[f"decoder.conv.{L}.0.bias"], padding=2)
logit.retain_grad()
xt = torch.sigmoid(logit)
loss, _, _ = O.vae_loss(x.double(), xt, mu, logvar)
loss.backward()

# HIP modules
V.set_conv_precision(prec)
vae = V.VAE(S, z, C, init_rule=False)
vae.load_state_dict(p0)
vae = vae.cuda().train()
xd, ed = x.cuda(), eps.cuda()
mu_h, lv_h = vae.encoder(xd)
z_h = V.reparameterize(mu_h, lv_h, eps=ed)
t = vae.decoder.fc(z_h)
from vae_play_amd import functional as FH  # noqa: E402
t = FH.unflatten_nchw(t, vae.decoder._c0, 8, 8)
acts = []
for i in range(L):
    t = vae.decoder.conv[i](t)
    t.retain_grad()
    acts.append(t)
xt_h = vae.decoder.conv[L](t)
loss_h = FH.vae_loss(xd, xt_h, mu_h, lv_h)[0]
loss_h.backward()
print(f"S={S} z={z} B={B} precision={prec}: loss {loss_h.item():.8g} vs fp64 {loss.item():.8g}")
print("x_tilde rel err", rel(xt_h, xt))
for i in range(L - 1, -1, -1):
    print(f"  d loss / d decoder.conv.{i} output: rel err {rel(acts[i].grad, acts64[i].grad):.2e}   (activation itself {rel(acts[i], acts64[i]):.2e})")
    g = acts64[i].grad
    gh = acts[i].grad.detach().cpu().double()
    d = gh - g
    print(f"      mean error / rms error per channel (systematic share): {(d.mean(dim=(0, 2, 3)).abs().mean() / d.pow(2).mean().sqrt()).item():.2e};"
          f" |mean g| / rms g: {(g.mean(dim=(0, 2, 3)).abs().mean() / g.pow(2).mean().sqrt()).item():.2e}")
hp = dict(vae.named_parameters())
for n in [k for k in O.trainable_names(p) if k.startswith("decoder.conv")]:
    print(f"  {n:32s} {rel(hp[n].grad, p[n].grad):.2e}")
