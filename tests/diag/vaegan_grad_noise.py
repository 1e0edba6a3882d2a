"""Per-tensor deviation of the VAE-GAN step's accumulated gradients from an fp64 run of the oracle: the oracle in fp32 (the
reference's own arithmetic noise), and the HIP modules with the skinny dense kernels off / on.
usage: python tests/diag/vaegan_grad_noise.py [img] [z] [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu as O, ref_vaegan as G  # noqa: E402  (checker only)

S, z, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 32, 4)


def oracle_grads(dtype):
    p = G.init_vaegan_params(S, z, seed=0)
    p = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in p.items()}
    O.require_grad(p)
    x, t, e, zp = [a.to(dtype) for a in G.synthetic_batch(B, S, z)]
    G.train_step(p, None, x, t, e, zp, S)
    return {n: p[n].grad.double() for n in O.trainable_names(p) if p[n].grad is not None}


def hip_grads(skinny, blocks=None):
    import vae_play_amd as V
    os.environ["VP_GEMM_SKINNY"] = skinny
    if blocks:
        os.environ["VP_GEMM_BLOCKS"] = blocks     # another split-K count in the tile kernels: same arithmetic, other rounding order
    else:
        os.environ.pop("VP_GEMM_BLOCKS", None)
    net = V.VaeGan(S, z)
    net.load_state_dict(G.init_vaegan_params(S, z, seed=0), strict=True)
    net = net.cuda().train()
    x, targets, eps, z_p = [a.cuda() for a in G.synthetic_batch(B, S, z)]
    x_tilde, dc, dl, mus, logvar, params = net(x, eps=eps, z_p=z_p)
    terms = V.VaeGan.loss(x, x_tilde, dl[:B], dl[B:-B], dl[-B:], dc[:B], dc[B:-B], dc[-B:], mus, logvar, targets, params)
    nle, kl, mse, bo, bp, bs, l1 = terms
    lam = G.LAMBDA_MSE
    ld = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    losses = [F.mse_loss(x, x_tilde), torch.sum(kl) + torch.sum(mse), torch.sum(lam * mse) - (1.0 - lam) * ld, ld, l1]
    net.zero_grad()
    for i, l in enumerate(losses):
        l.backward(retain_graph=i + 1 < len(losses))
    return {n: p.grad.detach().cpu().double() for n, p in net.named_parameters() if p.grad is not None}


g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
h0, h1, h2 = hip_grads("0"), hip_grads("1"), hip_grads("0", "96")
print(f"{'tensor':44s} {'rms':>9s} | max|d|/rms vs fp64: oracle32  hip(skinny=0)  hip(skinny=1)  hip(skinny=0, other split-K)")
for n in g64:
    rms = g64[n].pow(2).mean().sqrt().item()
    if rms < 1e-9 or not n.startswith(("encoder", "decoder", "param")):
        continue
    f = lambda g: (g[n] - g64[n]).abs().max().item() / rms
    print(f"{n:44s} {rms:9.2e} | {f(g32):9.2e} {f(h0):9.2e} {f(h1):9.2e} {f(h2):9.2e}")

# where do the deviations sit?  a ReLU mask that flips at one hidden unit changes one ROW of the Linear weight gradient in front of it
for n in ("encoder.fc.0.weight", "decoder.fc.0.weight", "discriminator.fc.0.weight"):
    if n not in g64:
        continue
    rms = g64[n].pow(2).mean().sqrt().item()
    for tag, h in (("skinny=0", h0), ("skinny=1", h1), ("other split-K", h2)):
        rowdev = (h[n] - g64[n]).abs().amax(dim=1) / rms
        bad = (rowdev > 20 * rowdev.median()).nonzero().flatten().tolist()
        print(f"{n} {tag}: median row deviation {rowdev.median().item():.1e}, rows > 20 x median: {bad[:8]} ({len(bad)} of {rowdev.numel()})")
