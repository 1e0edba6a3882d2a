"""Decoder alone, loss = mse(x, decoder(z)): gradient w.r.t. every block output and parameter, HIP modules vs the oracle in
fp64 (and the oracle in fp32 for scale).  usage: python tests/diag/decoder_bwd_diag.py [img] [z] [batch]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_cpu as O, ref_vaegan as G  # noqa: E402  (checker only)
import vae_play_amd as V  # noqa: E402

S, z, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 32, 4)
L = O.iter_level_for(S)
g = torch.Generator().manual_seed(5)
zin = torch.randn(B, z, generator=g)
x = torch.rand(B, 1, S, S, generator=g)
p0 = G.init_vaegan_params(S, z, seed=0)


def oracle(dtype):
    p = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in p0.items()}
    O.require_grad(p)
    acts = []
    t = F.linear(zin.to(dtype), p["decoder.fc.0.weight"])
    acts.append(("fc.0 out", t)); t.retain_grad()
    t = F.relu(O._bn(p, "decoder.fc.1", t, True))
    acts.append(("fc act", t)); t.retain_grad()
    t = t.view(len(t), -1, 8, 8)
    for i in range(L):
        c = F.conv_transpose2d(t, p[f"decoder.conv.{i}.conv.weight"], None, stride=2, padding=2, output_padding=1)
        c.retain_grad(); acts.append((f"conv.{i} convT out", c))
        t = F.relu(O._bn(p, f"decoder.conv.{i}.bn", c, True))
        t.retain_grad(); acts.append((f"conv.{i} act", t))
    y = torch.sigmoid(F.conv2d(t, p[f"decoder.conv.{L}.0.weight"], p[f"decoder.conv.{L}.0.bias"], stride=1, padding=2))
    F.mse_loss(x.to(dtype), y).backward()
    return ({n: a.grad.double() for n, a in acts},
            {n: p[n].grad.double() for n in O.trainable_names(p) if n.startswith("decoder.") and p[n].grad is not None})


a64, p64 = oracle(torch.float64)
a32, p32 = oracle(torch.float32)
net = V.VaeGan(S, z)
net.load_state_dict(p0, strict=True)
dec = net.decoder.cuda().train()
caps = {}


def keep(name):
    def hook(mod, inp, out):
        out.retain_grad(); caps[name] = out
    return hook


dec.fc[0].register_forward_hook(keep("fc.0 out"))
for i in range(L):
    dec.conv[i].conv.register_forward_hook(keep(f"conv.{i} convT out"))
    dec.conv[i].register_forward_hook(keep(f"conv.{i} act"))
y = dec(zin.cuda())
F.mse_loss(x.cuda(), y).backward()
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print("gradient w.r.t. activations (rel-l2 vs fp64):        hip      oracle-fp32")
for n in a64:
    if n in caps and caps[n].grad is not None:
        gh = caps[n].grad.detach().cpu().double().reshape(a64[n].shape)
        print(f"  {n:24s} {rel(gh, a64[n]):9.1e} {rel(a32[n], a64[n]):9.1e}")
print("gradient w.r.t. parameters:")
for n, q in net.named_parameters():
    if n in p64 and q.grad is not None:
        print(f"  {n:32s} {rel(q.grad.detach().cpu().double(), p64[n]):9.1e} {rel(p32[n], p64[n]):9.1e}")
