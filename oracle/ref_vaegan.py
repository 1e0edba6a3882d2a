"""CPU oracle for the VAE-GAN branch -- TEST INFRASTRUCTURE ONLY.

Functional restatement (plain torch on CPU, parameters in a dict keyed like the reference's
``state_dict``) of kungyao/vae-play's

  * ``DirectDecoder``           models/networks.py:118-148
  * ``Discriminator``           models/networks.py:151-198
  * ``VaeGan.forward`` (train)  models/networks.py:233-247   (eps and z_p injected instead of drawn)
  * ``VaeGan.loss``             models/networks.py:265-281
  * the five losses, the accumulate-then-step backward and the four RMSprop steps of train.py:43-78,136-140

Pinned: oracle/gen_golden.py runs the reference's own modules on the same seeded inputs and asserts that
every output, gradient and updated parameter of this file is bit-identical before it writes
tests/golden/vaegan_*.npz.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it;
the product package (vae_play_amd/) never does.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from . import ref_cpu as O

Params = Dict[str, torch.Tensor]
LAMBDA_MSE = 1e-6       # train.py:15
DISC_WIDTH0 = 32        # models/networks.py:160


def discriminator_channels(iter_level: int) -> List[int]:
    ch = [DISC_WIDTH0]
    for _ in range(iter_level):
        ch.append(ch[-1] * 2)
    return ch


def discriminator_spec(channel_in: int, iter_level: int):
    """(name, shape, kind) in state_dict order of the reference Discriminator."""
    ch = discriminator_channels(iter_level)
    out = [("conv.0.0.weight", (DISC_WIDTH0, channel_in, 5, 5), "conv_w"), ("conv.0.0.bias", (DISC_WIDTH0,), "conv_b")]
    for i in range(iter_level):
        out.append((f"conv.{i + 1}.conv.weight", (ch[i + 1], ch[i], 5, 5), "conv_w"))
        out += O._bn_entries(f"conv.{i + 1}.bn", ch[i + 1])
    out.append(("fc.0.weight", (512, 8 * 8 * ch[-1]), "lin_w"))
    out += O._bn_entries("fc.1", 512)
    out.append(("fc.3.weight", (1, 512), "lin_w"))
    out.append(("fc.3.bias", (1,), "lin_b"))
    return out


def direct_decoder_spec(z_size: int):
    dims = {"head": [z_size, 512, 256, 128, 64], "r_fc": [64, 32, 1], "xy_fc": [64, 32, 2]}
    out = []
    for grp in ("head", "r_fc", "xy_fc"):
        d = dims[grp]
        for i in range(len(d) - 1):
            out.append((f"{grp}.{i}.weight", (d[i + 1], d[i]), "lin_w"))
            out.append((f"{grp}.{i}.bias", (d[i + 1],), "lin_b"))
    return out


def vaegan_spec(img_size: int, z_size: int):
    L = O.iter_level_for(img_size)
    enc = O.encoder_spec(1, z_size, L)
    size = O.encoder_channels(1, L)[-1]
    dec = O.decoder_spec(z_size, size, 1, L)
    dis = discriminator_spec(1, L)
    aux = direct_decoder_spec(z_size)
    return ([("encoder." + n, s, k) for n, s, k in enc] + [("decoder." + n, s, k) for n, s, k in dec]
            + [("discriminator." + n, s, k) for n, s, k in dis] + [("param_encoder." + n, s, k) for n, s, k in aux])


def init_vaegan_params(img_size: int, z_size: int, seed: int = 0) -> Params:
    """VaeGan.init_parameters (models/networks.py:214-226) with a private generator: weights
    ~ U(+-1/sqrt(3*prod(shape[1:]))) in modules() == state_dict order, biases 0, BN at its defaults."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shape, kind in vaegan_spec(img_size, z_size):
        if kind in ("conv_w", "lin_w"):
            scale = 1.0 / math.sqrt(float(torch.Size(shape[1:]).numel())) / math.sqrt(3.0)
            p[name] = (torch.rand(shape, generator=g) * 2.0 - 1.0) * scale
        elif kind in ("conv_b", "lin_b", "bn_b", "bn_rm"):
            p[name] = torch.zeros(shape)
        elif kind in ("bn_w", "bn_rv"):
            p[name] = torch.ones(shape)
        elif kind == "bn_nbt":
            p[name] = torch.zeros((), dtype=torch.long)
    return p


def direct_decoder_forward(p: Params, z: torch.Tensor, prefix: str = "param_encoder.") -> torch.Tensor:
    """models/networks.py:143-148: no activations anywhere."""
    t = z
    for i in range(4):
        t = F.linear(t, p[f"{prefix}head.{i}.weight"], p[f"{prefix}head.{i}.bias"])
    r = t
    for i in range(2):
        r = F.linear(r, p[f"{prefix}r_fc.{i}.weight"], p[f"{prefix}r_fc.{i}.bias"])
    xy = t
    for i in range(2):
        xy = F.linear(xy, p[f"{prefix}xy_fc.{i}.weight"], p[f"{prefix}xy_fc.{i}.bias"])
    return torch.cat([r, xy], dim=-1)


def discriminator_forward(p: Params, orig, predicted, sampled, mode: str, recon_level: int, iter_level: int,
                          training: bool = True, prefix: str = "discriminator."):
    """models/networks.py:174-195.  REC: the pre-BatchNorm output of conv[recon_level], flattened in NCHW order
    (the block's BN still runs and updates its running statistics before the early return)."""
    t = torch.cat((orig, predicted, sampled), 0)
    for i in range(iter_level + 1):
        if i == 0:
            t = F.relu(F.conv2d(t, p[prefix + "conv.0.0.weight"], p[prefix + "conv.0.0.bias"], stride=1, padding=2))
            continue
        c = F.conv2d(t, p[f"{prefix}conv.{i}.conv.weight"], None, stride=2, padding=2)
        t = F.relu(O._bn(p, f"{prefix}conv.{i}.bn", c, training))
        if mode == "REC" and i == recon_level:
            return c.view(len(c), -1)
    t = t.view(len(t), -1)
    t = F.linear(t, p[prefix + "fc.0.weight"])
    t = F.relu(O._bn(p, prefix + "fc.1", t, training))
    t = F.linear(t, p[prefix + "fc.3.weight"], p[prefix + "fc.3.bias"])
    return torch.sigmoid(t)


def vaegan_forward(p: Params, x, eps, z_p, img_size: int, training: bool = True):
    """VaeGan.forward, training branch (models/networks.py:234-247)."""
    L = O.iter_level_for(img_size)
    mus, logvar = O.encoder_forward(p, x, L, training)
    z = O.reparameterize(mus, logvar, eps)
    x_tilde = O.decoder_forward(p, z, L, training)
    params = direct_decoder_forward(p, z)
    x_p = O.decoder_forward(p, z_p, L, training)
    disc_layer = discriminator_forward(p, x, x_tilde, x_p, "REC", L, L, training)
    disc_class = discriminator_forward(p, x, x_tilde, x_p, "GAN", L, L, training)
    return x_tilde, disc_class, disc_layer, mus, logvar, params


def vaegan_loss(x, x_tilde, dl_orig, dl_pred, dl_samp, dc_orig, dc_pred, dc_samp, mus, variances, targets, params):
    """VaeGan.loss (models/networks.py:265-281)."""
    nle = 0.5 * (x.view(len(x), -1) - x_tilde.view(len(x_tilde), -1)) ** 2
    kl = -0.5 * torch.sum(-variances.exp() - torch.pow(mus, 2) + variances + 1, 1)
    mse = torch.sum(0.5 * (dl_orig - dl_pred) ** 2, 1)
    bce_o = -torch.log(dc_orig + 1e-3)
    bce_p = -torch.log(1 - dc_pred + 1e-3)
    bce_s = -torch.log(1 - dc_samp + 1e-3)
    l1 = F.smooth_l1_loss(targets, params, reduction="sum") / x.size(0)
    return nle, kl, mse, bce_o, bce_p, bce_s, l1


def train_losses(p: Params, x, targets, eps, z_p, img_size: int):
    """train.py:43-66: forward, split of the 3B discriminator outputs, the five scalar losses."""
    B = x.size(0)
    x_tilde, disc_class, disc_layer, mus, logvar, params = vaegan_forward(p, x, eps, z_p, img_size, True)
    dl = (disc_layer[:B], disc_layer[B:-B], disc_layer[-B:])
    dc = (disc_class[:B], disc_class[B:-B], disc_class[-B:])
    nle, kl, mse, bo, bp, bs, l1 = vaegan_loss(x, x_tilde, *dl, *dc, mus, logvar, targets, params)
    loss_recon = F.mse_loss(x, x_tilde)
    loss_encoder = torch.sum(kl) + torch.sum(mse)
    loss_discriminator = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    loss_decoder = torch.sum(LAMBDA_MSE * mse) - (1.0 - LAMBDA_MSE) * loss_discriminator
    loss_aux = l1
    out = {"x_tilde": x_tilde, "disc_class": disc_class, "disc_layer": disc_layer, "mus": mus, "logvar": logvar,
           "params": params, "nle": nle, "kl": kl, "mse": mse, "bce_dis_original": bo, "bce_dis_predicted": bp,
           "bce_dis_sampled": bs, "l1_enc_param": l1}
    losses = {"loss_recon": loss_recon, "loss_encoder": loss_encoder, "loss_decoder": loss_decoder,
              "loss_discriminator": loss_discriminator, "loss_aux": loss_aux}
    return out, losses


GROUPS = ("encoder.", "decoder.", "discriminator.", "param_encoder.")


def make_optimizers(p: Params, lr: float = 1e-4):
    """train.py:136-140: one RMSprop (torch defaults) per sub-network."""
    names = O.trainable_names(p)
    return [torch.optim.RMSprop([p[n] for n in names if n.startswith(g)], lr=lr) for g in GROUPS]


def train_step(p: Params, opts, x, targets, eps, z_p, img_size: int):
    """train.py:68-78: zero_grad, five backward passes over one graph (gradients accumulate), four steps."""
    for n in O.trainable_names(p):
        p[n].grad = None
    out, losses = train_losses(p, x, targets, eps, z_p, img_size)
    order = ("loss_recon", "loss_encoder", "loss_decoder", "loss_discriminator", "loss_aux")
    for i, k in enumerate(order):
        losses[k].backward(retain_graph=i + 1 < len(order))
    if opts is not None:
        for o in opts:
            o.step()
    return {k: v.detach() for k, v in out.items()}, {k: v.detach() for k, v in losses.items()}


def seeded_disc_weights(state_dict, seed: int = 77):
    """Deterministic weights for a Discriminator state_dict, regenerated identically by the tests (the 4M-element
    fc.0.weight is not stored in the fixture): N(0, 0.05^2) weights, U(0.5, 1.5) BN scales, N(0, 0.1^2) biases,
    drawn in key order."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in state_dict.items():
        if not v.dtype.is_floating_point or "running" in k:
            out[k] = v.clone()
        elif k.endswith(".bn.weight") or k == "fc.1.weight":
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith("weight"):
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        else:
            out[k] = torch.randn(v.shape, generator=g) * 0.1
    return out


def synthetic_batch(B: int, S: int, z: int):
    """Inputs of the shape train.py feeds: images in [0,1] (B,1,S,S), circle parameters (B,3) in [0,1]."""
    g = torch.Generator().manual_seed(2468)
    x = torch.rand(B, 1, S, S, generator=g)
    targets = torch.rand(B, 3, generator=g)
    eps = torch.randn(B, z, generator=g)
    z_p = torch.randn(B, z, generator=g)
    return x, targets, eps, z_p
