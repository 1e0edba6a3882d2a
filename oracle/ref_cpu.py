"""CPU oracle for the convolutional-VAE training step.  TEST INFRASTRUCTURE ONLY.

This file is a *restatement* of the reference algorithm (kungyao/vae-play) in
plain fp32 torch CPU ops, written functionally over a flat ``{name: tensor}``
parameter dict.  It is what the HIP path is checked against.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``vae_play_amd/`` does, and the product path raises if
its HIP library is missing instead of falling back to anything in here.

Pinning: ``oracle/gen_golden.py`` imports the real reference modules from
``/root/reference`` (authoring container only), runs them on the same seeded
inputs and asserts this restatement reproduces them bit-for-bit before the
fixtures under ``tests/golden/`` are written.  Parity is therefore *pinned* by
reference-generated vectors (the reference has no tests/fixtures of its own,
SURVEY.md section 4).

Reference citations (relative to /root/reference):
  EncoderBlock   models/networks.py:10-30    conv5 s2 p2 (no bias) -> BN2d(momentum .9) -> ReLU
  DecoderBlock   models/networks.py:34-46    convT5 s2 p2 op1 (no bias) -> BN2d(momentum .9) -> ReLU
  Encoder        models/networks.py:49-81
  Decoder        models/networks.py:84-115
  reparameterize models/networks.py:228-231
  KL             models/networks.py:270
  pixel BCE      torch F.binary_cross_entropy, call form train_BE_font.py:107
  init rule      models/networks.py:214-226
  optimiser step train_BE.py:62-64,131 (Adam) / train.py:136-140 (RMSprop)
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BN_MOMENTUM = 0.9   # models/networks.py:16,40,66,89
BN_EPS = 1e-5       # torch default, not overridden by the reference

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# structure: names / shapes in the reference's registration order
# --------------------------------------------------------------------------
def encoder_channels(channel_in: int, iter_level: int) -> List[int]:
    """Channel progression of Encoder.__init__ (models/networks.py:55-61)."""
    chans = [channel_in]
    for i in range(iter_level):
        chans.append(64 if i == 0 else chans[-1] * 2)
    return chans


def decoder_channels(size: int, iter_level: int) -> List[int]:
    """Channel progression of Decoder.__init__ (models/networks.py:93-96)."""
    chans = [size, size]
    for _ in range(iter_level - 1):
        chans.append(chans[-1] // 2)
    return chans


def _bn_entries(prefix: str, n: int):
    return [
        (prefix + ".weight", (n,), "bn_w"),
        (prefix + ".bias", (n,), "bn_b"),
        (prefix + ".running_mean", (n,), "bn_rm"),
        (prefix + ".running_var", (n,), "bn_rv"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def encoder_spec(channel_in: int, z_size: int, iter_level: int):
    """(name, shape, kind) in state_dict order of the reference Encoder."""
    ch = encoder_channels(channel_in, iter_level)
    out = []
    for i in range(iter_level):
        out.append((f"conv.{i}.conv.weight", (ch[i + 1], ch[i], 5, 5), "conv_w"))
        out += _bn_entries(f"conv.{i}.bn", ch[i + 1])
    size = ch[-1]
    out.append(("fc.0.weight", (1024, 8 * 8 * size), "lin_w"))
    out += _bn_entries("fc.1", 1024)
    out.append(("l_mu.weight", (z_size, 1024), "lin_w"))
    out.append(("l_mu.bias", (z_size,), "lin_b"))
    out.append(("l_var.weight", (z_size, 1024), "lin_w"))
    out.append(("l_var.bias", (z_size,), "lin_b"))
    return out


def decoder_spec(z_size: int, size: int, channel_out: int, iter_level: int):
    """(name, shape, kind) in state_dict order of the reference Decoder."""
    ch = decoder_channels(size, iter_level)
    out = [("fc.0.weight", (8 * 8 * size, z_size), "lin_w")]
    out += _bn_entries("fc.1", 8 * 8 * size)
    for i in range(iter_level):
        # ConvTranspose2d weight layout is (Cin, Cout, k, k)
        out.append((f"conv.{i}.conv.weight", (ch[i], ch[i + 1], 5, 5), "conv_w"))
        out += _bn_entries(f"conv.{i}.bn", ch[i + 1])
    out.append((f"conv.{iter_level}.0.weight", (channel_out, ch[-1], 5, 5), "conv_w"))
    out.append((f"conv.{iter_level}.0.bias", (channel_out,), "conv_b"))
    return out


def vae_spec(channel_in: int, z_size: int, iter_level: int):
    enc = encoder_spec(channel_in, z_size, iter_level)
    size = encoder_channels(channel_in, iter_level)[-1]
    dec = decoder_spec(z_size, size, channel_in, iter_level)
    return [("encoder." + n, s, k) for n, s, k in enc] + [("decoder." + n, s, k) for n, s, k in dec]


def init_params(channel_in: int, z_size: int, iter_level: int, seed: int = 0) -> Params:
    """Seeded weights with the VaeGan.init_parameters rule (models/networks.py:214-226):
    every conv / convT / linear weight ~ U(+-1/sqrt(3*prod(shape[1:]))), biases 0,
    BN gamma 1 / beta 0 / running (0, 1).  Draw order = state_dict order."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}
    for name, shape, kind in vae_spec(channel_in, z_size, iter_level):
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


def trainable_names(params: Params) -> List[str]:
    return [n for n in params if not n.endswith(("running_mean", "running_var", "num_batches_tracked"))]


# --------------------------------------------------------------------------
# forward pieces
# --------------------------------------------------------------------------
def _bn(p: Params, prefix: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    rm, rv = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    y = F.batch_norm(x, rm, rv, p[prefix + ".weight"], p[prefix + ".bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training:
        p[prefix + ".num_batches_tracked"] += 1
    return y


def encoder_block(p: Params, prefix: str, x: torch.Tensor, training: bool = True):
    """models/networks.py:26-30."""
    y = F.conv2d(x, p[prefix + ".conv.weight"], None, stride=2, padding=2)
    return F.relu(_bn(p, prefix + ".bn", y, training))


def decoder_block(p: Params, prefix: str, x: torch.Tensor, training: bool = True):
    """models/networks.py:42-46."""
    y = F.conv_transpose2d(x, p[prefix + ".conv.weight"], None, stride=2, padding=2, output_padding=1)
    return F.relu(_bn(p, prefix + ".bn", y, training))


def encoder_forward(p: Params, x: torch.Tensor, iter_level: int, training: bool = True,
                    prefix: str = "encoder.") -> Tuple[torch.Tensor, torch.Tensor]:
    """models/networks.py:72-78."""
    t = x
    for i in range(iter_level):
        t = encoder_block(p, f"{prefix}conv.{i}", t, training)
    t = t.reshape(len(t), -1)                      # NCHW (C,H,W) flatten, :74
    t = F.linear(t, p[prefix + "fc.0.weight"])
    t = F.relu(_bn(p, prefix + "fc.1", t, training))
    mu = F.linear(t, p[prefix + "l_mu.weight"], p[prefix + "l_mu.bias"])
    logvar = F.linear(t, p[prefix + "l_var.weight"], p[prefix + "l_var.bias"])
    return mu, logvar


def reparameterize(mu: torch.Tensor, logvar: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """models/networks.py:228-231 with eps injected (the reference draws it with normal_())."""
    std = logvar.mul(0.5).exp()
    return eps.mul(std).add(mu)


def decoder_forward(p: Params, z: torch.Tensor, iter_level: int, training: bool = True,
                    prefix: str = "decoder.") -> torch.Tensor:
    """models/networks.py:107-112."""
    t = F.linear(z, p[prefix + "fc.0.weight"])
    t = F.relu(_bn(p, prefix + "fc.1", t, training))
    t = t.view(len(t), -1, 8, 8)
    for i in range(iter_level):
        t = decoder_block(p, f"{prefix}conv.{i}", t, training)
    t = F.conv2d(t, p[f"{prefix}conv.{iter_level}.0.weight"], p[f"{prefix}conv.{iter_level}.0.bias"],
                 stride=1, padding=2)
    return torch.sigmoid(t)


def kl_per_sample(mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
    """models/networks.py:270."""
    return -0.5 * torch.sum(-logvar.exp() - torch.pow(mu, 2) + logvar + 1, 1)


def vae_loss(x: torch.Tensor, x_tilde: torch.Tensor, mu: torch.Tensor, logvar: torch.Tensor):
    """(sum-BCE + sum-KL) / B."""
    recon = F.binary_cross_entropy(x_tilde, x, reduction="sum")
    kl = kl_per_sample(mu, logvar).sum()
    return (recon + kl) / x.shape[0], recon, kl


def vae_forward(p: Params, x: torch.Tensor, eps: torch.Tensor, iter_level: int, training: bool = True):
    mu, logvar = encoder_forward(p, x, iter_level, training)
    z = reparameterize(mu, logvar, eps)
    x_tilde = decoder_forward(p, z, iter_level, training)
    loss, recon, kl = vae_loss(x, x_tilde, mu, logvar)
    return {"mu": mu, "logvar": logvar, "z": z, "x_tilde": x_tilde, "loss": loss, "recon": recon, "kl": kl}


# --------------------------------------------------------------------------
# training step
# --------------------------------------------------------------------------
def make_optimizer(params: Params, kind: str = "adam", lr: float = 1e-4):
    """train_BE.py:131 (Adam, torch defaults) or train.py:137 (RMSprop, torch defaults)."""
    leaves = [params[n] for n in trainable_names(params)]
    if kind == "adam":
        return torch.optim.Adam(leaves, lr=lr)
    if kind == "rmsprop":
        return torch.optim.RMSprop(leaves, lr=lr)
    raise ValueError(kind)


def require_grad(params: Params) -> None:
    for n in trainable_names(params):
        params[n].requires_grad_(True)


def train_step(p: Params, opt: Optional[torch.optim.Optimizer], x: torch.Tensor, eps: torch.Tensor,
               iter_level: int, grad_hook=None):
    """One composed step: zero_grad -> fwd -> loss -> backward -> [grad_hook] -> step
    (train_BE.py:62-64 idiom).  grad_hook(params) lets the data-parallel oracle average
    gradients across shards before the update."""
    for n in trainable_names(p):
        p[n].grad = None
    out = vae_forward(p, x, eps, iter_level, training=True)
    out["loss"].backward()
    if grad_hook is not None:
        grad_hook(p)
    if opt is not None:
        opt.step()
    return {k: v.detach() for k, v in out.items()}


def clone_params(p: Params) -> Params:
    return {k: v.detach().clone() for k, v in p.items()}


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def synthetic_batch(B: int, C: int, S: int, z: int, rank: int = 0):
    gx = torch.Generator().manual_seed(1234 + rank)
    ge = torch.Generator().manual_seed(4321 + rank)
    x = torch.rand(B, C, S, S, generator=gx)
    eps = torch.randn(B, z, generator=ge)
    return x, eps


def iter_level_for(img_size: int) -> int:
    """models/networks.py:204."""
    return int(math.log2(img_size // 8))


# --------------------------------------------------------------------------
# checksums used by the golden fixtures
# --------------------------------------------------------------------------
def sample_indices(numel: int, n: int = 16, seed: int = 7) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed + numel % 9973)
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def checksum(t: torch.Tensor) -> Dict[str, torch.Tensor]:
    f = t.detach().double().flatten()
    idx = sample_indices(f.numel())
    return {"sum": f.sum().reshape(1), "l2": f.pow(2).sum().sqrt().reshape(1),
            "samples": t.detach().flatten()[idx].double()}


# --------------------------------------------------------------------------
# models/blocks.py vocabulary: functional restatement over a params dict whose keys are the
# reference state_dict keys ("conv.0.weight", "conv.1.running_mean", ...)
# --------------------------------------------------------------------------
def _blocks_act(y: torch.Tensor, act: Optional[str], lrelu_slope: float) -> torch.Tensor:
    if act == "relu":
        return F.relu(y)
    if act == "lrelu":
        return F.leaky_relu(y, lrelu_slope)
    if act == "tanh":
        return torch.tanh(y)
    return y


def blocks_conv2d(p: Params, prefix: str, x: torch.Tensor, kernel_size: int, stride: int = 1, bn: Optional[str] = None,
                  act: Optional[str] = "relu", training: bool = True) -> torch.Tensor:
    """models/blocks.py:5-34: conv(pad=(k-1)//2, bias iff bn is None) -> {BatchNorm2d | InstanceNorm2d | -} -> act."""
    y = F.conv2d(x, p[prefix + "conv.0.weight"], p.get(prefix + "conv.0.bias"), stride=stride, padding=(kernel_size - 1) // 2)
    if bn == "batch":
        y = F.batch_norm(y, p[prefix + "conv.1.running_mean"], p[prefix + "conv.1.running_var"], p[prefix + "conv.1.weight"],
                         p[prefix + "conv.1.bias"], training, 0.1, 1e-5)
        if training:
            p[prefix + "conv.1.num_batches_tracked"] += 1
    elif bn == "instance":
        y = F.instance_norm(y, eps=1e-5)
    return _blocks_act(y, act, 0.02)


def blocks_linear(p: Params, prefix: str, x: torch.Tensor, act: Optional[str] = "relu") -> torch.Tensor:
    """models/blocks.py:36-50 (LeakyReLU slope 0.2)."""
    return _blocks_act(F.linear(x, p[prefix + "fc.0.weight"], p.get(prefix + "fc.0.bias")), act, 0.2)


def blocks_add_coords(x: torch.Tensor, if_normalize: bool = False) -> torch.Tensor:
    """models/blocks.py:97-112: append column-index and row-index channels."""
    b, c, h, w = x.shape
    ci = torch.arange(0, w, dtype=x.dtype).reshape(1, 1, 1, -1).repeat(b, 1, h, 1)
    cj = torch.arange(0, h, dtype=x.dtype).reshape(1, 1, -1, 1).repeat(b, 1, 1, w)
    if if_normalize:
        ci = (ci / w - 0.5) / 0.5
        cj = (cj / h - 0.5) / 0.5
    return torch.cat([x, ci, cj], dim=1)


def blocks_down(p: Params, prefix: str, x: torch.Tensor, kernel_size: int, if_add_coord: bool = False, training: bool = True):
    """models/blocks.py:114-127."""
    if if_add_coord:
        x = blocks_add_coords(x)
    return blocks_conv2d(p, prefix + "conv.", x, kernel_size, 2, None, "relu", training)


def blocks_up(p: Params, prefix: str, x: torch.Tensor, if_add_coord: bool = False, training: bool = True):
    """models/blocks.py:129-146: (AddCoords) -> 2 x [conv3 + BN + ReLU] -> bilinear x2 (align_corners=False)."""
    if if_add_coord:
        x = blocks_add_coords(x)
    x = blocks_conv2d(p, prefix + "conv.0.", x, 3, 1, "batch", "relu", training)
    x = blocks_conv2d(p, prefix + "conv.1.", x, 3, 1, "batch", "relu", training)
    return F.interpolate(x, scale_factor=2, mode="bilinear")
