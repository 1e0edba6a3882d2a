"""Drop-in VAE modules for kungyao/vae-play's ``models/networks.py`` backed by HIP kernels.

Same constructor signatures, attribute names (``.size``, ``.conv``, ``.fc``, ``.l_mu``,
``.l_var``), ``state_dict`` keys and default-init RNG consumption as the reference classes
(so ``torch.manual_seed(s); Encoder(...)`` yields bit-identical weights), but ``forward``
runs on libvaeplay_hip.so: NHWC implicit-GEMM convolutions on the MFMA units, fused
BatchNorm+ReLU, HIP GEMMs for the dense layers.  Module I/O is logical NCHW fp32 like the
reference; outputs are channels_last in memory.

  EncoderBlock  <- models/networks.py:10-30      Encoder <- models/networks.py:49-81
  DecoderBlock  <- models/networks.py:34-46      Decoder <- models/networks.py:84-115
  reparameterize <- models/networks.py:228-231   init_parameters <- models/networks.py:214-226
  DirectDecoder <- models/networks.py:118-148    Discriminator <- models/networks.py:151-198
  VaeGan        <- models/networks.py:201-281 (forward, loss); training idiom train.py:43-78
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import functional as F_hip

BN_MOMENTUM = 0.9  # models/networks.py:16,40,66,89


class Conv5x5Params(nn.Module):
    """Parameter holder with nn.Conv2d / nn.ConvTranspose2d's weight layout, key names and
    default initialisation (kaiming_uniform_(a=sqrt(5)) + fan-in-bounded bias), 5x5 kernel."""

    def __init__(self, dim0: int, dim1: int, bias: bool, stride: int, transposed: bool):
        super().__init__()
        self.stride, self.transposed = stride, transposed
        self.weight = nn.Parameter(torch.empty(dim0, dim1, 5, 5))
        self.bias = nn.Parameter(torch.empty(dim1 if transposed else dim0)) if bias else None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            if fan_in != 0:
                bound = 1 / math.sqrt(fan_in)
                nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: torch.Tensor, act: Optional[str] = None) -> torch.Tensor:
        if self.transposed:
            return F_hip.conv_transpose5x5(x, self.weight, self.stride)
        return F_hip.conv5x5(x, self.weight, self.bias, self.stride, act)

    def extra_repr(self) -> str:
        kind = "ConvTranspose2d" if self.transposed else "Conv2d"
        return f"{kind}{tuple(self.weight.shape)}, kernel 5, stride {self.stride}, padding 2 [HIP]"


class LinearParams(nn.Module):
    """nn.Linear's parameters / init, forward on the HIP GEMM."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F_hip.linear(x, self.weight, self.bias)

    def extra_repr(self) -> str:
        return f"in={self.in_features}, out={self.out_features}, bias={self.bias is not None} [HIP]"


class BatchNormAct(nn.Module):
    """BatchNorm2d/1d state (weight, bias, running_mean, running_var, num_batches_tracked) with a
    fused activation; train()/eval() switch batch vs running statistics like torch."""

    def __init__(self, num_features: int, momentum: float = 0.1, eps: float = 1e-5, act: Optional[str] = "relu",
                 slope: float = 0.0):
        super().__init__()
        self.num_features, self.momentum, self.eps, self.act, self.slope = num_features, momentum, eps, act, slope
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            self.num_batches_tracked.add_(1)
        return F_hip.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                    self.momentum, self.eps, self.act, self.slope)

    def extra_repr(self) -> str:
        return f"{self.num_features}, momentum={self.momentum}, eps={self.eps}, act={self.act} [HIP]"


class EncoderBlock(nn.Module):
    """conv5x5 s2 p2 (no bias) -> BatchNorm2d(momentum=0.9) -> ReLU."""

    def __init__(self, channel_in, channel_out):
        super().__init__()
        self.conv = Conv5x5Params(channel_out, channel_in, bias=False, stride=2, transposed=False)
        self.bn = BatchNormAct(channel_out, momentum=BN_MOMENTUM, act="relu")

    def forward(self, ten, out=False, t=False):
        ten = self.conv(ten)
        if out:
            return self.bn(ten), ten  # the pre-BN conv output is the discriminator's tap
        return self.bn(ten)


class DecoderBlock(nn.Module):
    """convT5x5 s2 p2 op1 (no bias) -> BatchNorm2d(momentum=0.9) -> ReLU."""

    def __init__(self, channel_in, channel_out):
        super().__init__()
        self.conv = Conv5x5Params(channel_in, channel_out, bias=False, stride=2, transposed=True)
        self.bn = BatchNormAct(channel_out, momentum=BN_MOMENTUM, act="relu")

    def forward(self, ten):
        return self.bn(self.conv(ten))


class Encoder(nn.Module):
    def __init__(self, channel_in=3, z_size=128, iter_level=3):
        super().__init__()
        self.size = channel_in
        layers = []
        for i in range(iter_level):
            cout = 64 if i == 0 else self.size * 2
            layers.append(EncoderBlock(channel_in=self.size, channel_out=cout))
            self.size = cout
        self.conv = nn.Sequential(*layers)
        # fc.0 / fc.1 carry the reference's keys; BN1d and ReLU run as one fused kernel
        self.fc = nn.Sequential(LinearParams(8 * 8 * self.size, 1024, bias=False),
                                BatchNormAct(1024, momentum=BN_MOMENTUM, act="relu"))
        self.l_mu = LinearParams(1024, z_size)
        self.l_var = LinearParams(1024, z_size)

    def forward(self, ten):
        ten = self.conv(ten)
        ten = F_hip.flatten_nchw(ten)
        ten = self.fc(ten)
        return self.l_mu(ten), self.l_var(ten)


class _SigmoidConv(nn.Sequential):
    """Container whose child "0" holds the final conv's weight/bias (key ``conv.<L>.0.*``);
    conv + bias + sigmoid run as one kernel (sigmoid in the MFMA epilogue)."""

    def forward(self, x):
        return self[0](x, act="sigmoid")


class Decoder(nn.Module):
    def __init__(self, z_size, size, channel_out=3, iter_level=3):
        super().__init__()
        self.fc = nn.Sequential(LinearParams(z_size, 8 * 8 * size, bias=False),
                                BatchNormAct(8 * 8 * size, momentum=BN_MOMENTUM, act="relu"))
        self.size = size
        self._c0 = size
        layers = [DecoderBlock(channel_in=self.size, channel_out=self.size)]
        for _ in range(iter_level - 1):
            layers.append(DecoderBlock(channel_in=self.size, channel_out=self.size // 2))
            self.size = self.size // 2
        layers.append(_SigmoidConv(Conv5x5Params(channel_out, self.size, bias=True, stride=1, transposed=False)))
        self.conv = nn.Sequential(*layers)

    def forward(self, ten):
        ten = self.fc(ten)
        ten = F_hip.unflatten_nchw(ten, self._c0, 8, 8)
        return self.conv(ten)


def reparameterize(mu, logvar, *, eps=None, generator=None):
    """VaeGan.reparameterize (models/networks.py:228-231) with an injectable eps."""
    return F_hip.reparameterize(mu, logvar, eps=eps, generator=generator)


def init_parameters(*modules: nn.Module) -> None:
    """VaeGan.init_parameters (models/networks.py:214-226): every conv / transposed-conv / linear
    weight ~ U(+-1/sqrt(3 * prod(shape[1:]))), biases 0, in ``modules()`` order."""
    for root in modules:
        for m in root.modules():
            if isinstance(m, (Conv5x5Params, LinearParams)):
                if m.weight is not None and m.weight.requires_grad:
                    scale = 1.0 / math.sqrt(float(torch.Size(m.weight.shape[1:]).numel()))
                    scale /= math.sqrt(3)
                    nn.init.uniform_(m.weight, -scale, scale)
                if m.bias is not None and m.bias.requires_grad:
                    nn.init.constant_(m.bias, 0.0)


class DirectDecoder(nn.Module):
    """models/networks.py:118-148: a chain of bias-ed Linear layers WITHOUT activations (head 4, r_fc 2, xy_fc 2);
    returns cat([r, xy], -1) of shape (B, 3)."""

    def __init__(self, z_size, num_of_param=3):
        super().__init__()
        self.head = nn.Sequential(LinearParams(z_size, 512), LinearParams(512, 256), LinearParams(256, 128),
                                  LinearParams(128, 64))
        self.r_fc = nn.Sequential(LinearParams(64, 32), LinearParams(32, 1))
        self.xy_fc = nn.Sequential(LinearParams(64, 32), LinearParams(32, 2))

    def forward(self, ten):
        ten = self.head(ten)
        return torch.cat([self.r_fc(ten), self.xy_fc(ten)], dim=-1)


class _ReluConv(nn.Sequential):
    """Child "0" holds Conv2d(k5, s1, p2, bias)'s parameters (key ``conv.0.0.*``); child "1" of the reference is the
    parameter-free ReLU.  The conv runs on the MFMA kernel, bias in its epilogue, ReLU as one elementwise pass."""

    def forward(self, x):
        conv = self[0]
        if F_hip.first_conv_s1_edge(conv.weight, conv.stride):       # bias + ReLU in the convolution's epilogue
            return conv(x, act="relu")
        return F_hip.activation(conv(x), "relu")


class Discriminator(nn.Module):
    """models/networks.py:151-198.  ``forward(orig, predicted, sampled, mode)`` concatenates the three batches,
    mode "REC" returns the flattened PRE-BatchNorm output of conv[recon_level] (NCHW order), mode "GAN" the
    sigmoid score (3B, 1).  The attribute is spelled ``recon_levl`` as in the reference."""

    def __init__(self, channel_in=3, recon_level=3, iter_level=3):
        super().__init__()
        self.size = channel_in
        self.recon_levl = recon_level
        self.conv = nn.ModuleList()
        self.conv.append(_ReluConv(Conv5x5Params(32, self.size, bias=True, stride=1, transposed=False)))
        self.size = 32
        channel_out = self.size * 2
        for _ in range(iter_level):
            self.conv.append(EncoderBlock(channel_in=self.size, channel_out=channel_out))
            self.size = channel_out
            channel_out *= 2
        # fc.0 / fc.1 / fc.3 carry the reference's keys; BN1d + ReLU (fc.1, fc.2) run as one fused kernel
        self.fc = nn.Sequential(LinearParams(8 * 8 * self.size, 512, bias=False),
                                BatchNormAct(512, momentum=BN_MOMENTUM, act="relu"),
                                nn.Identity(),
                                LinearParams(512, 1))

    def forward(self, ten_orig, ten_predicted, ten_sampled, mode="REC"):
        ten = torch.cat((ten_orig, ten_predicted, ten_sampled), 0)
        if mode == "REC":
            for i, lay in enumerate(self.conv):
                if i == self.recon_levl:
                    ten, layer_ten = lay(ten, True)
                    return F_hip.flatten_nchw(layer_ten)
                ten = lay(ten)
            return None  # recon_level beyond the stack: the reference falls off the loop too
        for lay in self.conv:
            ten = lay(ten)
        ten = F_hip.flatten_nchw(ten)
        return F_hip.activation(self.fc(ten), "sigmoid")

    def forward_rec_and_gan(self, ten_orig, ten_predicted, ten_sampled):
        """Both results of ``forward(..., "REC")`` and ``forward(..., "GAN")`` from ONE pass over the conv stack.
        In training mode the two reference calls compute identical activations (BatchNorm normalises with the batch
        statistics) and differ only in side effects: the blocks up to ``recon_levl`` update their running statistics
        twice.  That second update is replayed here from the buffers before / after the single pass
        (rm2 = (1-m) rm1 + (rm1 - (1-m) rm0)), and ``num_batches_tracked`` is advanced twice."""
        ten = torch.cat((ten_orig, ten_predicted, ten_sampled), 0)
        layer_ten = None
        replay = []
        for i, lay in enumerate(self.conv):
            if i == self.recon_levl:
                bn = lay.bn
                if self.training:
                    replay.append((bn, bn.running_mean.clone(), bn.running_var.clone()))
                ten, tap = lay(ten, True)
                layer_ten = F_hip.flatten_nchw(tap)
            else:
                if self.training and 0 < i < self.recon_levl:
                    replay.append((lay.bn, lay.bn.running_mean.clone(), lay.bn.running_var.clone()))
                ten = lay(ten)
        with torch.no_grad():
            for bn, rm0, rv0 in replay:
                k = 1.0 - bn.momentum
                bn.running_mean.mul_(2.0 - bn.momentum).sub_(rm0, alpha=k)        # (1-m) rm1 + rm1 - (1-m) rm0
                bn.running_var.mul_(2.0 - bn.momentum).sub_(rv0, alpha=k)
                bn.num_batches_tracked.add_(1)
        ten = F_hip.flatten_nchw(ten)
        return layer_ten, F_hip.activation(self.fc(ten), "sigmoid")


class VaeGan(nn.Module):
    """models/networks.py:201-281: Encoder(channel_in=1) / Decoder(channel_out=1) / Discriminator / DirectDecoder.

    ``forward`` follows the reference (training: x_tilde, disc_class, disc_layer, mus, log_variances, params;
    eval: x_p for ``x is None`` else (x_tilde, params)); the two random draws may be injected with ``eps=`` (the
    reparameterisation noise) and ``z_p=`` (the prior sample) for parity tests, and are taken from torch's device
    RNG otherwise."""

    def __init__(self, img_size, z_size=128, num_of_param=3):
        super().__init__()
        self.iter_level = int(math.log2(img_size // 8))
        self.z_size = z_size
        self.encoder = Encoder(channel_in=1, z_size=self.z_size, iter_level=self.iter_level)
        self.decoder = Decoder(z_size=self.z_size, size=self.encoder.size, channel_out=1, iter_level=self.iter_level)
        self.discriminator = Discriminator(channel_in=1, recon_level=self.iter_level, iter_level=self.iter_level)
        self.param_encoder = DirectDecoder(z_size, num_of_param=num_of_param)
        # the reference runs the discriminator twice per step ("REC" then "GAN", models/networks.py:244-245); one pass
        # gives both (Discriminator.forward_rec_and_gan).  Set to False for the literal two-call sequence.
        self.single_pass_discriminator = True
        self.init_parameters()

    def init_parameters(self):
        init_parameters(self)

    def reparameterize(self, mu, logvar, eps=None):
        return reparameterize(mu, logvar, eps=eps)

    def forward(self, x, gen_size=10, *, eps=None, z_p=None):
        dev = next(self.parameters()).device
        if self.training:
            mus, log_variances = self.encoder(x)
            z = self.reparameterize(mus, log_variances, eps)
            x_tilde = self.decoder(z)
            params = self.param_encoder(z)
            if z_p is None:
                z_p = torch.randn(len(x), self.z_size, device=dev)
            z_p = z_p.detach().requires_grad_(True)
            x_p = self.decoder(z_p)
            if self.single_pass_discriminator and self.discriminator.recon_levl < len(self.discriminator.conv):
                disc_layer, disc_class = self.discriminator.forward_rec_and_gan(x, x_tilde, x_p)
            else:
                disc_layer = self.discriminator(x, x_tilde, x_p, "REC")
                disc_class = self.discriminator(x, x_tilde, x_p, "GAN")
            return x_tilde, disc_class, disc_layer, mus, log_variances, params
        if x is None:
            if z_p is None:
                z_p = torch.randn(gen_size, self.z_size, device=dev)
            return self.decoder(z_p)
        mus, log_variances = self.encoder(x)
        z = self.reparameterize(mus, log_variances, eps)
        return self.decoder(z), self.param_encoder(z)

    @staticmethod
    def backward_all(*losses, retain_graph: bool = False):
        """One traversal of the graph for train.py:69-73's five ``backward(retain_graph=True)`` calls: gradients
        accumulate, so the result is the gradient of the SUM of the losses (identical up to fp32 summation order)
        at a fifth of the backward kernel time."""
        total = losses[0]
        for l in losses[1:]:
            total = total + l
        total.backward(retain_graph=retain_graph)

    @staticmethod
    def loss(x, x_tilde, disc_layer_original, disc_layer_predicted, disc_layer_sampled, disc_class_original,
             disc_class_predicted, disc_class_sampled, mus, variances, targets, params):
        """models/networks.py:265-281, same 7 return values.  The per-pixel / per-feature terms (nle, mse) and the
        KL run on HIP kernels; the terms over O(B) elements (three -log(p + 1e-3) and the smooth-L1 over (B, 3))
        are torch elementwise ops on the same device."""
        nle = F_hip.half_sq_diff(x.reshape(len(x), -1), x_tilde.reshape(len(x_tilde), -1))
        kl = F_hip.kl_divergence(mus, variances)
        mse = F_hip.half_sq_diff_rowsum(disc_layer_original, disc_layer_predicted)
        bce_dis_original = -torch.log(disc_class_original + 1e-3)
        bce_dis_predicted = -torch.log(1 - disc_class_predicted + 1e-3)
        bce_dis_sampled = -torch.log(1 - disc_class_sampled + 1e-3)
        l1_enc_param = torch.nn.functional.smooth_l1_loss(targets, params, reduction="sum") / x.size(0)
        return nle, kl, mse, bce_dis_original, bce_dis_predicted, bce_dis_sampled, l1_enc_param


class VAE(nn.Module):
    """Encoder -> reparameterize -> Decoder, the plain-VAE composition of SURVEY.md 3.3
    (``iter_level = int(log2(img_size // 8))``, models/networks.py:204)."""

    def __init__(self, img_size: int, z_size: int = 128, channels: int = 3, init_rule: bool = True):
        super().__init__()
        self.iter_level = int(math.log2(img_size // 8))
        self.z_size = z_size
        self.encoder = Encoder(channel_in=channels, z_size=z_size, iter_level=self.iter_level)
        self.decoder = Decoder(z_size=z_size, size=self.encoder.size, channel_out=channels, iter_level=self.iter_level)
        if init_rule:
            init_parameters(self)

    def forward(self, x, eps=None):
        mu, logvar = self.encoder(x)
        z = reparameterize(mu, logvar, eps=eps)
        return self.decoder(z), mu, logvar

    loss = staticmethod(F_hip.vae_loss)
