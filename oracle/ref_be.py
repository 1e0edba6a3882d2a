"""CPU oracle for the networks_BE heads and the BE loss -- TEST INFRASTRUCTURE ONLY.

Functional restatement (plain torch on CPU, parameters in a dict keyed like the reference's ``state_dict``) of

  * ``FeatureNet.aux_convs``         models/networks_BE.py:18-26
  * ``MaskNet`` / ``EdgeNet``        models/networks_BE.py:39-66
  * the loss of train_BE.py:58-59    0.5 * BCEWithLogits + ``compute_dice_loss`` (tools/ops.py:12-19)
  * ``initialize_model``             tools/ops.py:216-229

Pinning.  ``models/networks_BE.py`` imports torchvision and ``tools/ops.py`` imports cv2; neither is installed here,
so those two files cannot be imported.  The heads are built ONLY from ``models/blocks.py`` classes, which do import:
oracle/gen_golden_be.py composes the reference's own ``Up`` / ``Conv2d`` modules in the order of
models/networks_BE.py:42-57 and :20-25 and asserts bit equality with this file before writing
tests/golden/be_*.npz -> the heads are PINNED.  ``F.binary_cross_entropy_with_logits`` is torch's own function
(pinned by construction).  ``compute_dice_loss`` and ``initialize_model`` are restated from the source text and
checked only against torch primitives: PARITY UNPINNED for those two functions (no fixture or test of the reference
covers them and the file that defines them cannot be executed here).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from . import ref_cpu as O

Params = Dict[str, torch.Tensor]


def aux_channels(in_channels: int = 256, target: int = 32):
    out = []
    for _ in range(int(math.log2(in_channels // target))):
        out.append((in_channels, in_channels // 2, 1))
        out.append((in_channels // 2, in_channels // 2, 3))
        in_channels //= 2
    return out


def aux_convs_forward(p: Params, x: torch.Tensor, in_channels: int = 256, target: int = 32, training: bool = True,
                      prefix: str = "aux_convs.") -> torch.Tensor:
    """models/networks_BE.py:18-26,36: Conv2d(c, c/2, 1, bn='batch') + Conv2d(c/2, c/2, 3, bn='batch') until ``target``."""
    for i, (_, _, k) in enumerate(aux_channels(in_channels, target)):
        x = O.blocks_conv2d(p, f"{prefix}{i}.", x, k, 1, "batch", "relu", training)
    return x


def masknet_forward(p: Params, x: torch.Tensor, training: bool = True, prefix: str = "") -> torch.Tensor:
    """models/networks_BE.py:54-58: Up(c, c/4, coord) -> Up(c/4, c/8, coord) -> three bias-only 3x3 convs (no norm, no act)."""
    x = O.blocks_up(p, prefix + "conv1.", x, True, training)
    x = O.blocks_up(p, prefix + "conv2.", x, True, training)
    for i in range(3):
        x = O.blocks_conv2d(p, f"{prefix}predictor.{i}.", x, 3, 1, None, None, training)
    return x


def dice_loss(inputs: torch.Tensor, targets: torch.Tensor, smooth: float = 1.0) -> torch.Tensor:
    """tools/ops.py:12-19 (restated; parity unpinned, see the header)."""
    nums = inputs.size(0)
    iflat = inputs.view(nums, -1)
    tflat = targets.view(nums, -1)
    intersection = iflat * tflat
    score = (2.0 * intersection.sum(1) + smooth) / (iflat.sum(1) + tflat.sum(1) + smooth)
    return 1 - score.sum() / nums


def be_loss(pred_logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """train_BE.py:58-59 for one head."""
    return 0.5 * F.binary_cross_entropy_with_logits(pred_logits, targets) + dice_loss(pred_logits.sigmoid(), targets)


def init_rule(shape, kind: str, g: torch.Generator) -> torch.Tensor:
    """tools/ops.py:216-229 per tensor (restated; parity unpinned): conv weight kaiming_uniform_(fan_in, relu) =
    U(+-sqrt(6 / fan_in)); biases 0; BatchNorm 1 / 0."""
    if kind == "conv_w":
        fan_in = shape[1] * shape[2] * shape[3]
        bound = math.sqrt(2.0) * math.sqrt(3.0 / fan_in)
        return (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
    if kind in ("bn_w", "bn_rv"):
        return torch.ones(shape)
    if kind == "bn_nbt":
        return torch.zeros((), dtype=torch.long)
    return torch.zeros(shape)


def seeded_weights(sd, seed):
    """Deterministic weights for a blocks-built state_dict, regenerated identically by the tests (key order)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point or "running" in k:
            out[k] = v.clone()
        elif v.dim() == 4:
            out[k] = init_rule(tuple(v.shape), "conv_w", g)
        elif k.endswith("conv.1.weight"):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        else:
            out[k] = torch.randn(v.shape, generator=g) * 0.1
    return out


def heads_step(p: Params, opt, feature: torch.Tensor, bimgs: torch.Tensor, eimgs: torch.Tensor):
    """train_BE.py:54-64 below the feature map: both heads, both losses, backward, optimiser step."""
    for n in O.trainable_names(p):
        p[n].grad = None
    masks = masknet_forward(p, feature, True, "mask_net.")
    edges = masknet_forward(p, feature, True, "edge_net.")
    loss_edge = be_loss(edges, eimgs)
    loss_mask = be_loss(masks, bimgs)
    losses = loss_edge + loss_mask
    losses.backward()
    if opt is not None:
        opt.step()
    return {"masks": masks.detach(), "edges": edges.detach(), "loss_edge": loss_edge.detach(), "loss_mask": loss_mask.detach()}


# --------------------------------------------------------------------------------------------------------------------
# models/networks_BE_GAN.py (the discriminator built on blocks) and the losses of train_BE_GAN.py:131-160.
# Pinning: as for the heads above -- the networks are rebuilt from the reference's own blocks classes by
# oracle/gen_golden_be.py (networks_BE_GAN.py imports torchvision); edge_loss / dice_loss are restated (tools/ops.py
# needs cv2): parity unpinned for those two.
# --------------------------------------------------------------------------------------------------------------------
def mask_mapper_forward(p: Params, prefix: str, x: torch.Tensor, m: torch.Tensor, in_size: int, training: bool = True):
    """models/networks_BE_GAN.py:101-112."""
    repeat = int(math.log2(in_size // 16)) - 2
    x = torch.cat([x, m], dim=1)
    x = O.blocks_conv2d(p, prefix + "convs.0.", x, 3, 2, None, "lrelu", training)
    x = O.blocks_conv2d(p, prefix + "convs.1.", x, 3, 2, None, "lrelu", training)
    feats = []
    for i in range(repeat):
        x = O.blocks_conv2d(p, f"{prefix}feat_modules.{i}.0.", x, 3, 2, "batch", "lrelu", training)
        x = O.blocks_conv2d(p, f"{prefix}feat_modules.{i}.1.", x, 3, 1, "batch", "lrelu", training)
        feats.append(x.reshape(x.size(0), -1) * (i // 2 + 1))
    feats = torch.cat(feats, dim=1)
    x = O.blocks_conv2d(p, prefix + "pooler.0.", x, 1, 1, None, None, training)
    x = F.adaptive_avg_pool2d(x, (1, 1))
    return x.reshape(x.size(0), -1), feats


def gan_discriminator_forward(p: Params, x: torch.Tensor, m1: torch.Tensor, m2: torch.Tensor, in_size: int, training: bool = True,
                              prefix: str = ""):
    """models/networks_BE_GAN.py:129-139."""
    x = x[:, 0, :, :].reshape(x.size(0), 1, x.size(2), x.size(3))
    a, fa = mask_mapper_forward(p, prefix + "content_disc.", x, m1, in_size, training)
    b, fb = mask_mapper_forward(p, prefix + "boundary_disc.", x, m2, in_size, training)
    h = torch.cat([a, b], dim=1)
    h = O.blocks_linear(p, prefix + "predictor.0.", h, "lrelu")
    h = O.blocks_linear(p, prefix + "predictor.1.", h, "lrelu")
    return O.blocks_linear(p, prefix + "predictor.2.", h, None), torch.cat([fa, fb], dim=1)


def edge_loss(mask_probs: torch.Tensor, mask_targets: torch.Tensor) -> torch.Tensor:
    """tools/ops.py:187-215 (restated; parity unpinned)."""
    k = torch.full((3, 3), -1.0)
    k[1, 1] = 8.0
    k = (k / 8).reshape(1, 1, 3, 3)
    return dice_loss(F.conv2d(mask_probs, k, padding=1).abs(), F.conv2d(mask_targets, k, padding=1).abs())


# --------------------------------------------------------------------------------------------------------------------
# train_BE_GAN.py:131-165 -- the two-phase loop body (discriminator step, generator step), below the backbone.
# Pinning: the NETWORKS are the blocks compositions above (pinned through the reference's own blocks by
# oracle/gen_golden_be.py: be_gan_train128_b2 runs this function next to the same loop body written with the reference's
# blocks modules, torch's own F.cross_entropy / F.binary_cross_entropy_with_logits and torch.optim.Adam, and asserts bit
# equality of the seven losses and of every parameter after each iteration).  compute_dice_loss / edge_loss are the
# restatements above on BOTH sides: parity unpinned for those two functions (tools/ops.py needs cv2).
# --------------------------------------------------------------------------------------------------------------------
def gan_generator_forward(p: Params, feature: torch.Tensor, feat_channels: int, training: bool = True):
    """models/networks_BE_GAN.py:60-72 below the backbone: aux_convs down to 64 channels, then the mask and the edge head."""
    h = aux_convs_forward(p, feature, feat_channels, 64, training, "aux_convs.")
    return {"masks": masknet_forward(p, h, training, "mask_net."), "edges": masknet_forward(p, h, training, "edge_net.")}


def gan_make_optimizers(pg: Params, pd: Params, lr: float = 1e-4):
    """train_BE_GAN.py:236-237: Adam(G, lr, betas (0.5, 0.999)), Adam(D, 0.1 lr, betas (0.5, 0.999))."""
    return (torch.optim.Adam([pg[n] for n in O.trainable_names(pg)], lr=lr, betas=(0.5, 0.999)),
            torch.optim.Adam([pd[n] for n in O.trainable_names(pd)], lr=lr * 0.1, betas=(0.5, 0.999)))


def gan_train_iteration(pg: Params, pd: Params, g_opt, d_opt, feature, imgs, bimgs, eimgs, labels, in_size: int, feat_channels: int):
    """One iteration of train_BE_GAN.py:131-165 (``feature`` = the backbone's stride-4 map of ``imgs``: the backbone is out of
    scope).  Returns the seven scalars the reference logs (:167-175)."""
    # D (:131-144)
    with torch.no_grad():
        preds = gan_generator_forward(pg, feature, feat_channels, True)
        pred_masks, pred_edges = preds["masks"].sigmoid(), preds["edges"].sigmoid()
    d_real_type, d_real_feats = gan_discriminator_forward(pd, imgs, bimgs, eimgs, in_size, True)
    d_fake_type, d_fake_feats = gan_discriminator_forward(pd, imgs, pred_masks, pred_edges, in_size, True)
    d_adv_loss = 1 - torch.mean(torch.abs(d_fake_feats - d_real_feats))
    d_type_loss = F.cross_entropy(d_real_type, labels)
    d_losses = d_adv_loss + d_type_loss
    d_opt.zero_grad()
    d_losses.backward()
    d_opt.step()
    # G (:147-165)
    preds = gan_generator_forward(pg, feature, feat_channels, True)
    pred_masks, pred_edges = preds["masks"], preds["edges"]
    with torch.no_grad():
        _, g_real_feats = gan_discriminator_forward(pd, imgs, bimgs, eimgs, in_size, True)
    g_pred_type, g_pred_feats = gan_discriminator_forward(pd, imgs, pred_masks.sigmoid(), pred_edges.sigmoid(), in_size, True)
    loss_mask = be_loss(pred_masks, bimgs)
    loss_edge = be_loss(pred_edges, eimgs)
    g_adv_loss = torch.mean(torch.abs(g_pred_feats - g_real_feats))
    g_type_loss = F.cross_entropy(g_pred_type, labels)
    loss_cnt = edge_loss(pred_masks.sigmoid(), bimgs) + edge_loss(pred_edges.sigmoid(), eimgs)
    losses = loss_mask * 2 + loss_edge * 2 + g_adv_loss + g_type_loss + loss_cnt * 0.5
    g_opt.zero_grad()
    losses.backward()
    g_opt.step()
    return {"d_adv_loss": d_adv_loss.detach(), "d_type_loss": d_type_loss.detach(), "loss_edge": loss_edge.detach(),
            "loss_mask": loss_mask.detach(), "g_adv_loss": g_adv_loss.detach(), "g_type_loss": g_type_loss.detach(),
            "loss_cnt": loss_cnt.detach(), "masks": pred_masks.detach(), "edges": pred_edges.detach()}
