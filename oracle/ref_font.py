"""CPU oracle for the font U-Net and its AC-GAN discriminator -- TEST INFRASTRUCTURE ONLY.

Functional restatement (plain torch on CPU, parameters in a dict keyed like the reference's ``state_dict``) of
``models/networks_BE_font.py`` (EmbedingBlock :21-45, StyleEncodeBlock :47-67, ParameterEmbedingNet :69-83,
MaskNet/EdgeNet :85-123, ComposeNet :125-232, Classifier :234-263, Discriminator :265-274), of
``SelfAttentionBlock`` (models/blocks.py:66-96) and of the three optimiser phases of train_BE_font.py:97-170.

Pinning.  ``models/networks_BE_font.py`` imports ``turtle`` (tkinter) and torchvision, ``train_BE_font.py`` imports
cv2-dependent modules: none can be imported here.  Every layer of these networks is a ``models/blocks.py`` class, and
that file does import: oracle/gen_golden_font.py rebuilds the networks from the reference's OWN ``Conv2d`` / ``Linear`` /
``Up`` / ``SelfAttentionBlock`` objects under the reference's attribute names (so ``state_dict`` keys and the forward
data flow are those of networks_BE_font.py:138-232,236-274) and asserts bit equality with this file -> the networks are
PINNED through their blocks.  The loss composition of train_BE_font.py:109-162 is restated here from the source
text (torch's own BCE / cross-entropy / L1 + the dice restatement of oracle/ref_be.py): PARITY UNPINNED for that
composition and for the dice term.  The quirk at train_BE_font.py:142 (``loss_g_aux = loss_g_adv * 5``: the auxiliary
cross-entropy is computed and then overwritten) is reproduced as written.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from . import ref_be as BE
from . import ref_cpu as O

Params = Dict[str, torch.Tensor]
LABEL_EMBED = STYLE_EMBED = 256


def self_attention(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """models/blocks.py:77-96 (q, k, v are Conv2d blocks with their default ReLU)."""
    b, c, h, w = x.shape
    q = O.blocks_conv2d(p, prefix + "q.", x, 1, 1, None, "relu").view(b, -1, h * w).permute(0, 2, 1)
    k = O.blocks_conv2d(p, prefix + "k.", x, 1, 1, None, "relu").view(b, -1, h * w)
    att = torch.softmax(torch.bmm(q, k), dim=-1)
    v = O.blocks_conv2d(p, prefix + "v.", x, 1, 1, None, "relu").view(b, -1, h * w)
    out = torch.bmm(v, att.permute(0, 2, 1)).view(b, c, h, w)
    return p[prefix + "gamma"] * out + x


def embeding_block(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """models/networks_BE_font.py:39-45."""
    x = O.blocks_linear(p, prefix + "convs_first.0.", x, None)
    x = O.blocks_linear(p, prefix + "convs_first.1.", x, None)
    x = x.reshape(x.size(0), x.size(1), 1, 1)
    for i in range(3):
        x = self_attention(p, f"{prefix}attention.{i}.", x)
    x = x.reshape(x.size(0), -1)
    x = O.blocks_linear(p, prefix + "embeding.0.", x, "lrelu")
    return O.blocks_linear(p, prefix + "embeding.1.", x, "lrelu")


def style_encode_block(p: Params, prefix: str, x: torch.Tensor, in_size: int) -> torch.Tensor:
    """models/networks_BE_font.py:47-67: stride-2 3x3 convs with InstanceNorm, a 1x1 conv, global average."""
    n = int(math.log2(in_size)) - 3 + 1
    for i in range(n):
        x = O.blocks_conv2d(p, f"{prefix}convs.{i}.", x, 3, 2, "instance", "relu")
    x = O.blocks_conv2d(p, f"{prefix}convs.{n}.", x, 1, 1, "instance", "relu")
    return F.adaptive_avg_pool2d(x, (1, 1)).reshape(x.size(0), -1)


def font_masknet(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """models/networks_BE_font.py:101-117."""
    x = O.blocks_conv2d(p, prefix + "predictor.0.", x, 3, 1, "instance", "relu")
    x = O.blocks_conv2d(p, prefix + "predictor.1.", x, 3, 1, "instance", "relu")
    return O.blocks_conv2d(p, prefix + "predictor.2.", x, 3, 1, None, None)


def compose_forward(p: Params, x: torch.Tensor, y, in_size: int, training: bool = True, prefix: str = ""):
    """models/networks_BE_font.py:189-232."""
    repeat = int(math.log2(in_size // 4))
    if y is not None:
        y_cls = embeding_block(p, prefix + "embeding_block.label_encode_block.", y["cls"])
        y_sty = embeding_block(p, prefix + "embeding_block.style_encode_block.", y["cnt_style"])
    else:
        y_cls = style_encode_block(p, prefix + "style_encoder.label_encode_block.", x, in_size)
        y_sty = style_encode_block(p, prefix + "style_encoder.style_encode_block.", x, in_size)
    feats = []
    x = O.blocks_conv2d(p, prefix + "down.0.", x, 3, 1, "instance", "relu", training)
    feats.append(x)
    for i in range(1, repeat + 1):
        x = O.blocks_conv2d(p, f"{prefix}down.{i}.0.", x, 3, 2, "batch", "relu", training)
        x = O.blocks_conv2d(p, f"{prefix}down.{i}.1.", x, 3, 1, "instance", "relu", training)
        feats.append(x)
    b, c, h, w = x.shape
    x = torch.cat([x.reshape(b, -1), y_cls, y_sty], dim=1)
    x = O.blocks_linear(p, prefix + "relay_convs.0.", x, "relu")
    x = O.blocks_linear(p, prefix + "relay_convs.1.", x, "relu")
    x = x.reshape(b, c, h, w)
    for i in range(repeat):
        idx = repeat - 1 - i
        x_up = O.blocks_up(p, f"{prefix}up.{idx}.", x, False, training)
        x_skip = O.blocks_conv2d(p, f"{prefix}skip.{idx}.", feats[len(feats) - 2 - i], 3, 1, "instance", "relu", training)
        x = O.blocks_conv2d(p, f"{prefix}cat.{idx}.", torch.cat([x_up, x_skip], dim=1), 3, 1, "instance", "relu", training)
    return {"edges": font_masknet(p, prefix + "edge_net.", x), "masks": font_masknet(p, prefix + "mask_net.", x)}


def classifier_forward(p: Params, prefix: str, x: torch.Tensor, y, training: bool = True) -> torch.Tensor:
    """models/networks_BE_font.py:254-263."""
    x = O.blocks_conv2d(p, prefix + "conv_first.", x, 3, 2, "instance", "lrelu", training)
    for i, bn in enumerate(("instance", "instance", "batch", "batch")):
        x = O.blocks_conv2d(p, f"{prefix}backbone.{i}.", x, 3, 2, bn, "lrelu", training)
    x = x.reshape(x.size(0), -1)
    y_cls = embeding_block(p, prefix + "embeding_block.label_encode_block.", y["cls"])
    y_sty = embeding_block(p, prefix + "embeding_block.style_encode_block.", y["cnt_style"])
    x = torch.cat([x, y_cls, y_sty], dim=1)
    x = O.blocks_linear(p, prefix + "cls_convs.0.", x, "lrelu")
    x = O.blocks_linear(p, prefix + "cls_convs.1.", x, "lrelu")
    return O.blocks_linear(p, prefix + "cls_convs.2.", x, None)


def discriminator_forward(p: Params, x: torch.Tensor, y, training: bool = True, prefix: str = ""):
    """models/networks_BE_font.py:271-274."""
    return classifier_forward(p, prefix + "adv_convs.", x, y, training).sigmoid(), classifier_forward(p, prefix + "aux_convs.", x, y, training)


def seg_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return BE.be_loss(pred, target)


def _zero(p: Params, names):
    for n in names:
        p[n].grad = None


def train_iteration(pn: Params, pd: Params, opt, opt_disc, opt_style, imgs, masks, edge_masks, labels, y_map, in_size: int):
    """train_BE_font.py:97-170: discriminator phase, generator phase, style-encoder phase."""
    b = imgs.size(0)
    out = {}
    # --- D (train_BE_font.py:97-114)
    gt_merge = torch.cat([masks, edge_masks], dim=1)
    with torch.no_grad():
        preds = compose_forward(pn, imgs, y_map, in_size)
        pred_merge = torch.cat([preds["masks"], preds["edges"]], dim=1)
    d_gt_adv, d_gt_aux = discriminator_forward(pd, gt_merge, y_map)
    d_pred_adv, _ = discriminator_forward(pd, pred_merge, y_map)
    _zero(pd, O.trainable_names(pd))
    d_adv_real = F.binary_cross_entropy(d_gt_adv, torch.ones((b, 1)))
    d_aux_real = F.cross_entropy(d_gt_aux, labels)
    d_adv_fake = F.binary_cross_entropy(d_pred_adv, torch.zeros((b, 1)))
    d_loss = (d_adv_real + d_adv_fake) * 0.5 + d_aux_real
    d_loss.backward()
    opt_disc.step()
    out.update(d_adv_real=d_adv_real.detach(), d_aux_real=d_aux_real.detach(), d_adv_fake=d_adv_fake.detach())
    # --- G (train_BE_font.py:116-147)
    preds = compose_forward(pn, imgs, y_map, in_size)
    g_adv, g_aux = discriminator_forward(pd, torch.cat([preds["masks"], preds["edges"]], dim=1), y_map)
    _zero(pn, O.trainable_names(pn))
    loss_mask = seg_loss(preds["masks"], masks) * 10
    loss_edge = seg_loss(preds["edges"], edge_masks) * 10
    loss_g_adv = F.binary_cross_entropy(g_adv, torch.ones((b, 1))) * 2
    loss_g_aux = F.cross_entropy(g_aux, labels)
    loss_g_aux = loss_g_adv * 5                   # train_BE_font.py:142, as written
    (loss_edge + loss_mask + loss_g_adv + loss_g_aux).backward()
    opt.step()
    out.update(loss_mask=loss_mask.detach(), loss_edge=loss_edge.detach(), loss_g_adv=loss_g_adv.detach(),
               masks=preds["masks"].detach(), edges=preds["edges"].detach())
    # --- style encoder (train_BE_font.py:149-164)
    with torch.no_grad():
        ref = compose_forward(pn, imgs, y_map, in_size)
    preds_ = compose_forward(pn, imgs, None, in_size)
    _zero(pn, O.trainable_names(pn))
    loss_embed = (F.l1_loss(preds_["masks"], ref["masks"]) + F.l1_loss(preds_["edges"], ref["edges"])) * 2.0
    (seg_loss(preds_["masks"], masks) + seg_loss(preds_["edges"], edge_masks) + loss_embed).backward()
    opt_style.step()
    out.update(loss_embed=loss_embed.detach())
    return out


def style_encoder_names(pn: Params):
    return [n for n in O.trainable_names(pn) if n.startswith("style_encoder.")]


def seeded_weights(sd, seed: int):
    """Deterministic weights for a blocks-built state_dict, regenerated identically by the tests (key order):
    kaiming-uniform-scaled conv / linear weights, small biases, BatchNorm scales in (0.5, 1.5), attention gammas 0.3
    (the reference initialises gamma to 0, which would leave the attention path untested)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point or "running" in k:
            out[k] = v.clone()
        elif k.endswith("gamma"):
            out[k] = torch.full_like(v, 0.3)
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            out[k] = (torch.rand(v.shape, generator=g) * 2.0 - 1.0) * math.sqrt(3.0 / fan_in)
        elif k.endswith("conv.1.weight"):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        else:
            out[k] = torch.randn(v.shape, generator=g) * 0.05
    return out


def synthetic_batch(B: int, S: int):
    g = torch.Generator().manual_seed(8642)
    imgs = torch.rand(B, 3, S, S, generator=g)
    masks = (torch.rand(B, 1, S, S, generator=g) > 0.5).float()
    edges = (torch.rand(B, 1, S, S, generator=g) > 0.8).float()
    labels = torch.randint(0, 143, (B,), generator=g)
    cls = torch.zeros(B, 143)
    cls[torch.arange(B), labels] = 1
    cnt_style = torch.rand(B, 5, generator=g)
    return imgs, masks, edges, labels, {"cls": cls, "cnt_style": cnt_style}
