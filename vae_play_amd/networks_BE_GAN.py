"""Drop-in classes for kungyao/vae-play's ``models/networks_BE_GAN.py`` (the "alt discriminator on blocks" of SURVEY.md
8f rank 1): the mask/edge generator heads and the feature-matching discriminator of train_BE_GAN.py, built from
``blocks.py`` so that everything runs on HIP kernels.

  MaskNet / EdgeNet   <- models/networks_BE_GAN.py:11-36 (identical to networks_BE's)
  ComposeNet          <- :38-72   (backbone injected: torchvision's ResNet-50-FPN is out of scope; 64-channel head)
  MaskMapper          <- :74-112  (3x3 stride-2 LeakyReLU convs, BatchNorm pairs, 1x1 conv + global average pool;
                                   returns the pooled vector and the concatenated, level-weighted feature maps)
  Discriminator       <- :114-139 (two MaskMappers on (image channel 0, mask) pairs + a three-layer Linear head)
Losses of train_BE_GAN.py:131-160: ``functional.l1_loss`` (feature matching), ``networks_BE.be_loss``,
``functional.edge_loss``; the cross-entropies over (B, classes) logits are O(B)-element torch ops.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .blocks import Conv2d, GlobalAvgPool, Linear
from .networks_BE import EdgeNet, MaskNet


def _halving(width: int, floor: int):
    """Channel widths of a stack that halves ``width`` until it reaches ``floor``: (c, c/2), (c/2, c/4), ..."""
    while width > floor:
        yield width, width // 2
        width //= 2


def _doubling(width: int, cap: int, steps: int):
    for _ in range(steps):
        nxt = min(2 * width, cap)
        yield width, nxt
        width = nxt


class ComposeNet(nn.Module):
    """Backbone feature map -> 64 channels by (1x1, 3x3) BatchNorm pairs that halve the width -> mask and edge heads."""

    _HEAD_WIDTH = 64

    def __init__(self, in_channels, in_size, backbone: nn.Module = None, feature_channels: int = 256):
        super().__init__()
        self.backbone = backbone
        width = backbone.out_channels if backbone is not None else feature_channels
        pairs = []
        for wide, narrow in _halving(width, self._HEAD_WIDTH):
            pairs += [Conv2d(wide, narrow, 1, stride=1, bn="batch"), Conv2d(narrow, narrow, 3, stride=1, bn="batch")]
        self.aux_convs = nn.Sequential(*pairs)
        self.mask_net = MaskNet(self._HEAD_WIDTH)
        self.edge_net = EdgeNet(self._HEAD_WIDTH)

    def forward(self, x):
        feat = x if self.backbone is None else self.backbone(x)
        if isinstance(feat, dict):               # an FPN hands back a dict of levels: the finest one is used
            feat = feat["0"]
        feat = self.aux_convs(feat)
        return {"masks": self.mask_net(feat), "edges": self.edge_net(feat)}


class MaskMapper(nn.Module):
    """(image channel, mask) -> a pooled 1x1-projected vector and the level-weighted, flattened feature maps of every
    (stride-2, stride-1) BatchNorm pair: the discriminator's feature-matching taps."""

    _FLOOR = 16

    def __init__(self, in_channels, in_size, max_channel=128):
        super().__init__()
        lrelu = dict(bn=None, activate="lrelu")
        self.convs = nn.Sequential(Conv2d(in_channels, 16, 3, 2, **lrelu), Conv2d(16, 32, 3, 2, **lrelu))
        self.feat_modules = nn.ModuleList()
        width = 32
        for cin, cout in _doubling(32, max_channel, int(math.log2(in_size // self._FLOOR)) - 2):
            self.feat_modules.append(nn.Sequential(Conv2d(cin, cout, 3, 2, bn="batch", activate="lrelu"),
                                                   Conv2d(cout, cout, 3, 1, bn="batch", activate="lrelu")))
            width = cout
        self.pooler = nn.Sequential(Conv2d(width, max_channel, 1, 1, bn=None, activate=None), GlobalAvgPool())

    def forward(self, x, m):
        h = self.convs(torch.cat([x, m], dim=1))
        taps = []
        for level, stage in enumerate(self.feat_modules):
            h = stage(h)
            taps.append(h.flatten(1) * (level // 2 + 1))
        return self.pooler(h).flatten(1), torch.cat(taps, dim=1)


class Discriminator(nn.Module):
    """Two MaskMappers -- (image, mask) "content" and (image, edge) "boundary" -- and a three-layer dense classifier over
    their pooled vectors; the concatenated taps are returned for the feature-matching loss."""

    _WIDTH = 64

    def __init__(self, in_channels, in_size, num_classes):
        super().__init__()
        w = self._WIDTH
        self.num_classes = num_classes
        self.content_disc = MaskMapper(2, in_size, max_channel=w)
        self.boundary_disc = MaskMapper(2, in_size, max_channel=w)
        self.predictor = nn.Sequential(Linear(2 * w, 2 * w, bias=True, activate="lrelu"), Linear(2 * w, w, bias=True, activate="lrelu"),
                                       Linear(w, num_classes, bias=False, activate=None))

    def forward(self, x, m1, m2):
        gray = x[:, :1]                           # the first image channel, kept as a 1-channel map
        vec1, taps1 = self.content_disc(gray, m1)
        vec2, taps2 = self.boundary_disc(gray, m2)
        return self.predictor(torch.cat([vec1, vec2], dim=1)), torch.cat([taps1, taps2], dim=1)
