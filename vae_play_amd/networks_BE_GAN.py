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


class ComposeNet(nn.Module):
    def __init__(self, in_channels, in_size, backbone: nn.Module = None, feature_channels: int = 256):
        super().__init__()
        target_out_channels = 64
        self.backbone = backbone
        c = backbone.out_channels if backbone is not None else feature_channels
        layers = []
        for _ in range(int(math.log2(c // target_out_channels))):
            layers.append(Conv2d(c, c // 2, 1, stride=1, bn="batch"))
            layers.append(Conv2d(c // 2, c // 2, 3, stride=1, bn="batch"))
            c //= 2
        self.aux_convs = nn.Sequential(*layers)
        self.mask_net = MaskNet(target_out_channels)
        self.edge_net = EdgeNet(target_out_channels)

    def forward(self, x):
        if self.backbone is not None:
            x = self.backbone(x)
            if isinstance(x, dict):
                x = x["0"]
        x = self.aux_convs(x)
        return {"masks": self.mask_net(x), "edges": self.edge_net(x)}


class MaskMapper(nn.Module):
    def __init__(self, in_channels, in_size, max_channel=128):
        super().__init__()
        min_in_size = 16
        repeat_num = int(math.log2(in_size // min_in_size)) - 2
        self.convs = nn.Sequential(Conv2d(in_channels, 16, 3, 2, bn=None, activate="lrelu"),
                                   Conv2d(16, 32, 3, 2, bn=None, activate="lrelu"))
        in_channels = 32
        out_channels = min(in_channels * 2, max_channel)
        self.feat_modules = nn.ModuleList()
        for _ in range(repeat_num):
            self.feat_modules.append(nn.Sequential(Conv2d(in_channels, out_channels, 3, 2, bn="batch", activate="lrelu"),
                                                   Conv2d(out_channels, out_channels, 3, 1, bn="batch", activate="lrelu")))
            in_channels = out_channels
            out_channels = min(in_channels * 2, max_channel)
        self.pooler = nn.Sequential(Conv2d(in_channels, max_channel, 1, 1, bn=None, activate=None), GlobalAvgPool())

    def forward(self, x, m):
        x = torch.cat([x, m], dim=1)
        x = self.convs(x)
        feat_list = []
        for idx, mod in enumerate(self.feat_modules):
            x = mod(x)
            feat_list.append(x.reshape(x.size(0), -1) * (idx // 2 + 1))
        feat_list = torch.cat(feat_list, dim=1)
        x = self.pooler(x)
        return x.reshape(x.size(0), -1), feat_list


class Discriminator(nn.Module):
    def __init__(self, in_channels, in_size, num_classes):
        super().__init__()
        max_channel = 64
        self.num_classes = num_classes
        self.content_disc = MaskMapper(2, in_size, max_channel=max_channel)
        self.boundary_disc = MaskMapper(2, in_size, max_channel=max_channel)
        self.predictor = nn.Sequential(Linear(max_channel * 2, max_channel * 2, bias=True, activate="lrelu"),
                                       Linear(max_channel * 2, max_channel, bias=True, activate="lrelu"),
                                       Linear(max_channel, num_classes, bias=False, activate=None))

    def forward(self, x, m1, m2):
        x = x[:, 0, :, :].reshape(x.size(0), 1, x.size(2), x.size(3))
        x_m1, feats_m1 = self.content_disc(x, m1)
        x_m2, feats_m2 = self.boundary_disc(x, m2)
        feats = torch.cat([feats_m1, feats_m2], dim=1)
        return self.predictor(torch.cat([x_m1, x_m2], dim=1)), feats
