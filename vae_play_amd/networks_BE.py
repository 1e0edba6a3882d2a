"""Drop-in heads of kungyao/vae-play's ``models/networks_BE.py`` on the HIP back end.

  FeatureNet.aux_convs  <- models/networks_BE.py:18-26   (1x1 / 3x3 Conv2d + BatchNorm + ReLU pairs, halving channels)
  MaskNet / EdgeNet     <- models/networks_BE.py:39-66   (two ``Up`` blocks with coordinate channels, three bias-only 3x3 convs)
  ComposeNet            <- models/networks_BE.py:68-89   ({"edges", "masks"} from one feature map)
  be_loss               <- train_BE.py:58-59 + tools/ops.py:12-19
  initialize_model      <- tools/ops.py:216-229

The reference's backbone is torchvision's ``resnet_fpn_backbone('resnet50', True)`` (third-party code plus a
pretrained-weight download): out of scope.  ``FeatureNet`` therefore takes the backbone as an
argument -- any module with ``.out_channels`` whose output is a tensor or a dict holding key "0" -- and owns only
``aux_convs``; ``state_dict`` keys of everything below it equal the reference's (``aux_convs.<i>.conv.*``,
``mask_net.conv1.conv.0.conv.0.weight``, ...).  Every convolution, normalisation, resize and the loss run as HIP
kernels through ``blocks.py`` / ``functional.py``.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import functional as F_hip
from .blocks import AddCoords, Conv2d, ConvKParams, Up
from .networks import BatchNormAct, Conv5x5Params, LinearParams


class FeatureNet(nn.Module):
    """models/networks_BE.py:13-37 with the backbone injected (``backbone=None``: the input already is the
    stride-4 feature map, ``in_channels`` wide)."""

    def __init__(self, backbone: nn.Module = None, in_channels: int = 256, target_out_channels: int = 32):
        super().__init__()
        if backbone is not None:
            self.backbone = backbone
            in_channels = backbone.out_channels
        else:
            self.backbone = None
        layers = []
        repeat_num = int(math.log2(in_channels // target_out_channels))
        for _ in range(repeat_num):
            layers.append(Conv2d(in_channels, in_channels // 2, 1, stride=1, bn="batch"))
            layers.append(Conv2d(in_channels // 2, in_channels // 2, 3, stride=1, bn="batch"))
            in_channels = in_channels // 2
        self.aux_convs = nn.Sequential(*layers)
        self.out_channels = target_out_channels

    def forward(self, x):
        if self.backbone is not None:
            x = self.backbone(x)
            if isinstance(x, dict):
                x = x["0"]
        return self.aux_convs(x)


class MaskNet(nn.Module):
    def __init__(self, in_channel):
        super().__init__()
        self.conv1 = Up(in_channel, in_channel // 4, if_add_coord=True)
        self.conv2 = Up(in_channel // 4, in_channel // 8, if_add_coord=True)
        self.out_channels = 1
        self.predictor = nn.Sequential(
            Conv2d(in_channel // 8, in_channel // 4, 3, stride=1, bn=None, activate=None),
            Conv2d(in_channel // 4, in_channel // 8, 3, stride=1, bn=None, activate=None),
            Conv2d(in_channel // 8, self.out_channels, 3, stride=1, bn=None, activate=None))

    def forward(self, x):
        return self.predictor(self.conv2(self.conv1(x)))


class EdgeNet(MaskNet):
    pass


class ComposeNet(nn.Module):
    def __init__(self, feature_net: nn.Module = None):
        super().__init__()
        self.feature_net = feature_net if feature_net is not None else FeatureNet()
        self.mask_net = MaskNet(self.feature_net.out_channels)
        self.edge_net = EdgeNet(self.feature_net.out_channels)
        self.add_coords = AddCoords()

    def forward(self, x):
        feature = self.feature_net(x)
        return {"edges": self.edge_net(feature), "masks": self.mask_net(feature)}


def be_loss(pred_logits, targets, bce_weight: float = 0.5, smooth: float = 1.0):
    """``0.5 * F.binary_cross_entropy_with_logits(pred, t) + compute_dice_loss(pred.sigmoid(), t)`` (train_BE.py:58-59)."""
    return F_hip.be_loss(pred_logits, targets, bce_weight, smooth)


def initialize_model(model: nn.Module) -> nn.Module:
    """tools/ops.py:216-229 on the drop-in parameter holders: conv weights kaiming_uniform_(fan_in, relu), conv/linear
    biases 0, BatchNorm 1/0, linear weights kaiming_uniform_(a=sqrt(5)); RNG is consumed in ``modules()`` order."""
    for m in model.modules():
        if isinstance(m, (ConvKParams, Conv5x5Params)):
            nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, BatchNormAct):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, LinearParams):
            nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
    return model
