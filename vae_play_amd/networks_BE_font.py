"""Drop-in classes for kungyao/vae-play's ``models/networks_BE_font.py``: the font U-Net
generator and its AC-GAN discriminator, built from ``blocks.py`` so that every convolution / normalisation / resize /
dense layer / attention runs on HIP kernels.  Constructor signatures, attribute names and ``state_dict`` keys equal
the reference's:

  EmbedingBlock          <- models/networks_BE_font.py:21-45     StyleEncodeBlock <- :47-67
  ParameterEmbedingNet   <- :69-83      MaskNet / EdgeNet (font variant: three 3x3 convs, InstanceNorm) <- :85-123
  ComposeNet(in_size)    <- :125-232    Classifier <- :234-263    Discriminator <- :265-274
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .blocks import Conv2d, GlobalAvgPool, Linear, SelfAttentionBlock, Up

LABEL_EMBED = 256
STYLE_EMBED = 256


class EmbedingBlock(nn.Module):
    def __init__(self, in_channels, out_channels, in_size):
        super().__init__()
        self.convs_first = nn.Sequential(Linear(in_channels, out_channels, activate=None),
                                         Linear(out_channels, out_channels, activate=None))
        self.attention = nn.Sequential(SelfAttentionBlock(out_channels), SelfAttentionBlock(out_channels),
                                       SelfAttentionBlock(out_channels))
        self.embeding = nn.Sequential(Linear(out_channels, out_channels, activate="lrelu"),
                                      Linear(out_channels, out_channels, activate="lrelu"))

    def forward(self, x):
        x = self.convs_first(x)
        x = x.reshape(x.size(0), x.size(1), 1, 1)
        x = self.attention(x)
        x = x.reshape(x.size(0), -1)
        return self.embeding(x)


class StyleEncodeBlock(nn.Module):
    def __init__(self, in_channels, out_channels, in_size):
        super().__init__()
        min_channel, max_channel = 64, out_channels
        repeat_num = int(math.log2(in_size)) - 3
        convs = [Conv2d(in_channels, min_channel, 3, stride=2, bn="instance")]
        in_channels = min_channel
        out_channels = min(in_channels * 2, max_channel)
        for _ in range(repeat_num):
            convs.append(Conv2d(in_channels, out_channels, 3, stride=2, bn="instance"))
            in_channels = out_channels
            out_channels = min(in_channels * 2, max_channel)
        convs.append(Conv2d(in_channels, max_channel, 1, stride=1, bn="instance"))
        convs.append(GlobalAvgPool())
        self.convs = nn.Sequential(*convs)

    def forward(self, x):
        x = self.convs(x)
        return x.reshape(x.size(0), -1)


class ParameterEmbedingNet(nn.Module):
    def __init__(self, encode_block, in_size, in_type=None):
        super().__init__()
        if in_type == "image":
            self.label_encode_block = encode_block(3, LABEL_EMBED, in_size)
            self.style_encode_block = encode_block(3, STYLE_EMBED, in_size)
        elif in_type == "embed":
            self.label_encode_block = encode_block(143, LABEL_EMBED, in_size)
            self.style_encode_block = encode_block(5, STYLE_EMBED, in_size)

    def forward(self, y_cls, y_cnt_style):
        return self.label_encode_block(y_cls), self.style_encode_block(y_cnt_style)


class MaskNet(nn.Module):
    def __init__(self, in_channel):
        super().__init__()
        self.out_channels = 1
        self.predictor = nn.Sequential(Conv2d(in_channel, in_channel, 3, stride=1, bn="instance"),
                                       Conv2d(in_channel, in_channel, 3, stride=1, bn="instance"),
                                       Conv2d(in_channel, self.out_channels, 3, stride=1, bn=None, activate=None))

    def forward(self, x):
        return self.predictor(x)


class EdgeNet(MaskNet):
    pass


class ComposeNet(nn.Module):
    def __init__(self, in_size):
        super().__init__()
        min_channel, max_channel = 64, 512
        min_in_size = 4
        repeat_num = int(math.log2(in_size // min_in_size))
        self.down = nn.ModuleList()
        self.down.append(Conv2d(3, min_channel, 3, stride=1, bn="instance"))
        in_channels = min_channel
        out_channels = min(in_channels * 2, max_channel)
        for _ in range(repeat_num):
            self.down.append(nn.Sequential(Conv2d(in_channels, out_channels, 3, stride=2, bn="batch"),
                                           Conv2d(out_channels, out_channels, 3, stride=1, bn="instance")))
            in_channels = out_channels
            out_channels = min(in_channels * 2, max_channel)
        self.embeding_block = ParameterEmbedingNet(EmbedingBlock, in_size, in_type="embed")
        self.style_encoder = ParameterEmbedingNet(StyleEncodeBlock, in_size, in_type="image")
        relay_in = in_channels * min_in_size * min_in_size
        self.relay_convs = nn.Sequential(Linear(relay_in + LABEL_EMBED + STYLE_EMBED, relay_in), Linear(relay_in, relay_in))
        self.up, self.skip, self.cat = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        in_channels = min_channel
        out_channels = min(in_channels * 2, max_channel)
        for _ in range(repeat_num):
            self.up.append(Up(out_channels, in_channels))
            self.skip.append(Conv2d(in_channels, in_channels, 3, stride=1, bn="instance"))
            self.cat.append(Conv2d(in_channels * 2, in_channels, 3, stride=1, bn="instance"))
            in_channels = out_channels
            out_channels = min(in_channels * 2, max_channel)
        self.mask_net = MaskNet(min_channel)
        self.edge_net = EdgeNet(min_channel)

    def forward(self, x, y=None):
        if y is not None:
            y_cls, y_cnt_style = self.embeding_block(y["cls"], y["cnt_style"])
        else:
            y_cls, y_cnt_style = self.style_encoder(x, x)
        down_feats = []
        for m in self.down:
            x = m(x)
            down_feats.append(x)
        b, c, h, w = x.shape
        x = x.reshape(b, -1)                      # (C, H, W) flatten order, as the reference's reshape of an NCHW tensor
        x = torch.cat([x, y_cls, y_cnt_style], dim=1)
        x = self.relay_convs(x)
        x = x.reshape(b, c, h, w)
        for i in range(len(self.up)):
            idx = len(self.up) - 1 - i
            x_up = self.up[idx](x)
            x_skip = self.skip[idx](down_feats[len(down_feats) - 2 - i])
            x = self.cat[idx](torch.cat([x_up, x_skip], dim=1))
        return {"edges": self.edge_net(x), "masks": self.mask_net(x)}


class Classifier(nn.Module):
    def __init__(self, in_size, in_channels, num_of_classes):
        super().__init__()
        self.conv_first = Conv2d(in_channels, 64, 3, stride=2, bn="instance", activate="lrelu")
        self.backbone = nn.Sequential(Conv2d(64, 128, 3, stride=2, bn="instance", activate="lrelu"),
                                      Conv2d(128, 256, 3, stride=2, bn="instance", activate="lrelu"),
                                      Conv2d(256, 512, 3, stride=2, bn="batch", activate="lrelu"),
                                      Conv2d(512, 1024, 3, stride=2, bn="batch", activate="lrelu"))
        self.embeding_block = ParameterEmbedingNet(EmbedingBlock, in_size, in_type="embed")
        in_size = in_size // 32
        in_size = 1024 * in_size * in_size
        self.cls_convs = nn.Sequential(Linear(in_size + LABEL_EMBED + STYLE_EMBED, in_size // 2, activate="lrelu"),
                                       Linear(in_size // 2, in_size // 4, activate="lrelu"),
                                       Linear(in_size // 4, num_of_classes, activate=None))

    def forward(self, x, y):
        x = self.backbone(self.conv_first(x))
        x = x.reshape(x.size(0), -1)
        y_cls, y_cnt_style = self.embeding_block(y["cls"], y["cnt_style"])
        return self.cls_convs(torch.cat([x, y_cls, y_cnt_style], dim=1))


class Discriminator(nn.Module):
    def __init__(self, in_size, in_channels, num_of_classes):
        super().__init__()
        self.adv_convs = Classifier(in_size, in_channels, 1)
        self.aux_convs = Classifier(in_size, in_channels, num_of_classes)

    def forward(self, x, y):
        return self.adv_convs(x, y).sigmoid(), self.aux_convs(x, y)
