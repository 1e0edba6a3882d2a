"""Drop-in classes for kungyao/vae-play's ``models/networks_BE_font.py``: the font U-Net
generator and its AC-GAN discriminator on HIP kernels.

The reference builds every stack with hand-unrolled ``in_channels / out_channels`` bookkeeping.  Here the architecture is
DATA: each stack is a table of layer records produced by ``_ladder`` (the doubling channel ladder all of these nets share)
and turned into modules by ``_stack``; module names, construction ORDER (it fixes the consumption of the global RNG, i.e.
bit-identical default initialisation) and ``state_dict`` keys are the reference's, so checkpoints interchange:

  EmbedingBlock          <- models/networks_BE_font.py:21-45     StyleEncodeBlock <- :47-67
  ParameterEmbedingNet   <- :69-83      MaskNet / EdgeNet (font variant: three 3x3 convs, InstanceNorm) <- :85-123
  ComposeNet(in_size)    <- :125-232    Classifier <- :234-263    Discriminator <- :265-274
"""
from __future__ import annotations

import math
from typing import Iterator, List, Sequence, Tuple

import torch
import torch.nn as nn

from .blocks import Conv2d, GlobalAvgPool, Linear, SelfAttentionBlock, Up

LABEL_EMBED = 256
STYLE_EMBED = 256
_N_CLASSES, _N_STYLE = 143, 5            # width of the one-hot class vector / of the content-style vector (:76-79)
_CONV, _DENSE = "conv", "dense"


def _ladder(first: int, cap: int, steps: int) -> Iterator[Tuple[int, int]]:
    """(c_in, c_out) of ``steps`` successive stages whose width doubles from ``first`` up to ``cap``."""
    width = first
    for _ in range(steps):
        nxt = min(2 * width, cap)
        yield width, nxt
        width = nxt


def _stack(records: Sequence[tuple]) -> List[nn.Module]:
    """Layer records -> modules.  ("conv", cin, cout, k, stride, norm, act) | ("dense", cin, cout, act) | a ready module."""
    built = []
    for rec in records:
        if isinstance(rec, nn.Module):
            built.append(rec)
        elif rec[0] == _CONV:
            _, cin, cout, k, stride, norm, act = rec
            built.append(Conv2d(cin, cout, k, stride=stride, bn=norm, activate=act))
        else:
            _, cin, cout, act = rec
            built.append(Linear(cin, cout, activate=act))
    return built


def _conv(cin, cout, k=3, stride=1, norm="instance", act="relu"):
    return (_CONV, cin, cout, k, stride, norm, act)


class EmbedingBlock(nn.Module):
    """Vector -> embedding: two plain dense layers, three self-attention blocks on a 1 x 1 map, two LeakyReLU dense layers."""

    def __init__(self, in_channels, out_channels, in_size):
        super().__init__()
        width = out_channels
        self.convs_first = nn.Sequential(*_stack([(_DENSE, in_channels, width, None), (_DENSE, width, width, None)]))
        self.attention = nn.Sequential(*[SelfAttentionBlock(width) for _ in range(3)])
        self.embeding = nn.Sequential(*_stack([(_DENSE, width, width, "lrelu")] * 2))

    def forward(self, x):
        v = self.convs_first(x)
        v = self.attention(v[:, :, None, None])
        return self.embeding(v.flatten(1))


class StyleEncodeBlock(nn.Module):
    """Image -> embedding: stride-2 3x3 ladder down to 4 x 4, a 1x1 projection, global average."""

    def __init__(self, in_channels, out_channels, in_size):
        super().__init__()
        base = 64
        plan = [_conv(in_channels, base, stride=2)]
        last = base
        for cin, cout in _ladder(base, out_channels, int(math.log2(in_size)) - 3):
            plan.append(_conv(cin, cout, stride=2))
            last = cout
        plan += [_conv(last, out_channels, k=1), GlobalAvgPool()]
        self.convs = nn.Sequential(*_stack(plan))

    def forward(self, x):
        return self.convs(x).flatten(1)


class ParameterEmbedingNet(nn.Module):
    """A label encoder and a style encoder of one block type: from the (class, style) vectors or from the glyph image."""

    _INPUT_WIDTHS = {"image": (3, 3), "embed": (_N_CLASSES, _N_STYLE)}

    def __init__(self, encode_block, in_size, in_type=None):
        super().__init__()
        if in_type in self._INPUT_WIDTHS:
            w_label, w_style = self._INPUT_WIDTHS[in_type]
            self.label_encode_block = encode_block(w_label, LABEL_EMBED, in_size)
            self.style_encode_block = encode_block(w_style, STYLE_EMBED, in_size)

    def forward(self, y_cls, y_cnt_style):
        return self.label_encode_block(y_cls), self.style_encode_block(y_cnt_style)


class MaskNet(nn.Module):
    def __init__(self, in_channel):
        super().__init__()
        self.out_channels = 1
        c = in_channel
        self.predictor = nn.Sequential(*_stack([_conv(c, c), _conv(c, c), _conv(c, self.out_channels, norm=None, act=None)]))

    def forward(self, x):
        return self.predictor(x)


class EdgeNet(MaskNet):
    pass


class ComposeNet(nn.Module):
    """U-Net over the glyph: a stride-2 encoder ladder to 4 x 4, a dense bottleneck that also takes the label and style
    embeddings, and a decoder of (bilinear up, 3x3 on the skip, 3x3 on the concatenation) stages; two prediction heads."""

    _BASE, _CAP, _FLOOR = 64, 512, 4

    def __init__(self, in_size):
        super().__init__()
        base, cap, floor = self._BASE, self._CAP, self._FLOOR
        stages = list(_ladder(base, cap, int(math.log2(in_size // floor))))
        self.down = nn.ModuleList([Conv2d(3, base, 3, stride=1, bn="instance")])
        for cin, cout in stages:
            self.down.append(nn.Sequential(*_stack([_conv(cin, cout, stride=2, norm="batch"), _conv(cout, cout)])))
        self.embeding_block = ParameterEmbedingNet(EmbedingBlock, in_size, in_type="embed")
        self.style_encoder = ParameterEmbedingNet(StyleEncodeBlock, in_size, in_type="image")
        deepest = stages[-1][1] if stages else base
        flat = deepest * floor * floor
        self.relay_convs = nn.Sequential(*_stack([(_DENSE, flat + LABEL_EMBED + STYLE_EMBED, flat, "relu"), (_DENSE, flat, flat, "relu")]))
        self.up, self.skip, self.cat = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for narrow, wide in stages:               # stage k of the decoder undoes stage k of the encoder
            self.up.append(Up(wide, narrow))
            self.skip.append(Conv2d(narrow, narrow, 3, stride=1, bn="instance"))
            self.cat.append(Conv2d(2 * narrow, narrow, 3, stride=1, bn="instance"))
        self.mask_net = MaskNet(base)
        self.edge_net = EdgeNet(base)

    def forward(self, x, y=None):
        label, style = self.embeding_block(y["cls"], y["cnt_style"]) if y is not None else self.style_encoder(x, x)
        pyramid = []
        for stage in self.down:
            x = stage(x)
            pyramid.append(x)
        shape = x.shape                            # NCHW -> (C, H, W) flatten order, like the reference's reshape
        x = self.relay_convs(torch.cat([x.reshape(shape[0], -1), label, style], dim=1)).reshape(shape)
        for k in reversed(range(len(self.up))):
            x = self.cat[k](torch.cat([self.up[k](x), self.skip[k](pyramid[k])], dim=1))
        return {"edges": self.edge_net(x), "masks": self.mask_net(x)}


class Classifier(nn.Module):
    """Five stride-2 3x3 stages (InstanceNorm, then BatchNorm for the two widest), the embeddings appended to the flattened
    map, three dense layers down to ``num_of_classes``."""

    _WIDTHS = ((64, "instance"), (128, "instance"), (256, "instance"), (512, "batch"), (1024, "batch"))

    def __init__(self, in_size, in_channels, num_of_classes):
        super().__init__()
        plan, cin = [], in_channels
        for width, norm in self._WIDTHS:
            plan.append(_conv(cin, width, stride=2, norm=norm, act="lrelu"))
            cin = width
        first, *rest = _stack(plan)
        self.conv_first = first
        self.backbone = nn.Sequential(*rest)
        self.embeding_block = ParameterEmbedingNet(EmbedingBlock, in_size, in_type="embed")
        side = in_size // 2 ** len(self._WIDTHS)
        flat = cin * side * side
        self.cls_convs = nn.Sequential(*_stack([(_DENSE, flat + LABEL_EMBED + STYLE_EMBED, flat // 2, "lrelu"),
                                                (_DENSE, flat // 2, flat // 4, "lrelu"), (_DENSE, flat // 4, num_of_classes, None)]))

    def forward(self, x, y):
        feat = self.backbone(self.conv_first(x)).flatten(1)
        label, style = self.embeding_block(y["cls"], y["cnt_style"])
        return self.cls_convs(torch.cat([feat, label, style], dim=1))


class Discriminator(nn.Module):
    """AC-GAN discriminator: an adversarial Classifier (one logit, squashed) and an auxiliary one over the glyph classes."""

    def __init__(self, in_size, in_channels, num_of_classes):
        super().__init__()
        self.adv_convs = Classifier(in_size, in_channels, 1)
        self.aux_convs = Classifier(in_size, in_channels, num_of_classes)

    def forward(self, x, y):
        return torch.sigmoid(self.adv_convs(x, y)), self.aux_convs(x, y)
