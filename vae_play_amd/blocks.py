"""Drop-in classes for kungyao/vae-play's ``models/blocks.py`` on the HIP back end.

Same constructor signatures and ``state_dict`` keys as the reference:
  Conv2d(in, out, k, stride=1, bn=None|"batch"|"instance", activate="relu"|"lrelu"|"tanh"|None)   models/blocks.py:5-34
     keys conv.0.weight[, conv.0.bias], conv.1.{weight,bias,running_mean,running_var,num_batches_tracked} (bn="batch")
  Linear(in, out, bias=True, activate=...)  (LeakyReLU slope 0.2)                                      models/blocks.py:36-50
  AddCoords(if_normalize=False)                                                                        models/blocks.py:97-112
  Down(in, out, k, if_add_coord=False)                                                                 models/blocks.py:114-127
  Up(in, out, if_add_coord=False): 2 x [conv3 + BN + ReLU] then bilinear x2                            models/blocks.py:129-146
Convolution, normalisation + activation, bilinear resize and coordinate channels all run as HIP kernels (k x k
implicit GEMM on the f32 MFMA path, fused norm/activation epilogues).
  SelfAttentionBlock(in_channel): q/k/v 1x1 Conv2d (+ReLU), softmax(Q K^T), gamma                     models/blocks.py:66-96
SCSEBlock is out of scope.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import functional as F_hip
from .networks import BatchNormAct, LinearParams

_CONV_LRELU, _LIN_LRELU = 0.02, 0.2   # models/blocks.py:28 and :44 use different slopes


class ConvKParams(nn.Module):
    """nn.Conv2d's weight/bias layout, key names and default init; k in {1, 3, 5}, padding (k-1)//2."""

    def __init__(self, in_channel: int, out_channel: int, kernel_size: int, stride: int, bias: bool):
        super().__init__()
        if kernel_size not in (1, 3, 5) or stride not in (1, 2):
            raise ValueError("HIP Conv2d supports kernel_size 1|3|5 and stride 1|2")
        self.stride, self.kernel_size = stride, kernel_size
        self.weight = nn.Parameter(torch.empty(out_channel, in_channel, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channel)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weight)
            if fan_in != 0:
                bound = 1 / math.sqrt(fan_in)
                nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return F_hip.conv2d(x, self.weight, self.bias, self.stride)


class _InstanceNormActM(nn.Module):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False) fused with the block's activation (no state)."""

    def __init__(self, act: Optional[str], slope: float):
        super().__init__()
        self.act, self.slope = act, slope

    def forward(self, x):
        return F_hip.instance_norm_act(x, 1e-5, self.act, self.slope)


class _ActM(nn.Module):
    def __init__(self, act: Optional[str], slope: float):
        super().__init__()
        self.act, self.slope = act, slope

    def forward(self, x):
        return F_hip.activation(x, self.act, self.slope)


def _act_name(activate: Optional[str]):
    if activate not in (None, "relu", "lrelu", "tanh"):
        return None   # the reference silently ignores unknown names (no module appended)
    return activate


class Conv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, stride=1, bn=None, activate='relu'):
        super().__init__()
        bias = bn is None
        act = _act_name(activate)
        mods = [ConvKParams(in_channel, out_channel, kernel_size, stride, bias)]
        if bn == "batch":
            mods.append(BatchNormAct(out_channel, momentum=0.1, eps=1e-5, act=act, slope=_CONV_LRELU))   # index 1 = nn.BatchNorm2d keys
        elif bn == "instance":
            mods.append(_InstanceNormActM(act, _CONV_LRELU))
        elif act is not None:
            mods.append(_ActM(act, _CONV_LRELU))
        self.conv = nn.Sequential(*mods)

    def forward(self, input):
        return self.conv(input)


class Linear(nn.Module):
    def __init__(self, in_channel, out_channel, bias=True, activate='relu'):
        super().__init__()
        mods = [LinearParams(in_channel, out_channel, bias=bias)]
        act = _act_name(activate)
        if act is not None:
            mods.append(_ActM(act, _LIN_LRELU))
        self.fc = nn.Sequential(*mods)

    def forward(self, x):
        return self.fc(x)


class AddCoords(nn.Module):
    def __init__(self, if_normalize=False):
        super().__init__()
        self.if_normalize = if_normalize

    def forward(self, x):
        return F_hip.add_coords(x, self.if_normalize)


class Down(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, if_add_coord=False):
        super().__init__()
        self.if_add_coord = if_add_coord
        coord_channel = 2 if if_add_coord else 0
        self.conv = Conv2d(in_channel + coord_channel, out_channel, kernel_size, stride=2)
        if if_add_coord:
            self.add_coord = AddCoords()

    def forward(self, x):
        if self.if_add_coord:
            x = self.add_coord(x)
        return self.conv(x)


class Up(nn.Module):
    def __init__(self, in_channel, out_channel, if_add_coord=False):
        super().__init__()
        self.if_add_coord = if_add_coord
        coord_channel = 2 if if_add_coord else 0
        self.conv = nn.Sequential(
            Conv2d(in_channel + coord_channel, out_channel, 3, stride=1, bn="batch"),
            Conv2d(out_channel, out_channel, 3, stride=1, bn="batch"))
        if if_add_coord:
            self.add_coord = AddCoords()

    def forward(self, x):
        if self.if_add_coord:
            x = self.add_coord(x)
        x = self.conv(x)
        return F_hip.upsample2x_bilinear(x)


class SelfAttentionBlock(nn.Module):
    """models/blocks.py:66-96: keys q.conv.0.*, k.conv.0.*, v.conv.0.*, gamma.  (The reference builds q/k/v with its
    ``Conv2d`` block's default activation, so they end in a ReLU; reproduced.)"""

    def __init__(self, in_channel):
        super().__init__()
        self.q = Conv2d(in_channel, in_channel // 8, 1)
        self.k = Conv2d(in_channel, in_channel // 8, 1)
        self.v = Conv2d(in_channel, in_channel, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return F_hip.self_attention(x, self.q(x), self.k(x), self.v(x), self.gamma)


class GlobalAvgPool(nn.Module):
    """nn.AdaptiveAvgPool2d((1, 1)); output (B, C, 1, 1) like the reference (no parameters)."""

    def forward(self, x):
        y = F_hip.global_avg_pool(x)
        return y.reshape(y.size(0), y.size(1), 1, 1)
