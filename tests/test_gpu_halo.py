"""GPU tests of the halo-resident MFMA convolution (vae_play_amd/csrc/halo.hip): the layers with 32 (or 3, padded
to 8) channels on the gathered side are dispatched to it by vp_conv5_gather_bf16x3 / vp_conv5_scatter_bf16x3.
Compared with torch's fp32 conv on CPU at the bf16x3 tolerance; VP_HALO=0 would run the same calls on igemm16."""
from ctypes import c_void_p

import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
X3_RTOL = 5e-5


def nhwc(x):
    return x.to(DEV).contiguous(memory_format=torch.channels_last)


# (B, Hs, Cb, Cs, stride, bias/act): every halo gather configuration, incl. ragged channel counts and >1 column tile
GATHER = [
    (2, 32, 32, 3, 1, True), (1, 16, 32, 32, 1, False), (3, 48, 32, 8, 1, True),      # kind 1: final conv forward
    (2, 16, 32, 128, 2, False), (1, 8, 32, 256, 2, False), (2, 24, 32, 64, 2, False),  # kind 2: last block dgrad
    (1, 16, 32, 96, 2, True),
    (2, 32, 64, 3, 1, True), (1, 16, 128, 5, 1, False),                                  # channel chunks of 32
]


@pytest.mark.parametrize("B,Hs,Cb,Cs,stride,epi", GATHER)
def test_halo_gather(B, Hs, Cb, Cs, stride, epi):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(11 + B + Hs + Cs)
    Hb = Hs * stride
    big = torch.randn(B, Cb, Hb, Hb, generator=g)
    w = torch.randn(Cs, Cb, 5, 5, generator=g) * 0.05
    bias = torch.randn(Cs, generator=g) if epi else None
    big_s = ops.split_f32(nhwc(big))
    p0, _ = ops.pack_w5_split(w.to(DEV), True, False)
    act = ops.ACT_SIGMOID if epi else ops.ACT_NONE
    y = ops.conv5_gather_bf16x3(big_s, big.shape, p0, Cs, bias.to(DEV) if epi else None, stride, act)
    ref = F.conv2d(big, w, bias, stride=stride, padding=2)
    if epi:
        ref = torch.sigmoid(ref)
    assert_close(y, ref, X3_RTOL, f"halo gather {B}x{Hs} Cb{Cb} Cs{Cs} s{stride}")


@pytest.mark.parametrize("B,H,Cb,C", [(2, 32, 32, 3), (1, 16, 32, 1), (2, 48, 16, 3)])
def test_halo_scatter_stride1_padded(B, H, Cb, C):
    """Input gradient of Conv2d(Cb -> C, k5, s1, p2) with dy padded to 8 channels (final conv, models/networks.py:100)."""
    from vae_play_amd import _lib, ops
    g = torch.Generator().manual_seed(5 + B + H + C)
    dy = torch.randn(B, C, H, H, generator=g)
    w = torch.randn(C, Cb, 5, 5, generator=g) * 0.05
    dy8 = torch.zeros(B, 8, H, H)
    dy8[:, :C] = dy
    dy_s = ops.split_f32(nhwc(dy8))
    p1 = ops.empty_split(Cb * 25 * 8, dy_s)
    wd = w.to(DEV)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.call("vp_pack_w5_p1_split_padded", ops._p(wd), ops._pv(p1), C, Cb, 8, st)
    dx = ops.conv5_scatter_bf16x3(dy_s, dy8.shape, p1, Cb, 1)
    ref = F.conv_transpose2d(dy, w, None, stride=1, padding=2)
    assert_close(dx, ref, X3_RTOL, f"halo scatter s1 {B}x{H} Cb{Cb} C{C}")
