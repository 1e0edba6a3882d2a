"""GPU tests of the split-bf16 ("bf16x3") convolution path: three bf16 MFMAs per product with fp32
accumulation must stay ~1e-5 of the fp32 result (plain bf16 would be ~2e-3, outside the 1e-3 bar)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
X3_RTOL = 5e-5

CASES = [  # (B, Hs, Cb, Cs, stride): channel counts multiples of 8
    (2, 4, 8, 8, 2), (2, 8, 64, 128, 2), (3, 5, 16, 24, 2), (8, 32, 32, 512, 2), (6, 64, 16, 64, 2),
    (2, 16, 128, 64, 2), (1, 8, 256, 256, 2), (2, 12, 64, 64, 1), (4, 16, 512, 256, 2),
    (32, 8, 256, 256, 2), (2, 8, 192, 320, 2), (4, 64, 32, 128, 2), (2, 16, 32, 96, 2), (3, 20, 24, 40, 1),
    # weight gradient on tap pairs (Cb = 32 with Cs % 64 == 0, Cb = 64 with Cs % 128 == 0): the VAE-GAN discriminator's 32 -> 64
    # block, stride 1 (padding on all sides), a K shorter than one K-tile per split
    (3, 16, 32, 64, 2), (2, 10, 64, 256, 1), (1, 4, 32, 64, 2), (2, 3, 64, 128, 1),
]


def nhwc(x):
    return x.to(DEV).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("B,Hs,Cb,Cs,stride", CASES)
def test_bf16x3_families(B, Hs, Cb, Cs, stride):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(7 + B + Hs + Cb + Cs)
    Hb = Hs * stride
    big = torch.randn(B, Cb, Hb, Hb, generator=g).clamp_min(0)    # post-ReLU-like
    small = torch.randn(B, Cs, Hs, Hs, generator=g)
    w = torch.randn(Cs, Cb, 5, 5, generator=g) * 0.05
    bias = torch.randn(Cs, generator=g)
    big_s, small_s = ops.split_f32(nhwc(big)), ops.split_f32(nhwc(small))
    rec = ops.unsplit(big_s).view(B, Hb, Hb, Cb).permute(0, 3, 1, 2)
    assert_close(rec, big, 2 ** -15, "split reconstruction")
    p0, p1 = ops.pack_w5_split(w.to(DEV), True, True)
    assert_close(ops.unsplit(p0).view(Cs, 25, Cb), w.permute(0, 2, 3, 1).reshape(Cs, 25, Cb), 2 ** -15, "p0")
    assert_close(ops.unsplit(p1).view(Cb, 25, Cs), w.permute(1, 2, 3, 0).reshape(Cb, 25, Cs), 2 ** -15, "p1")
    y = ops.conv5_gather_bf16x3(big_s, big.shape, p0, Cs, bias.to(DEV), stride, 0)
    assert_close(y, F.conv2d(big, w, bias, stride=stride, padding=2), X3_RTOL, "gather bf16x3")
    # without bias the few-tile / long-K shapes take the 2-way split-K path (atomic accumulation of two halves)
    y0 = ops.conv5_gather_bf16x3(big_s, big.shape, p0, Cs, None, stride, 0)
    assert_close(y0, F.conv2d(big, w, None, stride=stride, padding=2), X3_RTOL, "gather bf16x3 (no bias)")
    y1 = ops.conv5_gather_bf16x3(big_s, big.shape, p0, Cs, None, stride, 0)
    assert torch.equal(y0, y1), "split-K gather must be bit-reproducible"
    yt = ops.conv5_scatter_bf16x3(small_s, small.shape, p1, Cb, stride)
    assert_close(yt, F.conv_transpose2d(small, w, None, stride=stride, padding=2, output_padding=stride - 1), X3_RTOL, "scatter bf16x3")
    assert torch.equal(yt, ops.conv5_scatter_bf16x3(small_s, small.shape, p1, Cb, stride)), "scatter must be bit-reproducible"
    wr = w.clone().requires_grad_(True)
    F.conv2d(big, wr, None, stride=stride, padding=2).backward(small)
    dw = ops.conv5_wgrad_bf16x3(big_s, big.shape, small_s, small.shape, stride)
    assert_close(dw, wr.grad, X3_RTOL, "wgrad bf16x3")


def test_split_outputs_of_bn_and_transpose():
    from vae_play_amd import _lib, ops
    from ctypes import c_void_p
    g = torch.Generator().manual_seed(3)
    B, C, H = 2, 16, 8
    x = nhwc(torch.randn(B, C, H, H, generator=g))
    mean, rstd = ops.bn_stats(x, 1e-5, 0.9)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV)
    y = ops.bn_act_fwd(x, mean, rstd, gamma, beta, ops.ACT_RELU)
    ys = ops.empty_split(x.numel(), x)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.call("vp_bn_act_fwd_split_f32", ops._p(x), ops._p(mean), ops._p(rstd), ops._p(gamma), ops._p(beta), None, ops._pv(ys),
              B * H * H, C, ops.ACT_RELU, 0.0, st)
    assert_close(ops.unsplit(ys), y.permute(0, 2, 3, 1).reshape(-1), 2 ** -15, "bn split out")
    dy = nhwc(torch.randn(B, C, H, H, generator=g))
    dx, dg, db = ops.bn_act_bwd(x, dy, mean, rstd, gamma, beta, ops.ACT_RELU, 0.0, True)
    dxs = ops.empty_split(x.numel(), x)
    ws = torch.empty(_lib.load().vp_bn_workspace_bytes(B * H * H, C) // 4 + 4, device=DEV)
    dg2, db2 = torch.empty_like(dg), torch.empty_like(db)
    _lib.call("vp_bn_act_bwd_split_f32", ops._p(x), ops._p(dy), ops._p(mean), ops._p(rstd), ops._p(gamma), ops._p(beta), None,
              ops._pv(dxs), ops._p(dg2), ops._p(db2), B * H * H, C, ops.ACT_RELU, 0.0, 1, ops._p(ws), ws.numel() * 4, st)
    assert_close(ops.unsplit(dxs), dx.permute(0, 2, 3, 1).reshape(-1), 2 ** -15, "bn bwd split out")
    assert torch.equal(dg, dg2) and torch.equal(db, db2)
    xn = torch.randn(B, C, H, H, generator=g).to(DEV)
    ts = ops.empty_split(xn.numel(), xn)
    _lib.call("vp_nchw_to_nhwc_split_f32", ops._p(xn), None, ops._pv(ts), B, C, H, H, st)
    assert_close(ops.unsplit(ts), xn.permute(0, 2, 3, 1).reshape(-1), 2 ** -15, "transpose split out")


KXK = [  # (B, Hb, Wb, Cb, Cs, ks, stride): the models/blocks.py vocabulary on the split-bf16 kernels, incl. odd sizes
    (2, 16, 16, 16, 24, 3, 1), (2, 17, 13, 8, 16, 3, 2), (3, 12, 12, 64, 64, 3, 2), (2, 9, 9, 32, 8, 1, 1), (1, 20, 14, 16, 8, 1, 2),
    (2, 15, 15, 8, 8, 5, 2), (4, 32, 32, 128, 64, 3, 1), (2, 7, 9, 24, 40, 5, 1),
]


@pytest.mark.parametrize("B,Hb,Wb,Cb,Cs,ks,stride", KXK)
def test_bf16x3_kxk_families(B, Hb, Wb, Cb, Cs, ks, stride):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(B + Hb + Cb + Cs + ks)
    pad = (ks - 1) // 2
    big = torch.randn(B, Cb, Hb, Wb, generator=g)
    w = torch.randn(Cs, Cb, ks, ks, generator=g) * 0.1
    bias = torch.randn(Cs, generator=g)
    ref = F.conv2d(big, w, bias, stride=stride, padding=pad)
    small = torch.randn(ref.shape, generator=g)
    big_s, small_s = ops.split_f32(nhwc(big)), ops.split_f32(nhwc(small))
    p0, p1 = ops.pack_w_split(w.to(DEV), True, True)
    y = ops.conv_gather_bf16x3(big_s, big.shape, p0, Cs, bias.to(DEV), ks, stride, 0)
    assert_close(y, ref, X3_RTOL, "k x k gather bf16x3")
    bigr = big.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(bigr, wr, None, stride=stride, padding=pad).backward(small)
    dx = ops.conv_scatter_bf16x3(small_s, small.shape, p1, Cb, ks, stride, Hb, Wb)
    assert_close(dx, bigr.grad, X3_RTOL, "k x k scatter bf16x3")
    dw = ops.conv_wgrad_bf16x3(big_s, big.shape, small_s, small.shape, ks, stride)
    assert_close(dw, wr.grad, X3_RTOL, "k x k wgrad bf16x3")


PADDED = [  # (B, H, W, Cin, Cout, ks, stride): channel counts that are NOT multiples of 8 (coordinate channels, 1-2 outputs, RGB in)
    (2, 16, 16, 34, 32, 3, 1), (2, 16, 16, 32, 1, 3, 1), (2, 17, 13, 3, 16, 5, 2), (1, 12, 12, 34, 2, 1, 1), (2, 10, 10, 66, 12, 3, 2), (2, 16, 16, 2, 1, 3, 1), (1, 9, 11, 1, 1, 3, 1),
]


@pytest.mark.parametrize("B,H,W,Ci,Co,ks,stride", PADDED)
def test_conv2d_bf16x3_mode_with_padded_channels(B, H, W, Ci, Co, ks, stride):
    """functional.conv2d under set_conv_precision("bf16x3") zero-pads such channel counts to the next multiple of 8 and stays
    on the split-bf16 kernels: output, input gradient, weight gradient and bias gradient against torch (fp32)."""
    import vae_play_amd as V
    from vae_play_amd import functional as FH
    g = torch.Generator().manual_seed(B + H + Ci + Co + ks)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, ks, ks, generator=g) * 0.1
    b = torch.randn(Co, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = F.conv2d(xr, wr, br, stride=stride, padding=(ks - 1) // 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    V.set_conv_precision("bf16x3")
    try:
        yd = FH.conv2d(xd, wd, bd, stride)
        yd.backward(gy.to(DEV))
    finally:
        V.set_conv_precision("f32")
    assert tuple(yd.shape) == tuple(y.shape)
    assert_close(yd, y.detach(), X3_RTOL, "padded conv y")
    assert_close(xd.grad, xr.grad, X3_RTOL, "padded conv dx")
    assert_close(wd.grad, wr.grad, X3_RTOL, "padded conv dw")
    assert_close(bd.grad, br.grad, X3_RTOL, "padded conv db")


def test_bn_act_output_carries_its_split_planes_in_bf16x3_mode():
    """In bf16x3 mode BatchNorm+activation emits the bf16 hi/lo planes of its output in the same pass and the next
    convolution picks them up instead of running a split pass: the planes must equal split_f32(y), survive Function.apply,
    and be ignored once y is written in place."""
    import vae_play_amd as V
    from vae_play_amd import functional as FH, ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 12, 12, generator=g).to(DEV).requires_grad_(True)
    gamma, beta = (torch.rand(16, generator=g) + 0.5).to(DEV).requires_grad_(True), torch.randn(16, generator=g).to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(16, device=DEV), torch.ones(16, device=DEV)
    w = (torch.randn(8, 16, 3, 3, generator=g) * 0.1).to(DEV).requires_grad_(True)
    V.set_conv_precision("bf16x3")
    try:
        y = FH.batch_norm_act(x, gamma, beta, rm, rv, True, 0.9, 1e-5, "relu", 0.0)
        assert hasattr(y, "_vp_split"), "the producer's planes did not survive Function.apply"
        assert torch.equal(y._vp_split[0], ops.split_f32(y.detach()))
        assert FH._split_of(y) is y._vp_split[0]
        out = FH.conv2d(y, w, None, 1)
        ref = F.conv2d(y.detach().cpu(), w.detach().cpu(), None, padding=1)
        assert_close(out, ref, X3_RTOL, "conv on carried planes")
        out.sum().backward()
        y2 = FH.batch_norm_act(x.detach(), gamma.detach(), beta.detach(), rm, rv, True, 0.9, 1e-5, "relu", 0.0)
        y2.mul_(2.0)                                   # written in place: the carried planes are stale and must not be used
        assert FH._split_of(y2) is not y2._vp_split[0]
        assert torch.equal(FH._split_of(y2), ops.split_f32(y2))
    finally:
        V.set_conv_precision("f32")
