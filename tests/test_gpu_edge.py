"""Edge layers (the 3- / 1-channel image side of the first and last convolution) on the matrix cores.

Final conv forward (models/networks.py:100-103, Conv2d(64 -> C, k5, s1, p2) + bias + Sigmoid): the tap-in-N MFMA
kernel (csrc/narrow.hip conv5s1_tapn_kernel) against a plain torch fp32 reference of the same op on the CPU."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,nout,act", [(2, 16, 16, 3, 4), (1, 32, 48, 3, 0), (3, 64, 32, 1, 4), (2, 128, 128, 3, 4), (1, 256, 256, 1, 4)])
def test_final_conv_tapn_matches_torch(B, H, W, nout, act):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(7 + H + nout)
    x = torch.rand(B, 64, H, W, generator=g) * 2.0 - 0.3       # post-ReLU-like range with some negatives
    w = (torch.rand(nout, 64, 5, 5, generator=g) - 0.5) * 0.1
    b = torch.rand(nout, generator=g) - 0.5
    ref = F.conv2d(x, w, b, padding=2)
    if act == 4:
        ref = torch.sigmoid(ref)
    from vae_play_amd import _lib
    xd = ops.channels_last(x.cuda())
    p0, _ = ops.pack_w5(w.cuda(), True, False)
    bd = b.cuda()
    y = ops.empty_cl(B, nout, H, W, xd)
    _lib.call("vp_conv5_smallout_bf16x3", ops._p(xd), ops._p(p0), ops._p(bd), ops._p(y), B, H, W, 64, nout, act, ops._stream())
    # the exact-fp32 entry point keeps its VALU kernel: the two arithmetics agree to the split's 16 bits
    assert_close(y.cpu(), ops.conv5_gather(xd, p0, bd, 1, act).cpu(), 3e-5, "tap-in-N vs exact-f32 kernel")
    assert y.shape == ref.shape
    # split-bf16 contraction: 16 significant bits per operand, fp32 accumulation
    assert_close(y.cpu(), ref, 3e-5, f"final conv tap-in-N {B}x{H}x{W}x{nout}")
    # image borders are where the halo patch is zero-filled: check them on their own
    for sl in ((..., 0, slice(None)), (..., H - 1, slice(None)), (..., slice(None), 0), (..., slice(None), W - 1)):
        assert_close(y.cpu()[sl], ref[sl], 1e-4, "border")


@pytest.mark.parametrize("B,H,W,nout", [(2, 16, 64, 3), (1, 32, 128, 1), (3, 48, 64, 3), (2, 128, 128, 3), (1, 256, 256, 1)])
def test_final_conv_weight_gradient_tapm_matches_torch(B, H, W, nout):
    """csrc/edge.hip wgrad_tapm_kernel (taps folded into the MFMA rows) against torch's autograd weight gradient on the CPU and
    against the exact-fp32 VALU kernel behind vp_conv5_wgrad_f32."""
    from vae_play_amd import _lib, ops
    g = torch.Generator().manual_seed(11 + H + nout)
    u = torch.rand(B, 64, H, W, generator=g) * 1.5                      # post-ReLU activation
    dl = (torch.rand(B, nout, H, W, generator=g) - 0.5) / B             # (x_tilde - x) / B
    w = torch.zeros(nout, 64, 5, 5, requires_grad=True)
    (F.conv2d(u, w, None, padding=2) * dl).sum().backward()
    ref = w.grad
    ud, dld = ops.channels_last(u.cuda()), ops.channels_last(dl.cuda())
    lib = _lib.load()
    nbytes = lib.vp_conv5_smallout_wgrad_bf16x3_workspace_bytes(B, H, W, 64, nout)
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, device="cuda")
    dw = torch.full((nout, 64, 5, 5), float("nan"), device="cuda")
    _lib.call("vp_conv5_smallout_wgrad_bf16x3", ops._p(ud), ops._p(dld), ops._p(dw), B, H, W, 64, nout, ops._p(ws), nbytes, ops._stream())
    assert_close(dw.cpu(), ref, 3e-5, f"final conv wgrad taps-in-M {B}x{H}x{W}x{nout}")
    assert_close(dw.cpu(), ops.conv5_wgrad(ud, dld, 1).cpu(), 3e-5, "taps-in-M vs exact-f32 kernel")
    dw2 = torch.empty_like(dw)
    _lib.call("vp_conv5_smallout_wgrad_bf16x3", ops._p(ud), ops._p(dld), ops._p(dw2), B, H, W, 64, nout, ops._p(ws), nbytes, ops._stream())
    assert torch.equal(dw, dw2), "slab reduction must be bit-reproducible"
    # the exact-fp32 form of the same scheme (v_mfma_f32_32x32x2_f32, pixels split over the four waves, fixed-order sum through LDS)
    assert lib.vp_conv5_smallout_wgrad_f32_workspace_bytes(B, H, W, 64, nout) == nbytes
    dw3, dw4 = torch.full_like(dw, float("nan")), torch.empty_like(dw)
    _lib.call("vp_conv5_smallout_wgrad_f32", ops._p(ud), ops._p(dld), ops._p(dw3), B, H, W, 64, nout, ops._p(ws), nbytes, ops._stream())
    assert_close(dw3.cpu(), ref, 3e-6, f"final conv wgrad taps-in-M exact f32 {B}x{H}x{W}x{nout}")
    _lib.call("vp_conv5_smallout_wgrad_f32", ops._p(ud), ops._p(dld), ops._p(dw4), B, H, W, 64, nout, ops._p(ws), nbytes, ops._stream())
    assert torch.equal(dw3, dw4), "exact-f32 taps-in-M: bit-reproducible"


@pytest.mark.parametrize("B,H,W,nin", [(2, 16, 32, 3), (1, 24, 40, 3), (3, 33, 17, 1), (2, 128, 128, 3), (1, 256, 256, 1)])
def test_final_conv_input_gradient_rowk_matches_torch(B, H, W, nin):
    """csrc/edge.hip dgrad_rowk_kernel (one kernel row of taps per MFMA k-step) against torch's autograd input gradient of
    Conv2d(64 -> C, k5, s1, p2) on the CPU (ragged sizes included: tiles are 8 x 32 pixels)."""
    from vae_play_amd import _lib, ops
    g = torch.Generator().manual_seed(13 + H + nin)
    dl = (torch.rand(B, nin, H, W, generator=g) - 0.5) / B
    w = (torch.rand(nin, 64, 5, 5, generator=g) - 0.5) * 0.1
    u = torch.zeros(B, 64, H, W, requires_grad=True)
    (F.conv2d(u, w, None, padding=2) * dl).sum().backward()
    ref = u.grad
    dld = ops.channels_last(dl.cuda())
    out = torch.full((B, 64, H, W), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
    _lib.call("vp_conv5_smallin_dgrad_bf16x3", ops._p(dld), ops._p(w.cuda()), ops._p(out), B, H, W, nin, 64, ops._stream())
    assert torch.isfinite(out).all(), "every output element must be written"
    assert_close(out.cpu(), ref, 3e-5, f"final conv input gradient rows-in-K {B}x{H}x{W}x{nin}")
    for sl in ((..., 0, slice(None)), (..., H - 1, slice(None)), (..., slice(None), 0), (..., slice(None), W - 1)):
        assert_close(out.cpu()[sl], ref[sl], 1e-4, "border")
    # the exact-fp32 scatter kernel computes the same gradient
    _, p1 = ops.pack_w5(w.cuda(), False, True)
    assert_close(out.cpu(), ops.conv5_scatter(dld, p1, 1).cpu(), 3e-5, "rows-in-K vs exact-f32 kernel")
    # ... and so does the exact-fp32 form of the rows-in-K kernel (v_mfma_f32_32x32x2_f32, k permuted inside a kernel row)
    out32 = torch.full((B, 64, H, W), float("nan"), device="cuda").contiguous(memory_format=torch.channels_last)
    _lib.call("vp_conv5_smallin_dgrad_f32", ops._p(dld), ops._p(w.cuda()), ops._p(out32), B, H, W, nin, 64, ops._stream())
    assert torch.isfinite(out32).all(), "every output element must be written"
    assert_close(out32.cpu(), ref, 2e-6, f"final conv input gradient rows-in-K exact f32 {B}x{H}x{W}x{nin}")


@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 16, 16, 1, 32), (1, 24, 40, 3, 32), (2, 33, 17, 1, 64), (3, 128, 128, 1, 32)])
def test_first_conv_stride1_input_gradient_matches_torch(B, H, W, cin, cout):
    """Input gradient of Conv2d(1|3 -> 32|64, k5, s1, p2) (the VAE-GAN discriminator's first layer) through the autograd op in
    bf16x3 mode: the tap-in-N kernel with 32 / 64 gathered channels and flipped taps, against torch on the CPU."""
    from vae_play_amd import functional as Fh
    g = torch.Generator().manual_seed(17 + H + cout)
    x = torch.rand(B, cin, H, W, generator=g)
    w = (torch.rand(cout, cin, 5, 5, generator=g) - 0.5) * 0.2
    b = torch.rand(cout, generator=g) - 0.5
    gy = torch.randn(B, cout, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, b, padding=2).backward(gy)
    Fh.set_conv_precision("bf16x3")
    try:
        xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        y = Fh.conv5x5(xd, wd, bd, stride=1)
        y.backward(gy.cuda().contiguous(memory_format=torch.channels_last))
    finally:
        Fh.set_conv_precision("f32")
    assert_close(xd.grad.cpu(), xr.grad, 3e-5, "first conv input gradient")
    for sl in ((..., 0, slice(None)), (..., H - 1, slice(None)), (..., slice(None), 0), (..., slice(None), W - 1)):
        assert_close(xd.grad.cpu()[sl], xr.grad[sl], 1e-4, "border")


@pytest.mark.parametrize("B,H,W,cin,cout,act", [(2, 16, 16, 1, 32, "relu"), (1, 24, 40, 3, 32, None), (2, 33, 17, 1, 64, "relu"),
                                               (3, 128, 128, 1, 32, "relu")])
def test_first_conv_stride1_forward_matches_torch(B, H, W, cin, cout, act):
    """Forward pass of Conv2d(1|3 -> 32|64, k5, s1, p2) + bias (+ ReLU) through the autograd op in bf16x3 mode (the rows-in-K kernel's
    forward form) and its gradients, against torch on the CPU."""
    from vae_play_amd import functional as Fh
    g = torch.Generator().manual_seed(19 + H + cout)
    x = torch.rand(B, cin, H, W, generator=g)
    w = (torch.rand(cout, cin, 5, 5, generator=g) - 0.5) * 0.2
    b = torch.rand(cout, generator=g) - 0.5
    gy = torch.randn(B, cout, H, W, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=2)
    yr = torch.relu(yr) if act == "relu" else yr
    yr.backward(gy)
    Fh.set_conv_precision("bf16x3")
    try:
        xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        assert Fh.first_conv_s1_edge(wd, 1)
        y = Fh.conv5x5(xd, wd, bd, stride=1, act=act)
        y.backward(gy.cuda().contiguous(memory_format=torch.channels_last))
    finally:
        Fh.set_conv_precision("f32")
    assert_close(y.detach().cpu(), yr.detach(), 3e-5, "first conv forward")
    assert_close(xd.grad.cpu(), xr.grad, 3e-5, "input gradient")
    assert_close(wd.grad.cpu(), wr.grad, 3e-5, "weight gradient")
    assert_close(bd.grad.cpu(), br.grad, 3e-5, "bias gradient")
