"""vp_conv3_small_wgrad_f32 (csrc/small3.hip): weight gradient of the few-channel 3x3 convolutions of the networks_BE heads
(models/networks_BE.py:39-66) on the vector ALUs in exact fp32.  Checked against torch's own conv2d weight gradient in fp64 on the
CPU (ragged sizes: tiles are 8 x 32 pixels), for bit-reproducibility, and through the autograd front end in both arithmetic modes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_dw(x, dy):
    x64, dy64 = x.double().cpu(), dy.double().cpu()
    return torch.nn.grad.conv2d_weight(x64, (dy.shape[1], x.shape[1], 3, 3), dy64, stride=1, padding=1)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 16, 32, 8, 8), (1, 9, 33, 34, 8), (3, 8, 8, 10, 4), (2, 24, 40, 4, 1), (1, 130, 70, 36, 8),
                                         (4, 64, 64, 4, 8), (2, 5, 3, 1, 1), (16, 32, 32, 8, 4)])
def test_against_fp64_and_reproducible(B, H, W, Ci, Co):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(B + H * 3 + Ci * 7 + Co)
    x = torch.randn(B, Ci, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Co, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    assert ops.conv3_small_wgrad_applicable(B, H, W, Ci, Co)
    dw = ops.conv3_small_wgrad(x, dy)
    dw2 = ops.conv3_small_wgrad(x, dy)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2)
    ref = _ref_dw(x, dy)
    scale = (B * H * W) ** 0.5
    assert (dw.double().cpu() - ref).abs().max().item() <= 2e-6 * scale, (dw.double().cpu() - ref).abs().max().item()


def _ref_conv(x, w, b):
    return torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None if b is None else b.double().cpu(), stride=1, padding=1)


@pytest.mark.parametrize("B,H,W,Ci,Co,bias", [(2, 16, 32, 8, 8, False), (1, 9, 33, 34, 8, False), (3, 8, 8, 10, 4, True), (2, 24, 40, 4, 1, True),
                                              (1, 70, 130, 36, 8, False), (2, 40, 24, 4, 8, True), (2, 5, 3, 1, 1, True), (2, 17, 31, 12, 5, True),
                                              (2, 33, 70, 36, 1, True), (1, 20, 40, 33, 4, False), (2, 12, 64, 30, 6, True)])
def test_forward_and_input_gradient_against_fp64(B, H, W, Ci, Co, bias):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(B + H * 3 + Ci * 7 + Co)
    x = torch.randn(B, Ci, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Co, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * 0.2).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV) if bias else None
    y = ops.conv3_small_fwd(x, w, b)
    ref = _ref_conv(x, w, b)
    assert (y.double().cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    dx = ops.conv3_small_dgrad(dy, w)
    ref_dx = torch.nn.grad.conv2d_input((B, Ci, H, W), w.double().cpu(), dy.double().cpu(), stride=1, padding=1)
    assert (dx.double().cpu() - ref_dx).abs().max().item() <= 1e-5 * max(1.0, ref_dx.abs().max().item())
    torch.cuda.synchronize()


def test_shapes_outside_the_kernel_are_refused():
    from vae_play_amd import _lib, ops
    assert not ops.conv3_small_wgrad_applicable(1, 8, 8, 48, 8)        # Cin * Cout > 288: the MFMA kernels' territory
    assert not ops.conv3_small_wgrad_applicable(1, 8, 8, 8, 16)
    assert not ops.conv3_small_wgrad_applicable(1, 8, 8, 64, 1)
    x = torch.zeros(1, 48, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last)
    dy = torch.zeros(1, 8, 8, 8, device=DEV).contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.VaePlayHipError):
        ops.conv3_small_wgrad(x, dy)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("Ci,Co,bias", [(34, 8, False), (8, 8, False), (4, 1, True), (10, 4, False)])
def test_front_end_weight_gradient_matches_the_gemm_path(precision, Ci, Co, bias, monkeypatch):
    """blocks.Conv2d through autograd: the same layer on the VALU kernels (forward, input gradient, weight gradient) and on the
    implicit-GEMM ones (padded channels in bf16x3 mode) must agree to the arithmetic mode's tolerance."""
    import vae_play_amd as V
    from vae_play_amd import functional as F
    from vae_play_amd.blocks import Conv2d
    torch.manual_seed(Ci * 10 + Co)
    layer = Conv2d(Ci, Co, 3, bn=None if bias else "batch", activate="relu").to(DEV)
    x = torch.randn(2, Ci, 24, 40, device=DEV, requires_grad=True)
    V.set_conv_precision(precision)
    try:
        res = {}
        for small in (True, False):
            monkeypatch.setattr(F, "_SMALL3", small)
            monkeypatch.setattr(F, "_SMALL3_ALL", small)
            layer.zero_grad()
            x.grad = None
            layer(x).square().sum().backward()
            res[small] = ([p.grad.detach().clone() for p in layer.parameters()], x.grad.detach().clone())
    finally:
        V.set_conv_precision("f32")
    tol = 2e-5 if precision == "f32" else 2e-4
    for a, b in zip(res[True][0], res[False][0]):
        assert (a - b).norm().item() <= tol * b.norm().item() + 1e-12
    assert (res[True][1] - res[False][1]).norm().item() <= tol * res[False][1].norm().item() + 1e-12
