"""Size-independent properties at BASELINE.json's FULL sizes (128 x 128 x 3, latent 128, 32 images per GPU), where the CPU oracle
is too slow to be the checker for every layer: the three convolution families must be each other's adjoints and bilinear, and
BatchNorm's outputs / input gradients must satisfy the identities of the normalisation.  Computed entirely on the device through
the C ABI; the reference values are fp64 sums of the same device results."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (name, Cbig, Csmall, Hs): the seven 5x5 stride-2 layers of the 128 x 128 network (big = the 2x larger image side)
LAYERS = [("enc1", 64, 128, 32), ("enc2", 128, 256, 16), ("enc3", 256, 512, 8),
          ("dec0", 256, 512, 8), ("dec1", 128, 256, 16), ("dec2", 64, 128, 32), ("dec3", 64, 64, 64)]
B = 32


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


@pytest.mark.parametrize("name,Cb,Cs,Hs", LAYERS)
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_conv_families_are_adjoint_and_bilinear_at_full_size(name, Cb, Cs, Hs, precision):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    Hb = 2 * Hs
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    x, x2 = cl(torch.randn(B, Cb, Hb, Hb, generator=g)), cl(torch.randn(B, Cb, Hb, Hb, generator=g))
    y = cl(torch.randn(B, Cs, Hs, Hs, generator=g))
    w = (torch.randn(Cs, Cb, 5, 5, generator=g) * 0.05).to(DEV)
    if precision == "bf16x3":
        p0, p1 = ops.pack_w5_split(w, True, True)
        gather = lambda t: ops.conv5_gather_bf16x3(ops.split_f32(t), t.shape, p0, Cs, None, 2, 0)
        scatter = lambda t: ops.conv5_scatter_bf16x3(ops.split_f32(t), t.shape, p1, Cb, 2)
        wgrad = lambda big, small: ops.conv5_wgrad_bf16x3(ops.split_f32(big), tuple(big.shape), ops.split_f32(small), tuple(small.shape), 2)
        tol = 3e-5
    else:
        p0, p1 = ops.pack_w5(w, True, True)
        gather = lambda t: ops.conv5_gather(t, p0, None, 2, 0)
        scatter = lambda t: ops.conv5_scatter(t, p1, 2)
        wgrad = lambda big, small: ops.conv5_wgrad(big, small, 2)
        tol = 3e-6
    cx = gather(x)
    # <conv(x), y> = <x, conv^T(y)> = <w, wgrad(x, y)>: gather, scatter and weight-gradient kernels are one bilinear form
    lhs = _dot(cx, y)
    scale = (cx.double().pow(2).sum().sqrt() * y.double().pow(2).sum().sqrt()).item()
    assert abs(lhs - _dot(x, scatter(y))) <= tol * scale, f"{name}: scatter is not the adjoint of gather"
    assert abs(lhs - _dot(w, wgrad(x, y))) <= tol * scale, f"{name}: the weight gradient is not the form's derivative"
    # linearity in the activation
    a, b = 0.75, -1.5
    mix = gather(a * x + b * x2)
    ref = a * cx + b * gather(x2)
    err = ((mix - ref).double().pow(2).sum().sqrt() / ref.double().pow(2).sum().sqrt()).item()
    assert err <= tol * 4, f"{name}: gather not linear ({err:.2e})"


@pytest.mark.parametrize("C,H", [(64, 64), (64, 128), (512, 8)])
def test_batchnorm_identities_at_full_size(C, H):
    from vae_play_amd import functional as FH
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, H, generator=g) * 1.7 + torch.randn(1, C, 1, 1, generator=g) * 3).to(DEV)
    x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.3).to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y = FH.batch_norm_act(x, gamma, beta, rm, rv, True, 0.9, 1e-5, None, 0.0)
    yd = y.detach().double()
    m = yd.mean(dim=(0, 2, 3)); v = yd.var(dim=(0, 2, 3), unbiased=False)
    assert (m - beta.detach().double()).abs().max().item() <= 2e-5, "mean of the normalised output must be beta"
    assert ((v.sqrt() - gamma.detach().double()).abs() / gamma.detach().double()).max().item() <= 2e-5, "std must be gamma"
    gy = torch.randn(y.shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    dx = x.grad.double()
    xhat = ((x.detach().double() - x.detach().double().mean(dim=(0, 2, 3), keepdim=True))
            / x.detach().double().var(dim=(0, 2, 3), unbiased=False, keepdim=True).add(1e-5).sqrt())
    n = dx.abs().sum(dim=(0, 2, 3))
    # the input gradient of a batch-normalised layer is orthogonal to 1 and to xhat, channel by channel
    assert (dx.sum(dim=(0, 2, 3)).abs() / n).max().item() <= 1e-5
    assert ((dx * xhat).sum(dim=(0, 2, 3)).abs() / n).max().item() <= 1e-5
    assert ((beta.grad.double() - gy.double().sum(dim=(0, 2, 3))).abs() / gy.double().abs().sum(dim=(0, 2, 3))).max().item() <= 1e-5
