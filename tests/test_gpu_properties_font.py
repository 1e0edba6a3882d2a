"""Size-independent properties at the launch shapes of BASELINE config 5's LITERAL model (the font U-Net and its AC-GAN discriminator at
256 x 256, vae_play_amd/networks_BE_font.py): 3 x 3 convolutions, stride 1 and 2, on the generic (non-5x5) instantiations of the
split-bf16 kernels -- the 128x128 / 128x64 / 64x128 tiles and the 3x3 tap pairs of the weight gradient that the 16 - 32 px golden fixtures
of these networks do not reach.  As tests/test_gpu_properties.py: gather, scatter and weight gradient must be one bilinear form (each
other's adjoints / derivative) and linear in the activation; fp64 sums of device results are the reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (name, Cbig, Csmall, Hbig, stride): big = the convolution's input side
LAYERS = [("unet.down1", 64, 128, 256, 2), ("unet.down1b", 128, 128, 128, 1), ("unet.down2", 128, 256, 128, 2),
          ("unet.down3b", 512, 512, 32, 1), ("unet.down5", 512, 512, 16, 2), ("unet.skip0", 64, 64, 256, 1),
          ("unet.cat0", 128, 64, 256, 1), ("unet.cat1", 256, 128, 128, 1), ("unet.cat3", 1024, 512, 32, 1),
          ("disc.conv2", 64, 128, 128, 2), ("disc.conv4", 256, 512, 32, 2), ("disc.conv5", 512, 1024, 16, 2)]


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


@pytest.mark.parametrize("name,Cb,Cs,Hb,stride", LAYERS, ids=[l[0] for l in LAYERS])
def test_font_layer_shapes_are_adjoint_and_bilinear(name, Cb, Cs, Hb, stride):
    from vae_play_amd import ops
    B, ks = 2, 3
    Hs = ops.conv_out_size(Hb, ks, stride)
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    x, x2 = cl(torch.randn(B, Cb, Hb, Hb, generator=g)), cl(torch.randn(B, Cb, Hb, Hb, generator=g))
    y = cl(torch.randn(B, Cs, Hs, Hs, generator=g))
    w = (torch.randn(Cs, Cb, ks, ks, generator=g) * 0.05).to(DEV)
    p0, p1 = ops.pack_w_split(w, True, True)
    gather = lambda t: ops.conv_gather_bf16x3(ops.split_f32(t), t.shape, p0, Cs, None, ks, stride)
    scatter = lambda t: ops.conv_scatter_bf16x3(ops.split_f32(t), t.shape, p1, Cb, ks, stride, Hb, Hb)
    wgrad = lambda big, small: ops.conv_wgrad_bf16x3(ops.split_f32(big), tuple(big.shape), ops.split_f32(small), tuple(small.shape), ks, stride)
    tol = 3e-5
    cx = gather(x)
    assert cx.shape == y.shape
    lhs = _dot(cx, y)
    scale = (cx.double().pow(2).sum().sqrt() * y.double().pow(2).sum().sqrt()).item()
    assert abs(lhs - _dot(x, scatter(y))) <= tol * scale, f"{name}: scatter is not the adjoint of gather"
    assert abs(lhs - _dot(w, wgrad(x, y))) <= tol * scale, f"{name}: the weight gradient is not the form's derivative"
    a, b = 0.75, -1.5
    mix = gather(a * x + b * x2)
    ref = a * cx + b * gather(x2)
    err = ((mix - ref).double().pow(2).sum().sqrt() / ref.double().pow(2).sum().sqrt()).item()
    assert err <= tol * 4, f"{name}: gather not linear ({err:.2e})"
    # ... and against the fp32 reference convolution on a sample of outputs (the whole tensor at 256 px is too slow on the host)
    idx = torch.randint(0, Hs, (64, 2), generator=g)
    xr, wr = x.double().cpu(), w.double().cpu()
    worst = 0.0
    for (h, ww) in idx.tolist():
        hb0, wb0 = h * stride - 1, ww * stride - 1
        acc = torch.zeros(B, Cs, dtype=torch.float64)
        for r in range(ks):
            for q in range(ks):
                hh, wq = hb0 + r, wb0 + q
                if 0 <= hh < Hb and 0 <= wq < Hb:
                    acc += xr[:, :, hh, wq] @ wr[:, :, r, q].t()
        worst = max(worst, (cx[:, :, h, ww].double().cpu() - acc).abs().max().item())
    rms = cx.double().pow(2).mean().sqrt().item()
    assert worst <= 3e-5 * rms * 8, f"{name}: sampled outputs differ from the direct sum by {worst:.2e} (rms {rms:.2e})"
