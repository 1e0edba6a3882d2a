"""Size-independent properties at BASELINE.json's FULL sizes (128 x 128 x 3, latent 128, 32 images per GPU), where the CPU oracle
is too slow to be the checker for every layer: the three convolution families must be each other's adjoints and bilinear, and
BatchNorm's outputs / input gradients must satisfy the identities of the normalisation.  Computed entirely on the device through
the C ABI; the reference values are fp64 sums of the same device results."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (name, Cbig, Csmall, Hs): the 5x5 stride-2 layers on the split-bf16 / exact-f32 kernels (big = the 2x larger image side)
def _layers(img, L):
    size = 64 << (L - 1)
    out = [(f"enc{i}", 64 << (i - 1), 64 << i, img >> (i + 1)) for i in range(1, L)]
    for i in range(L):
        cin, cout = (size if i == 0 else size >> (i - 1)), size >> i
        out.append((f"dec{i}", cout, cin, 8 << i))
    return out


# BASELINE.json configs at their per-GPU launch shapes: every tile / split-K branch choose_tile16, wgrad_nsplit, the split-K
# rules of conv16.hip and (where enabled) the pipelined-kernel dispatch take on them runs under test
CONFIGS = {"c3_128px_b32": (32, _layers(128, 4)),           # config 3 (and 4): the benchmark shard
           "c2_64px_b128": (128, _layers(64, 3)),           # config 2
           "c5_256px_b8": (8, _layers(256, 5)),             # config 5: 256 x 256, iter_level 5, 64 images over 8 GPUs
           "c3_128px_b128": (128, _layers(128, 4))}         # batch sweep of bench.py
CASES = [(cfg, *ly) for cfg, (_, lys) in CONFIGS.items() for ly in lys]
B = 32


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


@pytest.mark.parametrize("cfg,name,Cb,Cs,Hs", CASES, ids=[f"{c[0]}-{c[1]}" for c in CASES])
@pytest.mark.parametrize("precision", ["bf16x3", "f32"])
def test_conv_families_are_adjoint_and_bilinear_at_full_size(cfg, name, Cb, Cs, Hs, precision):
    from vae_play_amd import ops
    if precision == "f32" and cfg != "c3_128px_b32":
        pytest.skip("the exact-f32 kernels are swept at the benchmark shard only")
    B = CONFIGS[cfg][0]
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
    if precision == "bf16x3":
        _check_epilogue_statistics(name, x, p0, p1, y, Cb, Cs, Hs, B, cx)


def _check_epilogue_statistics(name, x, p0, p1, y, Cb, Cs, Hs, B, cx):
    """vp_conv5_*_stats_bf16x3: same convolution output bit for bit, and mean / rstd / running statistics equal to the
    stand-alone statistics kernels' (which read the activation again) -- for every launch shape that can emit them."""
    from vae_play_amd import _lib, ops
    lib = _lib.load()
    for family in (0, 1):
        nbytes = lib.vp_conv5_stats_workspace_bytes(family, B, Hs, Hs, Cb, Cs, 2)
        if not nbytes:
            continue
        ws = torch.empty(nbytes // 4, device=DEV)
        if family == 0:
            ref_out, Cn, R = cx, Cs, B * Hs * Hs
            out = torch.empty_like(ref_out)
            inp, wq, name_ = ops.split_f32(x), p0, "vp_conv5_gather_stats_bf16x3"
            geom = (B, Hs, Hs, Cb, Cs, 2)
        else:
            ref_out, Cn, R = ops.conv5_scatter_bf16x3(ops.split_f32(y), y.shape, p1, Cb, 2), Cb, B * 4 * Hs * Hs
            out = torch.empty_like(ref_out)
            inp, wq, name_ = ops.split_f32(y), p1, "vp_conv5_scatter_stats_bf16x3"
            geom = (B, Hs, Hs, Cs, Cb, 2)
        mean, rstd = torch.empty(Cn, device=DEV), torch.empty(Cn, device=DEV)
        rm, rv = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
        _lib.call(name_, ops._pv(inp), ops._pv(wq), ops._p(out), *geom, 1e-5, 0.9, ops._p(mean), ops._p(rstd), ops._p(rm), ops._p(rv),
                  ops._p(ws), nbytes, ops._stream())
        # The epilogue changes no arithmetic of the kernel it runs in.  Round 3: a plain launch may take ANOTHER kernel than the
        # statistics launch of the same shape -- the pipelined kernel on the v_mfma_f32_16x16x32_bf16 form (conv16_impl.h plan16: the
        # gather shapes with 256 output columns and 16 K - 64 K rows), whose accumulator layout the epilogue does not read -- and that
        # form equals the 32x32x16 kernels to rounding, not bit for bit: the outputs must then agree to 2e-6 of their RMS.
        if not torch.equal(out, ref_out):
            rms = ref_out.double().pow(2).mean().sqrt().item()
            dmax = (out.double() - ref_out.double()).abs().max().item()
            assert dmax <= 2e-6 * max(rms, 1e-30) * 16, f"{name} family {family}: the statistics launch differs from the plain one by {dmax:.2e} (rms {rms:.2e})"
        rm2, rv2 = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
        mean2, rstd2 = ops.bn_stats(out, 1e-5, 0.9, rm2, rv2)
        xd = out.double()
        m64 = xd.mean(dim=(0, 2, 3)); v64 = xd.var(dim=(0, 2, 3), unbiased=False)
        sig = v64.sqrt()
        assert ((mean.double() - m64).abs() / sig).max().item() <= 1e-5, f"{name} family {family}: mean"
        assert ((rstd.double() - (v64 + 1e-5).rsqrt()).abs() * sig).max().item() <= 1e-5, f"{name} family {family}: rstd"
        assert ((mean - mean2).abs() / sig.float()).max().item() <= 1e-5 and ((rstd - rstd2).abs() * sig.float()).max().item() <= 1e-5
        assert ((rm - rm2).abs() / sig.float()).max().item() <= 1e-5 and ((rv - rv2).abs() / v64.float()).max().item() <= 1e-5, "running buffers"


@pytest.mark.parametrize("C,H", [(64, 64), (64, 128), (512, 8)])
def test_batchnorm_identities_at_full_size(C, H):
    from vae_play_amd import functional as FH
    B = 32
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
