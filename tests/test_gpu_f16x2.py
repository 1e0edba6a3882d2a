"""The "f16x2" contraction mode (fp16-pair operands, two MFMAs per fragment pair): DECLARED TOLERANCE 3e-4 relative L2 per layer
against fp64 (measured ~1e-4: the operand that keeps only its fp16 hi plane is rounded at 2^-12 rms), against 5e-6 for bf16x3.
Checked: the three convolution families on the benchmark's layer shapes vs torch's fp64 convolutions on the CPU, the power-of-two
gradient scale (tiny operands survive fp16's range), saturation instead of overflow, and that every *_fmt producer with format 0 is
bit-identical to its plain entry point."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 3e-4


def _rel(a, ref):
    return ((a.double().cpu() - ref).pow(2).sum().sqrt() / ref.pow(2).sum().sqrt()).item()


def _cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("B,Cb,Cs,Hs", [(4, 64, 128, 16), (2, 128, 256, 8), (8, 256, 512, 4), (4, 32, 64, 32), (2, 40, 24, 6)])
@pytest.mark.parametrize("gmag", [1.0, 3e-7])
def test_three_families_against_fp64(B, Cb, Cs, Hs, gmag):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cb + Cs + Hs)
    Hb = 2 * Hs
    x = torch.randn(B, Cb, Hb, Hb, generator=g) * 0.7 + 0.2
    y = torch.randn(B, Cs, Hs, Hs, generator=g) * gmag            # a gradient-like operand: tiny at gmag = 3e-7
    w = torch.randn(Cs, Cb, 5, 5, generator=g) * 0.03
    S = 1.0 if gmag == 1.0 else float(2 ** 14)
    p0, p1 = ops.pack_w5_split(w.to(DEV), True, True, ops.SPLIT_F16)
    xs = ops.split_f32(_cl(x), ops.SPLIT_F16)
    ys = ops.split_f32(_cl(y), ops.SPLIT_F16, S)
    # gather = Conv2d forward
    out = ops.conv5_gather_f16(xs, x.shape, p0, Cs, None, 2, 0, 2)
    ref = F.conv2d(x.double(), w.double(), None, 2, 2)
    e_f = _rel(out, ref)
    # scatter = its input gradient (ConvTranspose2d forward), operand scaled by S, accumulators by 1/S
    out_t = ops.conv5_scatter_f16(ys, y.shape, p1, Cb, 2, 2, 1.0 / S)
    ref_t = F.conv_transpose2d(y.double(), w.double(), None, 2, 2, 1)
    e_t = _rel(out_t, ref_t)
    # weight gradient
    dw = ops.conv5_wgrad_f16x2(xs, tuple(x.shape), ys, tuple(y.shape), 2, 1.0 / S)
    xd = x.double().requires_grad_(False)
    wd = w.double().requires_grad_(True)
    (F.conv2d(xd, wd, None, 2, 2) * y.double()).sum().backward()
    e_w = _rel(dw, wd.grad)
    assert e_f <= TOL and e_t <= TOL and e_w <= TOL, f"gather {e_f:.2e} scatter {e_t:.2e} wgrad {e_w:.2e}"
    # ... and it IS the cheaper mode: the lo plane of the A operand is used (an fp16-only product would sit at ~4e-4 .. 6e-4)
    assert e_f <= 2.5e-4 and e_t <= 2.5e-4
    # three products on the same planes (the forward layers of an "f16x2" plan): fp32-level
    e3 = _rel(ops.conv5_gather_f16(xs, x.shape, p0, Cs, None, 2, 0, 3), ref)
    e3t = _rel(ops.conv5_scatter_f16(ys, y.shape, p1, Cb, 2, 3, 1.0 / S), ref_t)
    # (a tiny gradient operand scaled by 2^14 has a subnormal lo plane: ~1e-5 per element there)
    assert e3 <= 2e-6 and e3t <= (2e-6 if gmag == 1.0 else 1e-5), f"three products: gather {e3:.2e} scatter {e3t:.2e}"


def test_unscaled_tiny_gradients_lose_precision_and_scale_restores_it():
    """Why the producer scale exists: 3e-7-sized gradients are fp16 subnormals (absolute step 6e-8)."""
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(5)
    y = torch.randn(4, 128, 16, 16, generator=g) * 3e-7
    ys0 = ops.split_f32(_cl(y), ops.SPLIT_F16, 1.0)
    ys1 = ops.split_f32(_cl(y), ops.SPLIT_F16, float(2 ** 14))
    yc = y.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1).double()
    e0 = _rel(ops.unsplit(ys0, ops.SPLIT_F16), yc)
    e1 = _rel(ops.unsplit(ys1, ops.SPLIT_F16) / 2 ** 14, yc)
    assert e0 > 1e-2 and e1 < 1e-5, (e0, e1)


def test_split_saturates_instead_of_overflowing():
    from vae_play_amd import ops
    x = torch.tensor([1e6, -1e6, 65504.0, 3.0, 0.0, -0.0, 1e-3, 7e4], device=DEV)
    s = ops.unsplit(ops.split_f32(x, ops.SPLIT_F16), ops.SPLIT_F16)
    assert torch.isfinite(s).all()
    assert s.tolist()[:3] == [65504.0, -65504.0, 65504.0] and s[7].item() == 65504.0
    assert abs(s[3].item() - 3.0) == 0 and abs(s[6].item() - 1e-3) < 4e-8      # lo is subnormal there: absolute step 2^-24


def test_fmt_producers_with_format_0_are_bit_identical():
    from vae_play_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    x = _cl(torch.randn(4, 64, 16, 16, generator=g))
    a = ops.split_f32(x)
    b = ops.empty_split(x.numel(), x)
    _lib.call("vp_split_fmt_f32", ops._p(x), ops._pv(b), x.numel(), 0, 1.0, ops._stream())
    assert torch.equal(a, b)
    # BatchNorm + ReLU forward / backward with split outputs
    R, C = 4 * 16 * 16, 64
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    mean, rstd = ops.bn_stats(x, 1e-5, 0.9)
    outs = []
    for fmt_call in (False, True):
        y, ysp = torch.empty_like(x), ops.empty_split(x.numel(), x)
        if fmt_call:
            _lib.call("vp_bn_act_fwd_split_fmt_f32", ops._p(x), ops._p(mean), ops._p(rstd), ops._p(gamma), ops._p(beta), ops._p(y), ops._pv(ysp),
                      R, C, 1, 0.0, 0, ops._stream())
        else:
            _lib.call("vp_bn_act_fwd_split_f32", ops._p(x), ops._p(mean), ops._p(rstd), ops._p(gamma), ops._p(beta), ops._p(y), ops._pv(ysp),
                      R, C, 1, 0.0, ops._stream())
        outs.append((y, ysp))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dy = _cl(torch.randn(4, 64, 16, 16, generator=g))
    nb = lib.vp_bn_workspace_bytes(R, C)
    outs = []
    for fmt_call in (False, True):
        ws = torch.empty(nb // 4, device=DEV)
        dx, dxs, dga, dbe = torch.empty_like(x), ops.empty_split(x.numel(), x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        args = [ops._p(x), ops._p(dy), ops._p(mean), ops._p(rstd), ops._p(gamma), ops._p(beta), ops._p(dx), ops._pv(dxs), ops._p(dga),
                ops._p(dbe), R, C, 1, 0.0, 1]
        if fmt_call:
            _lib.call("vp_bn_act_bwd_split_fmt_f32", *args, 0, 1.0, ops._p(ws), nb, ops._stream())
        else:
            _lib.call("vp_bn_act_bwd_split_f32", *args, ops._p(ws), nb, ops._stream())
        outs.append((dx, dxs, dga, dbe))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    # ... and with format 1 + scale the planes hold scale * dx
    ws = torch.empty(nb // 4, device=DEV)
    dxs = ops.empty_split(x.numel(), x)
    _lib.call("vp_bn_act_bwd_split_fmt_f32", ops._p(x), ops._p(dy), ops._p(mean), ops._p(rstd), ops._p(gamma), ops._p(beta), None, ops._pv(dxs),
              None, None, R, C, 1, 0.0, 1, 1, 1024.0, ops._p(ws), nb, ops._stream())
    got = ops.unsplit(dxs, ops.SPLIT_F16) / 1024.0
    ref = outs[0][0].permute(0, 2, 3, 1).reshape(-1)
    assert ((got - ref).abs().max() / ref.abs().max()).item() < 1e-5


def test_engine_reports_saturated_gradient_planes():
    """FusedVAEStep(precision="f16x2").f16_saturated(): zero at the default gradient scale, non-zero when the scale is absurd."""
    from tests.test_gpu_engine import build
    from oracle import ref_cpu as O
    import vae_play_amd as V
    from vae_play_amd import engine, optim
    C, S, z, B = 3, 32, 16, 8
    x, eps = O.synthetic_batch(B, C, S, z)
    vae, opt, fused, p0, L = build(C, S, z, B, precision="f16x2")
    fused.forward_backward(x.to(DEV), eps.to(DEV))
    assert fused.f16_saturated() == 0
    fused.sync_counters()                                  # the sticky device flag is clear: no exception
    g_ok = {n: q.grad.detach().clone() for n, q in vae.named_parameters()}
    vae2 = V.VAE(S, z, C, init_rule=False)
    vae2.load_state_dict(p0)
    vae2.to(DEV).train()
    opt2 = optim.Adam(vae2.parameters(), lr=1e-4)
    bad = engine.FusedVAEStep(vae2, opt2, B, S, C, precision="f16x2", grad_scale16=float(2 ** 40))
    bad.forward_backward(x.to(DEV), eps.to(DEV))
    assert bad.f16_saturated() > 0
    # ... and the producers raised the device-side flag: sync_counters() reports it once, then the flag is clear again
    from vae_play_amd._lib import VaePlayHipError
    with pytest.raises(VaePlayHipError, match="saturated"):
        bad.sync_counters()
    bad.sync_counters()
    # ... and a smaller power of two changes nothing but rounding
    vae3 = V.VAE(S, z, C, init_rule=False)
    vae3.load_state_dict(p0)
    vae3.to(DEV).train()
    opt3 = optim.Adam(vae3.parameters(), lr=1e-4)
    low = engine.FusedVAEStep(vae3, opt3, B, S, C, precision="f16x2", grad_scale16=256.0)
    low.forward_backward(x.to(DEV), eps.to(DEV))
    assert low.f16_saturated() == 0
    for n, q in vae3.named_parameters():
        a, b = q.grad.double(), g_ok[n].double()
        assert ((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt()).item() <= 2e-3, n
    with pytest.raises(ValueError):
        engine.FusedVAEStep(vae3, opt3, B, S, C, precision="f16x2", grad_scale16=1000.0)
