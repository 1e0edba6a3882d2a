"""vp_instnorm_act_{fwd,bwd}_split_f32: nn.InstanceNorm2d + activation (models/blocks.py:22-30) with the bf16 hi/lo planes of the
result written by the same pass.  The fp32 results must be the BITS of the plain entry points, the planes the bits of a split pass
over them (what the split-bf16 convolution behind the layer would otherwise launch); and the autograd front end must train the
same numbers with and without the fused planes."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("B,C,H,W,act", [(3, 16, 8, 8, 1), (2, 64, 32, 32, 2), (5, 8, 7, 9, 0), (1, 40, 16, 16, 1)])
def test_split_variants_equal_plain_plus_split_pass(B, C, H, W, act):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(B * 100 + C)
    x = torch.randn(B, C, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, C, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y0, m0, r0 = ops.instnorm_act_fwd(x, 1e-5, act, 0.2)
    y1, m1, r1, ys = ops.instnorm_act_fwd(x, 1e-5, act, 0.2, want_split=True)
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(r0, r1)
    assert torch.equal(ys, ops.split_f32(y0))
    d0 = ops.instnorm_act_bwd(x, dy, m0, r0, act, 0.2)
    d1, ds = ops.instnorm_act_bwd(x, dy, m0, r0, act, 0.2, want_split=True)
    assert torch.equal(d0, d1)
    assert torch.equal(ds, ops.split_f32(d0))


def test_front_end_uses_the_fused_planes_and_trains_the_same_numbers(monkeypatch):
    """conv3x3 -> InstanceNorm + ReLU -> conv3x3 in bf16x3 mode: the second convolution and the first one's backward find their operand
    planes on the tensors (no split pass), results identical to the unfused path."""
    import vae_play_amd as V
    from vae_play_amd import functional as F, ops
    from vae_play_amd.blocks import Conv2d
    torch.manual_seed(0)
    # (channel counts outside csrc/small3.hip's few-channel kernels, which need no operand planes at all)
    net = torch.nn.Sequential(Conv2d(16, 48, 3, bn="instance", activate="relu"), Conv2d(48, 16, 3, bn="instance", activate="relu")).to(DEV)
    x = torch.randn(2, 16, 16, 16, device=DEV)
    calls = {"n": 0}
    orig = ops.split_f32

    def counted(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    monkeypatch.setattr(ops, "split_f32", counted)
    V.set_conv_precision("bf16x3")
    try:
        res = {}
        for fused in (True, False):
            monkeypatch.setattr(F, "_BWD_SPLIT", fused)
            calls["n"] = 0
            net.zero_grad()
            y = net(x)
            y.square().sum().backward()
            res[fused] = (y.detach().clone(), [p.grad.detach().clone() for p in net.parameters()], calls["n"])
    finally:
        V.set_conv_precision("f32")
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    assert res[True][2] <= res[False][2] - 2, (res[True][2], res[False][2])
