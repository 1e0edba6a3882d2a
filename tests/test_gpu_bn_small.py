"""Single-launch BatchNorm for a handful of rows (csrc/bn.hip bn_small_*): the dense layers' nn.BatchNorm1d + ReLU
(models/networks.py:66-67,89-90) in the fused step."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,C", [(32, 1024), (4, 32768), (64, 72), (7, 8), (1, 64)])
def test_bn_small_is_bit_identical_to_the_three_launch_path(R, C):
    """vp_bn_small_fwd_f32 / vp_bn_small_bwd_f32 (the dense layers' BatchNorm1d in one launch) against vp_bn_stats_f32 +
    vp_bn_act_fwd_f32 and vp_bn_act_bwd_f32: same arithmetic step for step, so every output must be bit-identical."""
    from vae_play_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(R * 131 + C)
    x = (torch.randn(R, C, generator=g) * 1.5 + torch.randn(1, C, generator=g) * 20).cuda()       # |mean| >> sigma: the hard case
    dy = torch.randn(R, C, generator=g).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    P = ops._p
    nb = lib.vp_bn_workspace_bytes(R, C)
    ws = torch.empty(max(4, nb // 4), device="cuda")
    # three launches
    m1, r1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rm1, rv1 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y1 = torch.empty_like(x)
    _lib.call("vp_bn_stats_f32", P(x), R, C, 1e-5, 0.9, P(m1), P(r1), P(rm1), P(rv1), P(ws), ws.numel() * 4, ops._stream())
    _lib.call("vp_bn_act_fwd_f32", P(x), P(m1), P(r1), P(gamma), P(beta), P(y1), R, C, 1, 0.0, ops._stream())
    dx1, dg1, db1 = torch.empty_like(x), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    _lib.call("vp_bn_act_bwd_f32", P(x), P(dy), P(m1), P(r1), P(gamma), P(beta), P(dx1), P(dg1), P(db1), R, C, 1, 0.0, 1, P(ws),
              ws.numel() * 4, ops._stream())
    # one launch each
    m2, r2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    rm2, rv2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y2 = torch.full_like(x, float("nan"))
    _lib.call("vp_bn_small_fwd_f32", P(x), R, C, 1e-5, 0.9, P(gamma), P(beta), P(m2), P(r2), P(rm2), P(rv2), P(y2), 1, 0.0, ops._stream())
    dx2 = torch.full_like(x, float("nan"))
    dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    _lib.call("vp_bn_small_bwd_f32", P(x), P(dy), P(m2), P(r2), P(gamma), P(beta), P(dx2), P(dg2), P(db2), R, C, 1, 0.0, 1, ops._stream())
    for a, b, what in ((m1, m2, "mean"), (r1, r2, "rstd"), (rm1, rm2, "running_mean"), (rv1, rv2, "running_var"), (y1, y2, "y"),
                       (dx1, dx2, "dx"), (dg1, dg2, "dgamma"), (db1, db2, "dbeta")):
        assert torch.equal(a, b), what
    with pytest.raises(_lib.VaePlayHipError):
        _lib.call("vp_bn_small_fwd_f32", P(x), 65, C, 1e-5, 0.9, P(gamma), P(beta), P(m2), P(r2), None, None, P(y2), 1, 0.0, ops._stream())


@pytest.mark.parametrize("R,C", [(786432, 32), (5000, 8), (1000, 60), (70, 24), (65, 36), (300, 3), (200, 128)])
def test_colsum_variants_against_fp64(R, C):
    """vp_colsum_f32 (bias gradients): every dispatch branch -- the 16-B row form for narrow C % 4 == 0, the row-lane form, the wide
    form -- against an fp64 column sum."""
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(R + C)
    x = (torch.randn(R, C, generator=g) + 0.1).cuda()
    got = ops.colsum(x)
    ref = x.double().sum(dim=0)
    scale = x.double().abs().sum(dim=0)
    assert ((got.double() - ref).abs() / scale).max().item() <= 2e-6
    assert torch.equal(got, ops.colsum(x)), "column sums must be deterministic"
