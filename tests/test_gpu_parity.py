"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against
 (a) torch CPU ops / the oracle restatement on the same seeded inputs, and
 (b) the committed reference-generated golden vectors (tests/golden, written by oracle/gen_golden.py).
Tolerances: OP_RTOL for single ops (fp32 MFMA is exact-fp32 arithmetic), NORTH_STAR_RTOL = 1e-3
(BASELINE.json) for quantities that pass through the whole network.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import NORTH_STAR_RTOL, OP_RTOL, RAW_GRAD_L2, SAMPLE_FACTOR, assert_close, load_golden, record, rel_err, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(x):
    return x.to(DEV)


def nhwc_dev(x_nchw):
    return x_nchw.to(DEV).contiguous(memory_format=torch.channels_last)


# shapes chosen so every tile configuration of igemm.h is exercised (see choose_tile)
CONV_CASES = [  # (B, Hs, Cb, Cs, stride)
    (2, 4, 4, 8, 2),      # tiny, 64x64 tile, vector path
    (2, 8, 3, 8, 2),      # Cb=3 scalar path
    (1, 8, 1, 64, 2),     # Cb=1
    (3, 5, 8, 4, 2),      # odd sizes, M tail
    (2, 16, 64, 128, 2),  # 64x64 tiles, several K tiles
    (8, 32, 32, 512, 2),  # 128x128 tiles
    (6, 64, 16, 64, 2),   # 128x64 tiles
    (2, 16, 64, 3, 1),    # final-conv shape: N=3 -> 128x32 tiles, stride 1
    (2, 6, 4, 4, 1),
    (1, 8, 8, 3, 1),
    # edge-layer (narrow.hip) kernels: 1/3-channel side next to a 64-multiple side
    (2, 32, 64, 3, 1),    # final conv fwd + its wgrad (narrow small side)
    (3, 20, 128, 3, 1),   # non-multiple-of-16 image, two channel groups
    (2, 24, 64, 1, 1),    # single-channel output
    (2, 16, 3, 64, 2),    # first encoder conv wgrad (narrow big side, stride 2)
    (3, 9, 1, 128, 2),    # single-channel input, odd size
    (2, 8, 3, 64, 1),     # narrow big side at stride 1
]


@pytest.mark.parametrize("B,Hs,Cb,Cs,stride", CONV_CASES)
def test_conv5_gather_scatter_wgrad(B, Hs, Cb, Cs, stride):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(100 + B + Hs + Cb)
    Hb = Hs * stride
    big = torch.randn(B, Cb, Hb, Hb, generator=g)
    small = torch.randn(B, Cs, Hs, Hs, generator=g)
    w = torch.randn(Cs, Cb, 5, 5, generator=g) * 0.1
    bias = torch.randn(Cs, generator=g)
    p0, p1 = ops.pack_w5(dev(w), True, True)
    assert torch.equal(p0.cpu(), w.permute(0, 2, 3, 1).reshape(Cs, 25, Cb))
    assert torch.equal(p1.cpu(), w.permute(1, 2, 3, 0).reshape(Cb, 25, Cs))
    # F family == Conv2d forward
    y = ops.conv5_gather(nhwc_dev(big), p0, dev(bias), stride, 0)
    assert_close(y, F.conv2d(big, w, bias, stride=stride, padding=2), OP_RTOL, "gather")
    ys = ops.conv5_gather(nhwc_dev(big), p0, dev(bias), stride, 4)
    assert_close(ys, torch.sigmoid(F.conv2d(big, w, bias, stride=stride, padding=2)), OP_RTOL, "gather+sigmoid")
    # T family == ConvTranspose2d forward
    yt = ops.conv5_scatter(nhwc_dev(small), p1, stride)
    assert_close(yt, F.conv_transpose2d(small, w, None, stride=stride, padding=2, output_padding=stride - 1), OP_RTOL, "scatter")
    # W family == weight gradient
    wr = w.clone().requires_grad_(True)
    F.conv2d(big, wr, None, stride=stride, padding=2).backward(small)
    dw = ops.conv5_wgrad(nhwc_dev(big), nhwc_dev(small), stride)
    assert_close(dw, wr.grad, OP_RTOL, "wgrad")


@pytest.mark.parametrize("M,N,K", [(4, 16, 64), (32, 1024, 32768), (32, 128, 1024), (5, 7, 9), (130, 70, 33),
                                   (32, 32768, 128), (256, 512, 384),
                                   # skinny kernels (batch-sized M, weights streamed once): ragged M, N off the 128-column
                                   # block, odd tile counts per wave, split and unsplit K
                                   (7, 200, 320), (32, 132, 4096), (1, 1000, 192), (31, 4100, 64), (16, 256, 8192), (9, 320, 200), (3, 448, 132),
                                   # two blocks of 32 batch rows
                                   (64, 512, 4096), (48, 1024, 256), (33, 260, 192), (64, 256, 8192)])
def test_linear_forms(M, N, K):
    from vae_play_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    assert_close(ops.linear_fwd(dev(x), dev(W), dev(b)), F.linear(x, W, b), OP_RTOL, "linear fwd")
    assert_close(ops.linear_dgrad(dev(dy), dev(W)), dy @ W, OP_RTOL, "linear dgrad")
    assert_close(ops.linear_wgrad(dev(dy), dev(x)), dy.t() @ x, OP_RTOL, "linear wgrad")
    assert_close(ops.colsum(dev(dy)), dy.sum(0), OP_RTOL, "colsum")


@pytest.mark.parametrize("shape", [(4, 8, 16, 16), (2, 64, 8, 8), (32, 1024), (4, 8192), (3, 6, 5, 7), (16, 64, 64, 64)])
@pytest.mark.parametrize("act", ["relu", None, "lrelu", "tanh"])
def test_batchnorm_act_fwd_bwd(shape, act):
    from vae_play_amd import functional as FH
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    y = F.batch_norm(x, rm, rv, gamma, beta, True, 0.9, 1e-5)
    slope = 0.02
    y = {"relu": F.relu, None: lambda v: v, "lrelu": lambda v: F.leaky_relu(v, slope), "tanh": torch.tanh}[act](y)
    gy = torch.randn(shape, generator=g)
    y.backward(gy)
    xd = dev(x.detach()).requires_grad_(True)
    gd, bd = dev(gamma.detach()).requires_grad_(True), dev(beta.detach()).requires_grad_(True)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yd = FH.batch_norm_act(xd, gd, bd, rmd, rvd, True, 0.9, 1e-5, act, slope)
    yd.backward(dev(gy))
    tol = 1e-4
    assert_close(yd, y, tol, "bn y")
    assert_close(rmd, rm, tol, "running_mean")
    assert_close(rvd, rv, tol, "running_var")
    assert_close(xd.grad, x.grad, tol * 3, "bn dx")
    assert_close(gd.grad, gamma.grad, tol * 3, "dgamma")
    assert_close(bd.grad, beta.grad, tol * 3, "dbeta")
    # eval mode uses the running buffers
    ye = F.batch_norm(x.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.9, 1e-5)
    yde = FH.batch_norm_act(dev(x.detach()), gd.detach(), bd.detach(), rmd, rvd, False, 0.9, 1e-5, None)
    assert_close(yde, ye, tol, "bn eval")


@pytest.mark.parametrize("name,kind", [("encblock_4to8", "enc"), ("encblock_3to8", "enc"), ("encblock_1to64", "enc"),
                                       ("encblock_64to128", "enc"), ("decblock_8to4", "dec"), ("decblock_64to32", "dec"),
                                       ("decblock_128to128", "dec")])
def test_blocks_against_reference_golden(name, kind):
    """EncoderBlock / DecoderBlock forward + backward vs vectors produced by the reference classes."""
    import vae_play_amd as V
    g = load_golden(name)
    w = t(g["w"])
    if kind == "enc":
        blk = V.EncoderBlock(w.shape[1], w.shape[0])
    else:
        blk = V.DecoderBlock(w.shape[0], w.shape[1])
    with torch.no_grad():
        blk.conv.weight.copy_(w)
        blk.bn.weight.copy_(t(g["gamma"]))
        blk.bn.bias.copy_(t(g["beta"]))
    blk.to(DEV).train()
    x = dev(t(g["x"])).requires_grad_(True)
    y = blk(x)
    y.backward(dev(t(g["gy"])))
    tol = 1e-4
    assert_close(y, t(g["y"]), tol, "y")
    assert_close(x.grad, t(g["dx"]), tol * 3, "dx")
    assert_close(blk.conv.weight.grad, t(g["dw"]), tol * 3, "dw")
    assert_close(blk.bn.weight.grad, t(g["dgamma"]), tol * 3, "dgamma")
    assert_close(blk.bn.bias.grad, t(g["dbeta"]), tol * 3, "dbeta")
    assert_close(blk.bn.running_mean, t(g["running_mean"]), tol, "running_mean")
    assert_close(blk.bn.running_var, t(g["running_var"]), tol, "running_var")
    assert int(blk.bn.num_batches_tracked) == 1


def test_latent_against_reference_golden():
    import vae_play_amd as V
    g = load_golden("latent")
    mu, lv, eps = (dev(t(g[k])).requires_grad_(k != "eps") for k in ("mu", "logvar", "eps"))
    z = V.reparameterize(mu, lv, eps=eps)
    assert_close(z, t(g["z"]), OP_RTOL, "z")
    kl = V.kl_divergence(mu, lv)
    assert_close(kl, t(g["kl"]), OP_RTOL, "kl")
    # gradients vs torch autograd of the same formulas
    mu_c, lv_c = t(g["mu"]).requires_grad_(True), t(g["logvar"]).requires_grad_(True)
    gz = torch.randn(mu_c.shape, generator=torch.Generator().manual_seed(3))
    zc = t(g["eps"]) * torch.exp(0.5 * lv_c) + mu_c
    klc = -0.5 * torch.sum(-lv_c.exp() - mu_c.pow(2) + lv_c + 1, 1)
    ((zc * gz).sum() + klc.sum() * 0.7).backward()
    ((z * dev(gz)).sum() + kl.sum() * 0.7).backward()
    assert_close(mu.grad, mu_c.grad, OP_RTOL, "dmu")
    assert_close(lv.grad, lv_c.grad, OP_RTOL, "dlogvar")
    # default path draws eps on the device
    z2 = V.reparameterize(mu.detach(), lv.detach())
    assert z2.shape == mu.shape and torch.isfinite(z2).all()


@pytest.mark.parametrize("n", [(2, 3, 16, 16), (4, 1, 32, 32), (1, 3, 7, 5)])
def test_bce_matches_torch_including_clamp(n):
    from vae_play_amd import functional as FH
    g = torch.Generator().manual_seed(9)
    p = torch.rand(n, generator=g)
    p.view(-1)[0] = 0.0   # log clamp at -100 (torch semantics)
    p.view(-1)[1] = 1.0
    tt = torch.rand(n, generator=g)
    pc = p.clone().requires_grad_(True)
    ref = F.binary_cross_entropy(pc, tt, reduction="sum")
    ref.backward()
    pd = dev(p).requires_grad_(True)
    out = FH.binary_cross_entropy(pd, dev(tt), "sum")
    out.backward()
    assert_close(out.reshape(1), ref.detach().reshape(1), 1e-5, "bce sum")
    assert_close(pd.grad, pc.grad, 1e-5, "bce grad")


@pytest.mark.parametrize("kind", ["adam", "rmsprop"])
def test_flat_optimizer_matches_torch(kind):
    from vae_play_amd import optim
    g = torch.Generator().manual_seed(5)
    shapes = [(7,), (64, 3, 5, 5), (130, 33), (1,)]
    ps_ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    ps = [torch.nn.Parameter(dev(p.detach().clone())) for p in ps_ref]
    if kind == "adam":
        o_ref, o = torch.optim.Adam(ps_ref, lr=1e-4), optim.Adam(ps, lr=1e-4)
    else:
        o_ref, o = torch.optim.RMSprop(ps_ref, lr=1e-4), optim.RMSprop(ps, lr=1e-4)
    for step in range(3):
        o.zero_grad()
        for pr, pd in zip(ps_ref, ps):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad = gr.clone()
            pd.grad.copy_(dev(gr))
        o_ref.step(); o.step()
        for pr, pd in zip(ps_ref, ps):
            d = (pd.detach().cpu() - pr.detach()).abs().max().item()
            assert d <= 2e-7 * max(1.0, pr.abs().max().item()), f"{kind} step {step}: {d}"


def _build_vae_from_oracle_params(C, S, z):
    import vae_play_amd as V
    from oracle import ref_cpu as O
    L = O.iter_level_for(S)
    p0 = O.init_params(C, z, L, seed=0)
    vae = V.VAE(S, z, C, init_rule=False)
    missing = vae.load_state_dict(p0, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return vae.to(DEV).train(), p0, L


STEP_FIXTURES = ["step_32x32x1_z16_b4_adam", "step_32x32x1_z16_b4_rmsprop", "step_64x64x3_z64_b4_adam",
                 "step_128x128x3_z128_b4_adam", "step_128x128x3_z128_b32_adam"]


@pytest.mark.parametrize("name", STEP_FIXTURES)
def test_train_step_against_reference_golden(name):
    """The composed step on the drop-in modules + flat optimiser vs the vectors the
    REAL reference produced: mu/logvar/z/x_tilde, loss, every gradient, BN running stats, and the
    parameters after 1..3 optimiser steps."""
    import vae_play_amd as V
    from vae_play_amd import optim
    from oracle import ref_cpu as O
    g = load_golden(name)
    C, S, z, B = (int(g[k]) for k in ("meta_C", "meta_S", "meta_z", "meta_B"))
    steps, kind = int(g["meta_steps"]), str(g["meta_optim"])
    vae, p0, L = _build_vae_from_oracle_params(C, S, z)
    x, eps = O.synthetic_batch(B, C, S, z)
    assert abs(O.checksum(x)["sum"].item() - g["x_sum"][0]) < 1e-9
    xd, epsd = dev(x), dev(eps)
    opt = (optim.Adam if kind == "adam" else optim.RMSprop)(vae.parameters(), lr=1e-4)
    names = [n for n, _ in vae.named_parameters()]
    for step in range(1, steps + 1):
        opt.zero_grad()
        xt, mu, lv = vae(xd, eps=epsd)
        loss, recon, kl = V.vae_loss(xd, xt, mu, lv)
        loss.backward()
        if step == 1:
            tol = NORTH_STAR_RTOL
            errs = {"mu": assert_close(mu, t(g["mu"]), tol, "mu"), "logvar": assert_close(lv, t(g["logvar"]), tol, "logvar")}
            if "x_tilde" in g:
                errs["x_tilde"] = assert_close(xt, t(g["x_tilde"]), tol, "x_tilde")
            errs["x_tilde_s7"] = assert_close(xt.detach().cpu().contiguous().flatten()[::7][:8192], t(g["x_tilde_stride7"]), tol, "x_tilde[::7]")
            for k, v in (("loss", loss), ("recon", recon), ("kl", kl)):
                assert abs(v.item() - g[k][0]) <= tol * abs(g[k][0]), f"{k}: {v.item()} vs {g[k][0]}"
            params = dict(vae.named_parameters())
            worst = 0.0
            for n in names:
                gr = params[n].grad.detach().cpu().contiguous()
                l2 = g[f"grad_l2/{n}"][0]
                idx = O.sample_indices(gr.numel())
                d = (gr.flatten()[idx].double() - t(g[f"grad_samples/{n}"])).abs().max().item()
                # relative to the gradient tensor's RMS-scale: l2/sqrt(numel)
                scale = max(l2 / gr.numel() ** 0.5, 1e-12)
                # (exact-fp32 modules: SAMPLE_FACTOR x the mode's asserted raw l2 bound = 0.03 of the tensor's RMS, tests/util.py)
                sb = SAMPLE_FACTOR["f32"] * RAW_GRAD_L2["f32"]
                worst = max(worst, d / scale / sb)
                record(f"grad_sample_over_rms/{n}", d / scale)
                assert d <= sb * scale, f"grad samples {n}: {d} vs scale {scale}"
                record(f"grad_l2_rel/{n}", abs(gr.double().pow(2).sum().sqrt().item() - l2) / (l2 + 1e-30))
            sd = vae.state_dict()
            for k in g:
                if k.startswith("bn/"):
                    n = k[3:]
                    assert_close(sd[n].flatten()[:4096], t(g[k]).flatten(), tol, f"running stat {n}")
            print(f"{name}: rel errs {errs}, worst grad-sample err/budget {worst:.2e}")
        opt.step()
        params = dict(vae.named_parameters())
        for n in names:
            pv = params[n].detach().cpu().contiguous()
            l2 = g[f"param{step}_l2/{n}"][0]
            # a parameter moves by <= lr per Adam step: compare norms to 1e-5 of the norm + 2% of the update size
            upd = 1e-4 * step * pv.numel() ** 0.5
            assert abs(pv.double().pow(2).sum().sqrt().item() - l2) <= 1e-5 * l2 + 0.02 * upd, f"param l2 after step {step}: {n}"
            idx = O.sample_indices(pv.numel())
            dd = (pv.flatten()[idx].double() - t(g[f"param{step}_samples/{n}"])).abs()
            # Adam moves a weight by ~lr*sign(g) per step: elements whose gradient is ~0 may legitimately take
            # the other sign under a different fp32 summation order, so allow a few outliers among the 16
            # samples but bound every deviation by the total possible movement (2*lr*step).
            assert (dd > 0.05 * 1e-4 * step + 1e-7).sum().item() <= max(2, dd.numel() // 8), f"param samples after step {step}: {n}"
            assert dd.max().item() <= 2.05 * 1e-4 * step, f"param sample moved more than 2*lr*step: {n}"
        assert abs(loss.item() - g[f"loss_step{step}"][0]) <= NORTH_STAR_RTOL * abs(g[f"loss_step{step}"][0])


def test_train_step_full_tensors_vs_oracle_and_determinism():
    """Full-tensor comparison with the oracle at 32x32x3 (every gradient element), and run-to-run
    bit reproducibility of the HIP path (slab reductions, no atomics)."""
    import vae_play_amd as V
    from oracle import ref_cpu as O
    C, S, z, B = 3, 32, 16, 8
    vae, p0, L = _build_vae_from_oracle_params(C, S, z)
    x, eps = O.synthetic_batch(B, C, S, z)
    p = O.clone_params(p0); O.require_grad(p)
    ref = O.train_step(p, None, x, eps, L)

    def run():
        for q in vae.parameters():
            q.grad = None
        xt, mu, lv = vae(dev(x), eps=dev(eps))
        loss, _, _ = V.vae_loss(dev(x), xt, mu, lv)
        loss.backward()
        return xt.detach().clone(), loss.detach().clone(), {n: q.grad.detach().clone() for n, q in vae.named_parameters()}

    sd0 = {k: v.clone() for k, v in vae.state_dict().items()}
    xt1, loss1, g1 = run()
    assert_close(xt1, ref["x_tilde"], NORTH_STAR_RTOL, "x_tilde")
    assert abs(loss1.item() - ref["loss"].item()) <= NORTH_STAR_RTOL * abs(ref["loss"].item())
    for n in O.trainable_names(p):
        assert_close(g1[n], p[n].grad, NORTH_STAR_RTOL, f"grad {n}")
    vae.load_state_dict(sd0)
    xt2, loss2, g2 = run()
    assert torch.equal(xt1, xt2) and torch.equal(loss1, loss2)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), f"non-deterministic gradient {n}"


def test_eval_mode_and_encoder_block_tap():
    import vae_play_amd as V
    torch.manual_seed(0)
    blk = V.EncoderBlock(4, 8).to(DEV)
    x = torch.randn(2, 4, 16, 16, device=DEV)
    y, tap = blk(x, out=True)
    ref = F.conv2d(x.cpu(), blk.conv.weight.detach().cpu(), None, stride=2, padding=2)
    assert_close(tap, ref, OP_RTOL, "pre-BN tap")
    blk.eval()
    ye = blk(x)
    rm, rv = blk.bn.running_mean.cpu(), blk.bn.running_var.cpu()
    refe = F.relu(F.batch_norm(ref, rm, rv, blk.bn.weight.detach().cpu(), blk.bn.bias.detach().cpu(), False, 0.9, 1e-5))
    assert_close(ye, refe, 1e-4, "eval forward")
    assert int(blk.bn.num_batches_tracked) == 1


def test_ops_refuse_missing_gpu_tensors():
    from vae_play_amd import _lib, ops
    with pytest.raises(_lib.VaePlayHipError):
        ops.linear_fwd(torch.zeros(2, 4), torch.zeros(3, 4), None)
