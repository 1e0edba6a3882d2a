"""GPU tests of the fused training step (vae_play_amd/engine.py): it must reproduce the golden
reference vectors and be interchangeable with the autograd modules (same gradients), eagerly and
when replayed from a captured hipGraph."""
import pytest
import torch

from tests.util import RAW_GRAD_L2, SAMPLE_FACTOR, NORTH_STAR_RTOL, assert_close, load_golden, record, t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(C, S, z, B, kind="adam", precision="f32"):
    import vae_play_amd as V
    from vae_play_amd import engine, optim
    from oracle import ref_cpu as O
    L = O.iter_level_for(S)
    p0 = O.init_params(C, z, L, seed=0)
    vae = V.VAE(S, z, C, init_rule=False)
    vae.load_state_dict(p0)
    vae.to(DEV).train()
    opt = (optim.Adam if kind == "adam" else optim.RMSprop)(vae.parameters(), lr=1e-4)
    step = engine.FusedVAEStep(vae, opt, B, S, C, precision=precision)
    return vae, opt, step, p0, L


@pytest.mark.parametrize("name", ["step_32x32x1_z16_b4_adam", "step_64x64x3_z64_b4_adam", "step_128x128x3_z128_b4_adam",
                                  "step_128x128x3_z128_b32_adam"])
@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x2"])
def test_fused_step_against_reference_golden(name, precision):
    from oracle import ref_cpu as O
    g = load_golden(name)
    C, S, z, B = (int(g[k]) for k in ("meta_C", "meta_S", "meta_z", "meta_B"))
    steps = int(g["meta_steps"])
    vae, opt, fused, p0, L = build(C, S, z, B, precision=precision)
    x, eps = O.synthetic_batch(B, C, S, z)
    xd, epsd = x.to(DEV), eps.to(DEV)
    names = [n for n, _ in vae.named_parameters()]
    for step in range(1, steps + 1):
        loss, recon, kl = fused.forward_backward(xd, epsd)
        if step == 1:
            tol = NORTH_STAR_RTOL
            assert_close(fused.mu, t(g["mu"]), tol, "mu")
            assert_close(fused.logvar, t(g["logvar"]), tol, "logvar")
            assert_close(fused.z, t(g["z"]), tol, "z")
            xt = fused.x_tilde.detach().cpu().contiguous()
            assert_close(xt.flatten()[::7][:8192], t(g["x_tilde_stride7"]), tol, "x_tilde")
            for k, v in (("loss", loss), ("recon", recon), ("kl", kl)):
                assert abs(v.item() - g[k][0]) <= tol * abs(g[k][0]), f"{k}: {v.item()} vs {g[k][0]}"
            params = dict(vae.named_parameters())
            for n in names:
                gr = params[n].grad.detach().cpu().contiguous()
                l2 = g[f"grad_l2/{n}"][0]
                gerr = record(f"grad_l2_rel/{n}", abs(gr.double().pow(2).sum().sqrt().item() - l2) / (l2 + 1e-30))
                # BASELINE's 1e-3 bar is on the outputs (x_tilde, mu/logvar, loss; asserted above for both
                # precisions).  Gradients are held to 1e-3 in exact-fp32 mode; in bf16x3 mode the ~5e-6
                # contraction noise is amplified along the backward chain by small-batch BatchNorm,
                # so each gradient tensor's norm is held to 5e-3 (measured worst: 2.6e-3 at batch 4).
                # The batch-4 fixtures amplify even fp32 rounding-ORDER noise the same way (BatchNorm1d over 4 samples):
                # two dense kernels that are both 2e-7 from the fp64 product (round-1 measurement, profiles/r01_c_notes.md) moved
                # encoder.l_var.bias from 4.8e-4 to 1.7e-3, so exact-fp32 mode keeps the 1e-3 bar at batch 32 only.
                # f16x2 (forward on three fp16 products, backward on two: ~2e-4 of contraction noise per backward layer) is held
                # to bf16x3's budgets (measured worst gradient norm 9.4e-4); its per-tensor accuracy claim is
                # tests/test_gpu_grad_accuracy.py's (fp64 oracle, same ReLU masks).
                budget = tol if (precision == "f32" and B >= 32) else 5e-3
                assert gerr <= budget, f"grad l2 {n}: {gerr:.2e} > {budget:.0e}"
                idx = O.sample_indices(gr.numel())
                d = (gr.flatten()[idx].double() - t(g[f"grad_samples/{n}"])).abs().max().item()
                # samples: SAMPLE_FACTOR x the mode's ASSERTED raw l2 bound (tests/util.py, tests/test_gpu_grad_accuracy.py), in units of
                # the tensor's RMS -- 0.03 RMS in exact fp32, 0.15 RMS in the split modes
                rms = max(l2 / gr.numel() ** 0.5, 1e-12)
                record(f"grad_sample_over_rms/{n}", d / rms)
                assert d <= SAMPLE_FACTOR[precision] * RAW_GRAD_L2[precision] * rms, f"grad samples {n}: {d / rms:.2e} RMS"
            sd = vae.state_dict()
            for k in g:
                if k.startswith("bn/"):
                    assert_close(sd[k[3:]].flatten()[:4096], t(g[k]).flatten(), tol, k)
        opt.step()
        params = dict(vae.named_parameters())
        for n in names:
            l2 = g[f"param{step}_l2/{n}"][0]
            pv = params[n].detach().cpu().double()
            upd = 1e-4 * step * pv.numel() ** 0.5
            slack = {"f32": 0.02, "bf16x3": 0.06, "f16x2": 0.06}[precision]
            assert abs(pv.pow(2).sum().sqrt().item() - l2) <= 1e-5 * l2 + slack * upd, f"param l2 step {step} {n}"
    fused.sync_counters()
    assert int(vae.encoder.conv[0].bn.num_batches_tracked) == steps


def test_fused_step_equals_autograd_modules_and_graph_replay():
    import vae_play_amd as V
    from oracle import ref_cpu as O
    C, S, z, B = 3, 32, 16, 8
    vae, opt, fused, p0, L = build(C, S, z, B, precision="f32")
    x, eps = O.synthetic_batch(B, C, S, z)
    xd, epsd = x.to(DEV), eps.to(DEV)
    sd0 = {k: v.clone() for k, v in vae.state_dict().items()}
    # autograd path
    opt.zero_grad()
    xt, mu, lv = vae(xd, eps=epsd)
    loss, _, _ = V.vae_loss(xd, xt, mu, lv)
    loss.backward()
    g_auto = opt.flat_grad.clone()
    rm_auto = vae.decoder.conv[0].bn.running_mean.clone()
    # fused eager
    vae.load_state_dict(sd0)
    opt.flat_grad.fill_(float("nan"))       # every gradient element must be overwritten
    loss_f, _, _ = fused.forward_backward(xd, epsd)
    g_fused = opt.flat_grad.clone()
    used = torch.zeros_like(g_fused, dtype=torch.bool)
    for p, o in zip(opt.arena.params, opt.arena.offsets):
        used[o:o + p.numel()] = True
    assert torch.isfinite(g_fused[used]).all(), "a gradient tensor was not written by the fused step"
    assert_close(g_fused[used], g_auto[used], 1e-5, "fused vs autograd grads")
    assert abs(loss_f.item() - loss.item()) <= 1e-6 * abs(loss.item())
    assert_close(vae.decoder.conv[0].bn.running_mean, rm_auto, 1e-6, "running stats")
    # graph replay: same numbers, BN buffers untouched by the capture itself
    vae.load_state_dict(sd0)
    fused.capture()
    assert torch.equal(vae.decoder.conv[0].bn.running_mean, sd0["decoder.conv.0.bn.running_mean"])
    opt.flat_grad.fill_(float("nan"))
    loss_g, _, _ = fused.forward_backward(xd, epsd)
    assert torch.equal(opt.flat_grad[used], g_fused[used]), "graph replay differs from eager launch"
    assert loss_g.item() == loss_f.item()
    # full step runs (single process: no collective); alignment padding of the arena is never read back
    opt.flat_grad[~used] = 0
    fused.step(xd, epsd)
    assert torch.isfinite(opt.flat_param[used]).all()


def test_autograd_modules_in_bf16x3_mode_match_golden():
    """vae_play_amd.set_conv_precision("bf16x3"): the drop-in modules run their 5x5 convolutions on the split-bf16
    kernels (inputs split on the fly); outputs stay inside the 1e-3 bar, gradients inside the bf16x3 budget."""
    import vae_play_amd as V
    from oracle import ref_cpu as O
    from tests.util import load_golden, t, NORTH_STAR_RTOL
    g = load_golden("step_64x64x3_z64_b4_adam")
    C, S, z, B = (int(g[k]) for k in ("meta_C", "meta_S", "meta_z", "meta_B"))
    L = O.iter_level_for(S)
    vae = V.VAE(S, z, C, init_rule=False)
    vae.load_state_dict(O.init_params(C, z, L, seed=0), strict=True)
    vae = vae.to("cuda").train()
    x, eps = O.synthetic_batch(B, C, S, z)
    V.set_conv_precision("bf16x3")
    try:
        xt, mu, lv = vae(x.cuda(), eps=eps.cuda())
        loss, recon, kl = V.vae_loss(x.cuda(), xt, mu, lv)
        loss.backward()
    finally:
        V.set_conv_precision("f32")
    assert (mu.detach().cpu() - t(g["mu"])).abs().max().item() <= NORTH_STAR_RTOL * t(g["mu"]).abs().max().item()
    assert abs(loss.item() - g["loss"][0]) <= NORTH_STAR_RTOL * abs(g["loss"][0])
    for n, p in vae.named_parameters():
        l2 = g[f"grad_l2/{n}"][0]
        got = p.grad.double().pow(2).sum().sqrt().item()
        assert abs(got - l2) <= 5e-3 * l2 + 1e-12, f"{n}: {got} vs {l2}"


def test_concurrent_schedule_is_bit_identical_to_the_serial_one(monkeypatch):
    """The side-stream schedule (weight gradients, weight re-pack, bias column sums overlapping the main chain) changes no
    arithmetic: from the same state, 25 steps under VP_SIDE_WGRAD=1 and under VP_SIDE_WGRAD=0 must leave bit-identical
    parameters, optimiser moments and BatchNorm buffers.  A missing event wait (a race on the ping-ponged gradient
    buffers or the shared weight-gradient workspace) shows up here as a difference."""
    import vae_play_amd as V
    from vae_play_amd import optim
    from vae_play_amd.engine import FusedVAEStep
    states = []
    # (the weight gradients take the side stream's CU budget on BOTH streams here: their pixel ranges -- the summation order of the
    # slabs -- follow the budget, and the production step gives them the whole chip when they run alone on the main stream)
    monkeypatch.setenv("VP_WGRAD_MAIN_CUS", "160")
    monkeypatch.setenv("VP_WGRAD_SIDE_CUS", "160")
    for mode in ("1", "0"):
        monkeypatch.setenv("VP_SIDE_WGRAD", mode)
        torch.manual_seed(3)
        vae = V.VAE(64, 32, 3).cuda()
        opt = optim.Adam(vae.parameters(), lr=1e-4)
        st = FusedVAEStep(vae, opt, 16, 64, 3)
        assert (st._side_ctx() is not None) == (mode == "1")
        g = torch.Generator().manual_seed(4)
        for i in range(25):
            x = torch.rand(16, 3, 64, 64, generator=g).cuda()
            eps = torch.randn(16, 32, generator=g).cuda()
            st.step(x, eps)
        torch.cuda.synchronize()
        states.append((opt.flat_param.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                       [b.clone() for b in vae.buffers() if b.dtype.is_floating_point]))
    a, b = states
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for u, v in zip(a[3], b[3]):
        assert torch.equal(u, v)


def test_fused_step_binds_or_copies_the_callers_batch():
    """The launches that consume the batch read the caller's tensors in place when they are fp32, contiguous and on the
    device; anything else (a strided view, fp64 data, a host tensor) is copied into the plan's static buffers.  All
    forms must give the same numbers, and a second call with OTHER tensors must not see the first call's data."""
    from oracle import ref_cpu as O
    C, S, z, B = 3, 32, 16, 8
    vae, opt, fused, p0, L = build(C, S, z, B, precision="f32")
    x, eps = O.synthetic_batch(B, C, S, z)
    xd, epsd = x.to(DEV), eps.to(DEV)
    loss0 = fused.forward_backward(xd, epsd)[0].item()      # the returned scalars are the plan's static buffers: read them now
    g0 = opt.flat_grad.clone()
    wide = torch.zeros(B, C, S, 2 * S, device=DEV)
    wide[..., ::2] = xd
    used = torch.zeros_like(g0, dtype=torch.bool)          # alignment padding of the arena is never written
    for p, o in zip(opt.arena.params, opt.arena.offsets):
        used[o:o + p.numel()] = True
    forms = {"strided view": (wide[..., ::2], epsd), "fp64": (xd.double(), epsd.double()), "host tensors": (x, eps),
             "fresh device copies": (xd.clone(), epsd.clone())}
    for tag, (xi, ei) in forms.items():
        opt.flat_grad.fill_(float("nan"))
        li, _, _ = fused.forward_backward(xi, ei)
        assert li.item() == loss0, tag
        assert torch.equal(opt.flat_grad[used], g0[used]), tag
    x2, eps2 = torch.rand_like(xd), torch.randn_like(epsd)
    l2, _, _ = fused.forward_backward(x2, eps2)
    assert l2.item() != loss0
    l3, _, _ = fused.forward_backward(xd, epsd)
    assert l3.item() == loss0


def test_graph_replay_of_the_concurrent_bf16x3_plan_is_bit_identical():
    """FusedVAEStep.capture() on the split-bf16 plan records the side-stream forks / joins (weight gradients and the weight
    re-pack run on a second stream, the split output gradient ping-pongs between two buffers): the replayed graph must write
    exactly the gradients of the eager launch, and capturing must leave the BatchNorm buffers untouched."""
    from oracle import ref_cpu as O
    C, S, z, B = 3, 64, 32, 8
    vae, opt, fused, p0, L = build(C, S, z, B, precision="bf16x3")
    assert fused._n_side_events > 0
    x, eps = O.synthetic_batch(B, C, S, z)
    xd, epsd = x.to(DEV), eps.to(DEV)
    sd0 = {k: v.clone() for k, v in vae.state_dict().items()}
    used = torch.zeros_like(opt.flat_grad, dtype=torch.bool)
    for p, o in zip(opt.arena.params, opt.arena.offsets):
        used[o:o + p.numel()] = True
    opt.flat_grad.fill_(float("nan"))
    loss_e = fused.forward_backward(xd, epsd)[0].item()
    torch.cuda.synchronize()
    g_eager = opt.flat_grad.clone()
    rm_eager = vae.decoder.conv[0].bn.running_mean.clone()
    assert torch.isfinite(g_eager[used]).all()
    vae.load_state_dict(sd0)
    fused.capture()
    for k in ("decoder.conv.0.bn.running_mean", "encoder.conv.1.bn.running_var", "encoder.fc.1.running_mean"):
        assert torch.equal(vae.state_dict()[k], sd0[k]), f"capture() changed {k}"
    for rep in range(2):
        vae.load_state_dict(sd0)
        opt.flat_grad.fill_(float("nan"))
        loss_g = fused.forward_backward(xd, epsd)[0].item()
        torch.cuda.synchronize()
        assert torch.equal(opt.flat_grad[used], g_eager[used]), f"graph replay {rep} differs from the eager launch"
        assert loss_g == loss_e
        assert torch.equal(vae.decoder.conv[0].bn.running_mean, rm_eager)
