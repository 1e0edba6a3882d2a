"""Parameter gradients written straight into the optimiser's flat arena (functional._grad_out): with ``.grad`` dropped before the
backward pass (``module.zero_grad()`` of train.py:68 / ``optimizer.zero_grad(set_to_none=True)``) every backward function hands
autograd a view of the parameter's arena slice, which AccumulateGrad adopts without an add kernel.  The gradients must equal the
zero-filled-arena path's bit for bit -- also for parameters used twice in one graph (the VAE-GAN's decoder and discriminator)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _vaegan_grads(set_to_none: bool, module_zero: bool = False):
    import vae_play_amd as V
    from vae_play_amd import optim
    torch.manual_seed(0)
    net = V.VaeGan(32, 16, num_of_param=3).to(DEV)
    net.train()
    opts = [optim.RMSprop(m.parameters(), lr=1e-4) for m in (net.encoder, net.decoder, net.discriminator, net.param_encoder)]
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(4, 1, 32, 32, generator=g).to(DEV)
    targets = torch.rand(4, 3, generator=g).to(DEV)
    noise = torch.randn(4, 16, generator=g).to(DEV)
    outs = []
    for _ in range(2):                      # two steps: the markers must re-arm
        if module_zero:
            net.zero_grad()                 # torch's: sets every .grad to None
        else:
            for o in opts:
                o.zero_grad(set_to_none=set_to_none)
        x_tilde, dc, dl, mus, lv, params = net(imgs, eps=noise, z_p=noise)
        B = imgs.size(0)
        nle, kl, mse, bo, bp, bs, l1 = V.VaeGan.loss(imgs, x_tilde, dl[:B], dl[B:-B], dl[-B:], dc[:B], dc[B:-B], dc[-B:], mus, lv, targets, params)
        V.VaeGan.backward_all(torch.nn.functional.mse_loss(imgs, x_tilde), kl.sum() + mse.sum(), 1e-6 * mse.sum() - (bo.sum() + bp.sum() + bs.sum()),
                              bo.sum() + bp.sum() + bs.sum(), l1)
        for o in opts:
            o.arena.gather_grads()
        outs.append({n: p.grad.detach().clone() for n, p in net.named_parameters()})
        for o in opts:
            o.step()
    return outs


def test_vaegan_gradients_are_identical_with_and_without_direct_writes():
    import os
    from vae_play_amd import functional as Fh
    ref = _vaegan_grads(set_to_none=False)
    for kw in (dict(set_to_none=True), dict(set_to_none=False, module_zero=True)):
        got = _vaegan_grads(**kw)
        for step, (a, b) in enumerate(zip(ref, got)):
            for n in a:
                # a parameter used twice receives view + fresh tensor: autograd sums them in another order than onto zeros
                assert torch.allclose(a[n], b[n], rtol=2e-6, atol=1e-9), (kw, step, n, (a[n] - b[n]).abs().max().item())
    # the switch
    old = Fh._DIRECT_GRADS
    try:
        Fh._DIRECT_GRADS = False
        got = _vaegan_grads(set_to_none=True)
        for a, b in zip(ref, got):
            for n in a:
                assert torch.allclose(a[n], b[n], rtol=2e-6, atol=1e-9), n
    finally:
        Fh._DIRECT_GRADS = old


def test_gradient_lands_in_the_arena_without_a_copy():
    import vae_play_amd as V
    from vae_play_amd import optim
    torch.manual_seed(0)
    vae = V.VAE(32, 16, 3).to(DEV)
    opt = optim.Adam(vae.parameters(), lr=1e-4)
    opt.zero_grad(set_to_none=True)
    opt.flat_grad.fill_(float("nan"))
    x = torch.rand(4, 3, 32, 32, device=DEV)
    xt, mu, lv = vae(x, eps=torch.randn(4, 16, device=DEV))
    loss, _, _ = V.vae_loss(x, xt, mu, lv)
    loss.backward()
    base = opt.flat_grad.data_ptr()
    for p, o in zip(opt.arena.params, opt.arena.offsets):
        assert p.grad is not None and p.grad.data_ptr() == base + 4 * o, "autograd adopted something other than the arena slice"
        assert torch.isfinite(p.grad).all()
        assert not p._vp_pending


def test_packed_weight_cache_follows_optimiser_and_torch_updates():
    """functional._packed: a conv weight is re-packed when (and only when) its value changed -- by a flat-arena optimiser step
    (kernels behind torch's version counter) or by a torch in-place op."""
    import vae_play_amd as V
    from vae_play_amd import functional as Fh, optim
    torch.manual_seed(0)
    vae = V.VAE(32, 16, 3).to(DEV)
    opt = optim.Adam(vae.parameters(), lr=1e-2)
    Fh.set_conv_precision("bf16x3")
    try:
        x = torch.rand(4, 3, 32, 32, device=DEV)
        eps = torch.randn(4, 16, device=DEV)

        def run():
            xt, mu, lv = vae(x, eps=eps)
            return xt.detach().clone()
        a = run()
        b = run()                              # served from the cache
        assert torch.equal(a, b)
        opt.zero_grad()
        xt, mu, lv = vae(x, eps=eps)
        V.vae_loss(x, xt, mu, lv)[0].backward()
        opt.step()                             # weights change behind torch's back
        c = run()
        assert not torch.equal(a, c), "stale packed weights after an optimiser step"
        old = Fh._PACK_CACHE_ON
        Fh._PACK_CACHE_ON = False
        try:
            assert torch.equal(c, run()), "cached and freshly packed weights disagree"
        finally:
            Fh._PACK_CACHE_ON = old
        with torch.no_grad():
            vae.decoder.conv[0].conv.weight.mul_(0.5)      # torch in-place update: the version counter moves
        d = run()
        Fh._PACK_CACHE_ON = False
        try:
            assert torch.equal(d, run())
        finally:
            Fh._PACK_CACHE_ON = old
    finally:
        Fh.set_conv_precision("f32")
