"""Structure of the fused VAE-GAN launch plan, checked without a GPU (the plan is built over host buffers and never run): every
parameter's gradient slice -- and, for the decoder, its slice of the shadow arena of the second pass -- is an output argument of
some backward launch, and every launch resolves to a symbol of the C ABI."""
import pytest


@pytest.mark.parametrize("S,z,B", [(32, 16, 4), (128, 128, 16)])
def test_plan_covers_every_gradient(S, z, B):
    import vae_play_amd as V
    from vae_play_amd import optim
    from vae_play_amd.engine_gan import FusedVAEGANStep
    net = V.VaeGan(S, z).train()
    opts = [optim.RMSprop(m.parameters(), lr=1e-4) for m in (net.encoder, net.decoder, net.discriminator, net.param_encoder)]
    st = FusedVAEGANStep(net, opts, B, S, _plan_only=True)
    ptrs = set()
    for c in st._bwd.calls:
        for a in c[2]:
            if hasattr(a, "value") and a.value:
                ptrs.add(a.value)
    dec_ids = {id(p) for p in net.decoder.parameters()}
    for n, p in net.named_parameters():
        assert p._vp_arena.grad_view(p).data_ptr() in ptrs, n
        if id(p) in dec_ids:
            assert st._dec_shadow.data_ptr() + 4 * p._vp_off in ptrs, n + " (second decoder pass)"
    # BatchNorm forward passes per step: decoder layers run twice (z, z_p), discriminator blocks count both reference calls
    counts = {}
    for bn, c in st._bn_counts:
        counts[id(bn)] = counts.get(id(bn), 0) + c
    for m in net.decoder.modules():
        if hasattr(m, "num_batches_tracked"):
            assert counts[id(m)] == 2
    for m in net.encoder.modules():
        if hasattr(m, "num_batches_tracked"):
            assert counts[id(m)] == 1
    assert counts[id(net.discriminator.fc[1])] == 1
    for blk in list(net.discriminator.conv)[1:]:
        assert counts[id(blk.bn)] == 2
    # the two loss coefficients as fp32 accumulation forms them
    assert abs(st.c_disc - 1e-6) < 2e-8 and st.c_disc != 1e-6
    assert abs(st.c_mse - 1.000001) < 1e-7
