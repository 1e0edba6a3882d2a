"""CPU tests of the VAE-GAN row: the oracle restatement (oracle/ref_vaegan.py) against the
vectors the REAL reference produced (tests/golden/vaegan_*.npz, oracle/gen_golden_vaegan.py), and the drop-in
classes' state_dict keys / default-init RNG consumption against the reference's.  No GPU compute here."""
import numpy as np
import torch

from tests.util import load_golden, t


def test_vaegan_state_dict_and_seeded_init_match_reference():
    import vae_play_amd as V
    from oracle import ref_cpu as O
    g = load_golden("vaegan_init")
    torch.manual_seed(7)
    net = V.VaeGan(32, 16)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            assert np.array_equal(g[f"{k}/samples"], O.checksum(v)["samples"].numpy()), k
            assert abs(g[f"{k}/sum"][0] - O.checksum(v)["sum"].item()) <= 1e-9 * max(1.0, abs(g[f"{k}/sum"][0])), k


def test_oracle_spec_names_the_same_tensors_as_the_drop_in():
    import vae_play_amd as V
    from oracle import ref_vaegan as G
    net = V.VaeGan(32, 16)
    spec = [(n, tuple(s)) for n, s, _ in G.vaegan_spec(32, 16)]
    ours = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert spec == ours
    disc = V.Discriminator(channel_in=3, recon_level=1, iter_level=2)
    assert [n for n, _, _ in G.discriminator_spec(3, 2)] == list(disc.state_dict().keys())
    assert disc.recon_levl == 1 and disc.size == 128


def test_vaegan_oracle_matches_golden_step():
    """Re-run the oracle's accumulate-then-step (train.py:68-78) and compare with the reference-generated vectors."""
    from oracle import ref_cpu as O
    from oracle import ref_vaegan as G
    g = load_golden("vaegan_32x32_z16_b4")
    S, z, B = (int(g[k]) for k in ("meta_S", "meta_z", "meta_B"))
    x, targets, eps, z_p = G.synthetic_batch(B, S, z)
    for a, k in ((x, "x"), (targets, "targets"), (eps, "eps"), (z_p, "z_p")):
        assert torch.equal(a, t(g[k])), k
    p = G.init_vaegan_params(S, z, seed=0)
    O.require_grad(p)
    opts = G.make_optimizers(p)
    out, losses = G.train_step(p, opts, x, targets, eps, z_p, S)
    for k, v in out.items():
        if f"out/{k}" in g:
            assert torch.allclose(v, t(g[f"out/{k}"]), rtol=1e-5, atol=1e-6), k
        else:
            assert abs(O.checksum(v)["l2"].item() - g[f"out_l2/{k}"][0]) <= 1e-5 * g[f"out_l2/{k}"][0], k
    for k, v in losses.items():
        assert abs(v.item() - g[f"loss/{k}"][0]) <= 1e-5 * abs(g[f"loss/{k}"][0]) + 1e-7, k
    for n in O.trainable_names(p):
        cs = O.checksum(p[n])
        assert abs(cs["l2"].item() - g[f"param1_l2/{n}"][0]) <= 1e-5 * g[f"param1_l2/{n}"][0] + 1e-9, n
    for n in p:
        if n.endswith("num_batches_tracked"):
            assert int(p[n]) == int(g[f"nbt/{n}"]), n   # discriminator BN layers run twice per step (REC + GAN)
