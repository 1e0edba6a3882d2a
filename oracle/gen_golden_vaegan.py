"""Generate tests/golden/vaegan_*.npz from the REAL reference and pin oracle/ref_vaegan.py to it.

Same rules as oracle/gen_golden.py: runs only where the reference checkout is mounted (default
/root/reference), imports its modules, executes them on seeded synthetic inputs on the CPU and writes inputs /
outputs / checksums only.  ``VaeGan.forward`` itself cannot run without a GPU (it calls ``.cuda()``,
models/networks.py:241) and draws its noise internally, so the training branch is composed here from the
reference's own sub-modules in the order of models/networks.py:234-247 with eps / z_p injected, the losses come
from the reference's ``VaeGan.loss`` and the five scalar losses / backward order are those of train.py:61-72.

    python oracle/gen_golden_vaegan.py
"""
from __future__ import annotations

import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_cpu as O  # noqa: E402
from oracle import ref_vaegan as G  # noqa: E402
from oracle.gen_golden import bit_equal, import_reference, np_  # noqa: E402


def ref_train_step(nets, ref, opts, x, targets, eps, z_p):
    B = x.size(0)
    mus, logvar = ref.encoder(x)
    z = eps * torch.exp(0.5 * logvar) + mus                      # reparameterize with the injected eps
    x_tilde = ref.decoder(z)
    params = ref.param_encoder(z)
    zp = z_p.clone().requires_grad_(True)
    x_p = ref.decoder(zp)
    disc_layer = ref.discriminator(x, x_tilde, x_p, "REC")
    disc_class = ref.discriminator(x, x_tilde, x_p, "GAN")
    dl = (disc_layer[:B], disc_layer[B:-B], disc_layer[-B:])
    dc = (disc_class[:B], disc_class[B:-B], disc_class[-B:])
    nle, kl, mse, bo, bp, bs, l1 = nets.VaeGan.loss(x, x_tilde, *dl, *dc, mus, logvar, targets, params)
    lam = G.LAMBDA_MSE
    loss_recon = F.mse_loss(x, x_tilde)
    loss_encoder = torch.sum(kl) + torch.sum(mse)
    loss_discriminator = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    loss_decoder = torch.sum(lam * mse) - (1.0 - lam) * loss_discriminator
    loss_aux = l1
    ref.zero_grad()
    loss_recon.backward(retain_graph=True)
    loss_encoder.backward(retain_graph=True)
    loss_decoder.backward(retain_graph=True)
    loss_discriminator.backward(retain_graph=True)
    loss_aux.backward()
    for o in opts:
        o.step()
    out = {"x_tilde": x_tilde, "disc_class": disc_class, "disc_layer": disc_layer, "mus": mus, "logvar": logvar,
           "params": params, "nle": nle, "kl": kl, "mse": mse, "bce_dis_original": bo, "bce_dis_predicted": bp,
           "bce_dis_sampled": bs, "l1_enc_param": l1}
    losses = {"loss_recon": loss_recon, "loss_encoder": loss_encoder, "loss_decoder": loss_decoder,
              "loss_discriminator": loss_discriminator, "loss_aux": loss_aux}
    return out, losses


def vaegan_fixture(nets, name, S, z, B, n_steps):
    p0 = G.init_vaegan_params(S, z, seed=0)
    x, targets, eps, z_p = G.synthetic_batch(B, S, z)

    ref = nets.VaeGan(S, z)
    missing = ref.load_state_dict({k: v.clone() for k, v in p0.items()})
    assert not missing.missing_keys and not missing.unexpected_keys, missing
    ref.train()
    ropts = [torch.optim.RMSprop(m.parameters(), lr=1e-4)
             for m in (ref.encoder, ref.decoder, ref.discriminator, ref.param_encoder)]

    p = O.clone_params(p0)
    O.require_grad(p)
    oopts = G.make_optimizers(p)

    fx = {"meta_S": S, "meta_z": z, "meta_B": B, "meta_steps": n_steps,
          "x": np_(x), "targets": np_(targets), "eps": np_(eps), "z_p": np_(z_p)}
    for step in range(1, n_steps + 1):
        r_out, r_loss = ref_train_step(nets, ref, ropts, x, targets, eps, z_p)
        o_out, o_loss = G.train_step(p, oopts, x, targets, eps, z_p, S)
        for k in o_out:
            bit_equal(o_out[k], r_out[k].detach(), f"{name} step{step} {k}")
        for k in o_loss:
            bit_equal(o_loss[k], r_loss[k].detach(), f"{name} step{step} {k}")
        rsd = ref.state_dict(keep_vars=True)
        for n in p:
            bit_equal(p[n].detach(), rsd[n].detach(), f"{name} step{step} param {n}")
        if step == 1:
            for n in O.trainable_names(p):
                bit_equal(p[n].grad, rsd[n].grad, f"{name} grad {n}")
            for k, v in o_out.items():
                if v.numel() <= 8192:
                    fx[f"out/{k}"] = np_(v)
                else:
                    cs = O.checksum(v)
                    fx[f"out_sum/{k}"], fx[f"out_l2/{k}"] = np_(cs["sum"]), np_(cs["l2"])
                    fx[f"out_stride7/{k}"] = np_(v.flatten()[::7][:8192])
            for k, v in o_loss.items():
                fx[f"loss/{k}"] = np_(v.double().reshape(1))
            for n in O.trainable_names(p):
                cs = O.checksum(p[n].grad)
                fx[f"grad_sum/{n}"] = np_(cs["sum"]); fx[f"grad_l2/{n}"] = np_(cs["l2"])
                fx[f"grad_samples/{n}"] = np_(cs["samples"])
            for n in p:
                if n.endswith(("running_mean", "running_var")):
                    t = p[n]
                    fx[f"bn/{n}"] = np_(t if t.numel() <= 4096 else t[:4096])
                if n.endswith("num_batches_tracked"):
                    fx[f"nbt/{n}"] = np_(p[n])
        for n in O.trainable_names(p):
            cs = O.checksum(p[n])
            fx[f"param{step}_sum/{n}"] = np_(cs["sum"]); fx[f"param{step}_l2/{n}"] = np_(cs["l2"])
            fx[f"param{step}_samples/{n}"] = np_(cs["samples"])
        for k, v in o_loss.items():
            fx[f"loss_step{step}/{k}"] = np_(v.double().reshape(1))
    return fx


def vaegan_init_fixture(nets):
    """torch.manual_seed(7); VaeGan(32, 16): the drop-in class must consume the global RNG identically."""
    torch.manual_seed(7)
    ref = nets.VaeGan(32, 16)
    out = {}
    for k, v in ref.state_dict().items():
        if v.dtype.is_floating_point:
            cs = O.checksum(v)
            out[f"{k}/sum"] = np_(cs["sum"]); out[f"{k}/samples"] = np_(cs["samples"])
    out["keys"] = np.array(list(ref.state_dict().keys()))
    return out


def disc_fixture(nets):
    """Discriminator alone (3 input channels, recon_level below the top) forward/backward in both modes."""
    g = torch.Generator().manual_seed(31)
    ref = nets.Discriminator(channel_in=3, recon_level=1, iter_level=2)
    ref.load_state_dict(G.seeded_disc_weights(ref.state_dict()))
    ref.train()
    B = 2
    xs = [torch.rand(B, 3, 32, 32, generator=g).requires_grad_(True) for _ in range(3)]
    p = {"discriminator." + k: v.detach().clone() for k, v in ref.state_dict().items()}
    O.require_grad(p)
    xo = [t.detach().clone().requires_grad_(True) for t in xs]
    fx = {"x0": np_(xs[0]), "x1": np_(xs[1]), "x2": np_(xs[2]), "weight_seed": np.array(77)}
    for mode in ("REC", "GAN"):
        y = ref(xs[0], xs[1], xs[2], mode)
        gy = torch.randn(y.shape, generator=g)
        ref.zero_grad()
        for t in xs:
            t.grad = None
        y.backward(gy)
        yo = G.discriminator_forward(p, xo[0], xo[1], xo[2], mode, 1, 2, True)
        for n in O.trainable_names(p):
            p[n].grad = None
        for t in xo:
            t.grad = None
        yo.backward(gy)
        bit_equal(yo.detach(), y.detach(), f"disc {mode} y")
        fx[f"{mode}/y"] = np_(y); fx[f"{mode}/gy"] = np_(gy)
        for i in range(3):
            bit_equal(xo[i].grad, xs[i].grad, f"disc {mode} dx{i}")
            fx[f"{mode}/dx{i}"] = np_(xs[i].grad)
        rsd = ref.state_dict(keep_vars=True)
        for n in O.trainable_names(p):
            rg = rsd[n[len("discriminator."):]].grad
            og = p[n].grad
            if rg is None:
                assert og is None or float(og.abs().max()) == 0.0, n
                continue
            bit_equal(og, rg, f"disc {mode} grad {n}")
            if rg.numel() <= 65536:
                fx[f"{mode}/grad/{n}"] = np_(rg)
            else:
                cs = O.checksum(rg)
                fx[f"{mode}/grad_sum/{n}"], fx[f"{mode}/grad_l2/{n}"] = np_(cs["sum"]), np_(cs["l2"])
                fx[f"{mode}/grad_samples/{n}"] = np_(cs["samples"])
    for k, v in ref.state_dict().items():
        if "running" in k:
            bit_equal(p["discriminator." + k], v, f"disc bn {k}")
            fx["bn/" + k] = np_(v)
    return fx


def main():
    nets, _ = import_reference()
    outdir = os.path.join(ROOT, "tests", "golden")

    def save(name, fx):
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")

    save("vaegan_init", vaegan_init_fixture(nets))
    save("vaegan_disc_c3_l2", disc_fixture(nets))
    save("vaegan_32x32_z16_b4", vaegan_fixture(nets, "vg32", 32, 16, 4, 2))
    save("vaegan_64x64_z32_b4", vaegan_fixture(nets, "vg64", 64, 32, 4, 1))
    print("oracle == reference (bit-exact) on every VAE-GAN fixture")


if __name__ == "__main__":
    main()
