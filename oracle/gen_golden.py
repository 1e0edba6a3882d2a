"""Generate tests/golden/*.npz from the REAL reference and pin the oracle to it.

Runs only in the authoring container, where the reference checkout is mounted
read-only (default /root/reference, override with VAEPLAY_REFERENCE).  Nothing
from the reference is copied: its modules are imported, executed on seeded
synthetic inputs, and only inputs / outputs / checksums are written.  Before a
fixture is written, oracle/ref_cpu.py is asserted to reproduce the reference
bit-for-bit on the same inputs -- that is what "parity pinned" means here.

    python oracle/gen_golden.py            # writes tests/golden/*.npz

The GPU box never runs this script (no reference there); it consumes the .npz.
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
REF = os.environ.get("VAEPLAY_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import ref_cpu as O  # noqa: E402

torch.set_num_threads(max(1, (os.cpu_count() or 2)))
torch.set_flush_denormal(False)


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit(f"reference checkout not found at {REF}")
    # this repository has a top-level ``models`` package of its own (import-path aliases of the drop-ins), and the reference's
    # ``models/`` has no __init__.py (a namespace package, which a regular package of the same name shadows whatever the
    # sys.path order): load the two reference files BY PATH under private names.  Both import only torch / numpy / math.
    import importlib.util

    def load(private_name, rel):
        path = os.path.join(REF, rel)
        spec = importlib.util.spec_from_file_location(private_name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if not os.path.realpath(mod.__file__).startswith(os.path.realpath(REF) + os.sep):
            raise SystemExit(f"{private_name} resolved to {mod.__file__}, not to the reference checkout {REF}")
        return mod

    nets = load("_ref_models_networks", os.path.join("models", "networks.py"))
    blocks = load("_ref_models_blocks", os.path.join("models", "blocks.py"))
    return nets, blocks


def bit_equal(a: torch.Tensor, b: torch.Tensor, what: str):
    if a.shape != b.shape or not torch.equal(a, b):
        d = (a.double() - b.double()).abs().max().item() if a.shape == b.shape else float("nan")
        raise AssertionError(f"oracle != reference for {what}: max|d|={d}")


def np_(t):
    # copy: parameters / BN buffers are updated in place by later steps
    return t.detach().cpu().numpy().copy()


# --------------------------------------------------------------------------
# whole-step fixtures
# --------------------------------------------------------------------------
def ref_modules(nets, p0, C, z, L):
    enc = nets.Encoder(channel_in=C, z_size=z, iter_level=L)
    dec = nets.Decoder(z_size=z, size=enc.size, channel_out=C, iter_level=L)
    enc.load_state_dict({k[len("encoder."):]: v.clone() for k, v in p0.items() if k.startswith("encoder.")})
    dec.load_state_dict({k[len("decoder."):]: v.clone() for k, v in p0.items() if k.startswith("decoder.")})
    enc.train(); dec.train()
    return enc, dec


def ref_named(enc, dec):
    sd = {"encoder." + k: v for k, v in enc.state_dict(keep_vars=True).items()}
    sd.update({"decoder." + k: v for k, v in dec.state_dict(keep_vars=True).items()})
    return sd


def ref_step(nets, enc, dec, opt, x, eps):
    """SURVEY.md 3.3 composed step run on the reference's own modules."""
    opt.zero_grad()
    mu, logvar = enc(x)
    z = eps * torch.exp(0.5 * logvar) + mu
    x_tilde = dec(z)
    recon = F.binary_cross_entropy(x_tilde, x, reduction="sum")
    d = torch.zeros(len(x), 1)
    kl = nets.VaeGan.loss(x, x_tilde, d, d, d, d + 0.5, d + 0.5, d + 0.5, mu, logvar, d, d)[1]
    loss = (recon + kl.sum()) / len(x)
    loss.backward()
    opt.step()
    return {"mu": mu, "logvar": logvar, "z": z, "x_tilde": x_tilde, "loss": loss, "recon": recon, "kl": kl.sum()}


def step_fixture(nets, name, C, S, z, B, optim_kind, n_steps, full_xt):
    L = O.iter_level_for(S)
    p0 = O.init_params(C, z, L, seed=0)
    x, eps = O.synthetic_batch(B, C, S, z)

    # reference run
    enc, dec = ref_modules(nets, p0, C, z, L)
    ref_params = list(enc.parameters()) + list(dec.parameters())
    ropt = (torch.optim.Adam if optim_kind == "adam" else torch.optim.RMSprop)(ref_params, lr=1e-4)

    # oracle run
    p = O.clone_params(p0)
    O.require_grad(p)
    oopt = O.make_optimizer(p, optim_kind, lr=1e-4)

    fx = {"meta_C": C, "meta_S": S, "meta_z": z, "meta_B": B, "meta_L": L,
          "meta_optim": optim_kind, "meta_steps": n_steps}
    if full_xt:
        fx["x"] = np_(x); fx["eps"] = np_(eps)
    fx["x_sum"] = np_(O.checksum(x)["sum"]); fx["eps_sum"] = np_(O.checksum(eps)["sum"])

    for step in range(1, n_steps + 1):
        r = ref_step(nets, enc, dec, ropt, x, eps)
        o = O.train_step(p, oopt, x, eps, L)
        for k in ("mu", "logvar", "z", "x_tilde", "loss", "recon", "kl"):
            bit_equal(o[k], r[k].detach(), f"{name} step{step} {k}")
        rn = ref_named(enc, dec)
        for n in p:
            bit_equal(p[n].detach(), rn[n].detach(), f"{name} step{step} param {n}")
        if step == 1:
            for n in O.trainable_names(p):
                bit_equal(p[n].grad, rn[n].grad, f"{name} grad {n}")
            for k in ("mu", "logvar", "z"):
                fx[k] = np_(o[k])
            xt = o["x_tilde"]
            if full_xt:
                fx["x_tilde"] = np_(xt)
            fx["x_tilde_stride7"] = np_(xt.flatten()[::7][:8192])
            cs = O.checksum(xt)
            fx["x_tilde_sum"], fx["x_tilde_l2"] = np_(cs["sum"]), np_(cs["l2"])
            for k in ("loss", "recon", "kl"):
                fx[k] = np_(o[k].double().reshape(1))
            for n in O.trainable_names(p):
                cs = O.checksum(p[n].grad)
                fx[f"grad_sum/{n}"] = np_(cs["sum"]); fx[f"grad_l2/{n}"] = np_(cs["l2"])
                fx[f"grad_samples/{n}"] = np_(cs["samples"])
            for n in p:
                if n.endswith(("running_mean", "running_var")):
                    t = p[n]
                    fx[f"bn/{n}"] = np_(t if t.numel() <= 4096 else t[:4096])
        for n in O.trainable_names(p):
            cs = O.checksum(p[n])
            fx[f"param{step}_sum/{n}"] = np_(cs["sum"]); fx[f"param{step}_l2/{n}"] = np_(cs["l2"])
            fx[f"param{step}_samples/{n}"] = np_(cs["samples"])
        fx[f"loss_step{step}"] = np_(o["loss"].double().reshape(1))
    return fx


# --------------------------------------------------------------------------
# single-op fixtures (full tensors, tiny shapes)
# --------------------------------------------------------------------------
def block_fixture(nets, kind, cin, cout, B, H, seed):
    g = torch.Generator().manual_seed(seed)
    blk = (nets.EncoderBlock if kind == "enc" else nets.DecoderBlock)(cin, cout)
    with torch.no_grad():
        blk.conv.weight.copy_(torch.randn(blk.conv.weight.shape, generator=g) * 0.1)
        blk.bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        blk.bn.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    blk.train()
    x = torch.randn(B, cin, H, H, generator=g, requires_grad=True)
    y = blk(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    # oracle on the same numbers
    p = {"b.conv.weight": blk.conv.weight.detach().clone().requires_grad_(True),
         "b.bn.weight": blk.bn.weight.detach().clone().requires_grad_(True),
         "b.bn.bias": blk.bn.bias.detach().clone().requires_grad_(True),
         "b.bn.running_mean": torch.zeros(cout), "b.bn.running_var": torch.ones(cout),
         "b.bn.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    xo = x.detach().clone().requires_grad_(True)
    yo = (O.encoder_block if kind == "enc" else O.decoder_block)(p, "b", xo, True)
    yo.backward(gy)
    tag = f"{kind}block {cin}->{cout}"
    bit_equal(yo.detach(), y.detach(), tag + " y")
    bit_equal(xo.grad, x.grad, tag + " dx")
    bit_equal(p["b.conv.weight"].grad, blk.conv.weight.grad, tag + " dw")
    bit_equal(p["b.bn.weight"].grad, blk.bn.weight.grad, tag + " dgamma")
    bit_equal(p["b.bn.bias"].grad, blk.bn.bias.grad, tag + " dbeta")
    bit_equal(p["b.bn.running_mean"], blk.bn.running_mean, tag + " rm")
    bit_equal(p["b.bn.running_var"], blk.bn.running_var, tag + " rv")
    return {"x": np_(x), "w": np_(blk.conv.weight), "gamma": np_(blk.bn.weight), "beta": np_(blk.bn.bias),
            "y": np_(y), "gy": np_(gy), "dx": np_(x.grad), "dw": np_(blk.conv.weight.grad),
            "dgamma": np_(blk.bn.weight.grad), "dbeta": np_(blk.bn.bias.grad),
            "running_mean": np_(blk.bn.running_mean), "running_var": np_(blk.bn.running_var)}


def latent_fixture(nets):
    """reparameterize (models/networks.py:228-231) with its own normal_() draw and KL (:270)."""
    g = torch.Generator().manual_seed(99)
    mu = torch.randn(8, 32, generator=g)
    lv = torch.randn(8, 32, generator=g) * 0.5
    lv_before = lv.clone()
    torch.manual_seed(4321)
    z_ref = nets.VaeGan.reparameterize(None, mu, lv)
    bit_equal(lv, lv_before, "reparameterize must not mutate logvar")
    torch.manual_seed(4321)
    eps = torch.randn(8, 32)
    bit_equal(O.reparameterize(mu, lv, eps), z_ref, "reparameterize with same-seed eps")
    d = torch.zeros(8, 1)
    kl_ref = nets.VaeGan.loss(d, d, d, d, d, d + 0.5, d + 0.5, d + 0.5, mu, lv, d, d)[1]
    bit_equal(O.kl_per_sample(mu, lv), kl_ref, "kl")
    return {"mu": np_(mu), "logvar": np_(lv), "eps": np_(eps), "z": np_(z_ref), "kl": np_(kl_ref)}


def init_fixture(nets):
    """Default-constructor init and the VaeGan.init_parameters rule under a global seed
    (SURVEY.md 8a-10): samples + checksums so the drop-in classes can be checked on the box."""
    out = {}
    torch.manual_seed(0)
    enc = nets.Encoder(3, 16, 2)
    dec = nets.Decoder(16, enc.size, 3, 2)
    for tag, mod in (("enc", enc), ("dec", dec)):
        for k, v in mod.state_dict().items():
            if v.dtype.is_floating_point:
                cs = O.checksum(v)
                out[f"default/{tag}.{k}/sum"] = np_(cs["sum"]); out[f"default/{tag}.{k}/samples"] = np_(cs["samples"])
    # the rule itself, applied the reference's way (borrow the unbound method on a holder module)
    holder = torch.nn.Module()
    holder.encoder, holder.decoder = enc, dec
    torch.manual_seed(5)
    nets.VaeGan.init_parameters(holder)
    for tag, mod in (("enc", enc), ("dec", dec)):
        for k, v in mod.state_dict().items():
            if v.dtype.is_floating_point:
                cs = O.checksum(v)
                out[f"rule/{tag}.{k}/sum"] = np_(cs["sum"]); out[f"rule/{tag}.{k}/samples"] = np_(cs["samples"])
    return out


def _params_of(mod):
    return {k: v.detach().clone() for k, v in mod.state_dict().items()}


def _run_block(mod, oracle_fn, x, seed, tag):
    """Forward + backward of a reference blocks module and of the oracle restatement on the same numbers;
    returns a fixture with full tensors."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for q in mod.parameters():          # non-trivial affine parameters
            q.copy_(q + 0.1 * torch.randn(q.shape, generator=g))
    p0 = _params_of(mod)
    mod.train()
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    p = {k: v.clone() for k, v in p0.items()}
    for k in p:
        if p[k].dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            p[k].requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    yo = oracle_fn(p, xo)
    yo.backward(gy)
    bit_equal(yo.detach(), y.detach(), tag + " y")
    bit_equal(xo.grad, xr.grad, tag + " dx")
    fx = {"x": np_(x), "y": np_(y), "gy": np_(gy), "dx": np_(xr.grad)}
    named = dict(mod.named_parameters())
    for k, v in p0.items():
        fx["param/" + k] = np_(v)
    for k, q in named.items():
        bit_equal(p[k].grad, q.grad, tag + " grad " + k)
        fx["grad/" + k] = np_(q.grad)
    for k, v in mod.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            bit_equal(p[k], v, tag + " " + k)
            fx["after/" + k] = np_(v)
    return fx


def blocks_fixtures(blocks, save):
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 6, 10, 10, generator=g)
    i = 0
    for bn in (None, "batch", "instance"):
        for act in ("relu", "lrelu", "tanh", None):
            for ks, stride in ((3, 1), (3, 2)) if i % 2 == 0 else ((1, 1), (5, 2)):
                torch.manual_seed(100 + i)
                mod = blocks.Conv2d(6, 8, ks, stride, bn, act)
                fx = _run_block(mod, lambda p, xx: O.blocks_conv2d(p, "", xx, ks, stride, bn, act, True), x, 200 + i,
                                f"blocks.Conv2d k{ks} s{stride} {bn} {act}")
                fx["meta"] = np.array([6, 8, ks, stride])
                save(f"blocks_conv2d_k{ks}s{stride}_{bn}_{act}", fx)
            i += 1
    torch.manual_seed(300)
    mod = blocks.Up(6, 4, True)
    save("blocks_up_coord", _run_block(mod, lambda p, xx: O.blocks_up(p, "", xx, True, True), x, 301, "blocks.Up"))
    torch.manual_seed(310)
    mod = blocks.Up(6, 8, False)
    save("blocks_up", _run_block(mod, lambda p, xx: O.blocks_up(p, "", xx, False, True), torch.randn(2, 6, 7, 9, generator=g), 311, "blocks.Up odd"))
    torch.manual_seed(320)
    mod = blocks.Down(6, 8, 3, True)
    save("blocks_down_coord", _run_block(mod, lambda p, xx: O.blocks_down(p, "", xx, 3, True, True), x, 321, "blocks.Down"))
    for act in ("relu", "lrelu", "tanh", None):
        torch.manual_seed(330)
        mod = blocks.Linear(12, 7, True, act)
        xl = torch.randn(5, 12, generator=g)
        save(f"blocks_linear_{act}", _run_block(mod, lambda p, xx: O.blocks_linear(p, "", xx, act), xl, 331, f"blocks.Linear {act}"))
    for norm in (False, True):
        ac = blocks.AddCoords(norm)
        xa = torch.randn(2, 3, 5, 7, generator=g)
        ya = ac(xa)
        bit_equal(O.blocks_add_coords(xa, norm), ya, "AddCoords")
        save(f"blocks_addcoords_{int(norm)}", {"x": np_(xa), "y": np_(ya)})


def main():
    nets, blocks = import_reference()
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)

    def save(name, fx):
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")

    blocks_fixtures(blocks, save)
    save("latent", latent_fixture(nets))
    save("init", init_fixture(nets))
    save("encblock_4to8", block_fixture(nets, "enc", 4, 8, 2, 16, 11))
    save("encblock_3to8", block_fixture(nets, "enc", 3, 8, 2, 16, 12))
    save("encblock_1to64", block_fixture(nets, "enc", 1, 64, 2, 16, 13))
    save("encblock_64to128", block_fixture(nets, "enc", 64, 128, 2, 8, 14))
    save("decblock_8to4", block_fixture(nets, "dec", 8, 4, 2, 8, 21))
    save("decblock_64to32", block_fixture(nets, "dec", 64, 32, 2, 8, 22))
    save("decblock_128to128", block_fixture(nets, "dec", 128, 128, 3, 8, 23))
    # whole composed step; 32x32x1 is BASELINE config 1 run at the nearest valid size
    save("step_32x32x1_z16_b4_adam", step_fixture(nets, "s32", 1, 32, 16, 4, "adam", 3, True))
    save("step_32x32x1_z16_b4_rmsprop", step_fixture(nets, "s32r", 1, 32, 16, 4, "rmsprop", 1, False))
    save("step_64x64x3_z64_b4_adam", step_fixture(nets, "s64", 3, 64, 64, 4, "adam", 1, False))
    save("step_128x128x3_z128_b4_adam", step_fixture(nets, "s128", 3, 128, 128, 4, "adam", 1, False))
    # BASELINE config 3 at its per-GPU shard size (32 images): the shape bench.py times
    save("step_128x128x3_z128_b32_adam", step_fixture(nets, "s128b32", 3, 128, 128, 32, "adam", 1, False))
    print("oracle == reference (bit-exact) on every fixture")


if __name__ == "__main__":
    main()
