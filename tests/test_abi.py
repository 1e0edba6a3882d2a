"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/vaeplay_hip.h declares, the ctypes table mirrors the header, and the product path
refuses to run without a GPU instead of falling back to anything."""
import os
import re

import pytest
import torch

from tests.util import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "vaeplay_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vp_[a-z0-9_]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    from vae_play_amd import _lib
    lib = _lib.load()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libvaeplay_hip.so does not export {n}"
    assert lib.vp_abi_version() >= 1


def test_ctypes_table_matches_header():
    from vae_play_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_header_argument_counts_match_ctypes():
    from vae_play_amd import _lib
    src = open(os.path.join(ROOT, "include", "vaeplay_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        body = m.group(1).strip()
        n = 0 if body in ("", "void") else body.count(",") + 1
        assert n == len(args), f"{name}: header has {n} parameters, ctypes table {len(args)}"


def test_no_cpu_fallback_in_product_path():
    """Ops must raise on CPU tensors (the oracle is never reachable from the package)."""
    from vae_play_amd import _lib, ops
    with pytest.raises(_lib.VaePlayHipError):
        ops._p(torch.zeros(4))
    pkg = os.path.join(ROOT, "vae_play_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            text = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in text.replace("the oracle is never", ""), f"{fn} mentions the oracle"


def test_drop_in_state_dict_keys_match_reference_fixture():
    """Key names / shapes / seeded init equal the reference's (fixture written by oracle/gen_golden.py)."""
    import numpy as np
    import vae_play_amd as V
    from oracle import ref_cpu as O
    from tests.util import load_golden
    g = load_golden("init")
    torch.manual_seed(0)
    enc = V.Encoder(3, 16, 2)
    dec = V.Decoder(16, enc.size, 3, 2)
    ref_keys = {k.split("/")[1] for k in g if k.startswith("default/")}
    ours = {f"{tag}.{k}" for tag, m in (("enc", enc), ("dec", dec)) for k, v in m.state_dict().items()
            if v.dtype.is_floating_point}
    assert ours == ref_keys
    for tag, m in (("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                assert np.array_equal(g[f"default/{tag}.{k}/samples"], O.checksum(v)["samples"].numpy()), k
    torch.manual_seed(5)
    V.init_parameters(enc, dec)
    for tag, m in (("enc", enc), ("dec", dec)):
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                assert np.array_equal(g[f"rule/{tag}.{k}/samples"], O.checksum(v)["samples"].numpy()), k
    # spec used by the oracle names exactly the same tensors
    spec = {n for n, _, _ in O.vae_spec(3, 16, 2)}
    vae = V.VAE(32, 16, 3)
    assert set(vae.state_dict().keys()) == spec


def test_oracle_matches_golden_step():
    """The oracle travels to the GPU box; re-check it there (and here) against the committed
    reference-generated vectors."""
    from oracle import ref_cpu as O
    from tests.util import load_golden, t
    g = load_golden("step_32x32x1_z16_b4_adam")
    C, S, z, B, L = (int(g[k]) for k in ("meta_C", "meta_S", "meta_z", "meta_B", "meta_L"))
    p = O.init_params(C, z, L, seed=0)
    O.require_grad(p)
    opt = O.make_optimizer(p, "adam", 1e-4)
    x, eps = O.synthetic_batch(B, C, S, z)
    assert torch.equal(x, t(g["x"])) and torch.equal(eps, t(g["eps"]))
    out = O.train_step(p, opt, x, eps, L)
    for k in ("mu", "logvar", "z", "x_tilde"):
        assert torch.allclose(out[k], t(g[k]), rtol=1e-5, atol=1e-6), k
    assert abs(out["loss"].item() - g["loss"][0]) <= 1e-5 * abs(g["loss"][0])


def test_reference_module_paths_resolve_to_the_drop_in_classes():
    """``models.networks`` / ``models.blocks`` / ``models.networks_BE*`` (the reference's import paths, which its scripts and its
    pickled checkpoints name: train.py:9, test_BE.py:79-80) are aliases of the HIP-backed implementations."""
    import importlib
    import vae_play_amd
    for name, probe in (("networks", "Encoder"), ("networks", "VaeGan"), ("blocks", "Conv2d"), ("networks_BE", "ComposeNet"),
                        ("networks_BE_GAN", "Discriminator"), ("networks_BE_font", "ComposeNet")):
        alias = importlib.import_module(f"models.{name}")
        impl = importlib.import_module(f"vae_play_amd.{name}")
        assert getattr(alias, probe) is getattr(impl, probe), f"models.{name}.{probe}"
