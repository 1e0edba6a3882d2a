"""GPU parity of the models/blocks.py drop-ins against vectors produced by the REAL reference
classes (tests/golden/blocks_*.npz, written by oracle/gen_golden.py): all 12 norm x activation variants of
Conv2d at kernel sizes 1/3/5 and strides 1/2, Up (with and without AddCoords, odd sizes), Down, Linear, AddCoords."""
import glob
import os

import pytest
import torch

from tests.util import GOLDEN, assert_close, load_golden, t

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _check(mod, g):
    sd = {k[len("param/"):]: t(v) for k, v in g.items() if k.startswith("param/")}
    missing = mod.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    mod.to(DEV).train()
    x = t(g["x"]).to(DEV).requires_grad_(True)
    y = mod(x)
    assert_close(y, t(g["y"]), TOL, "y")
    y.backward(t(g["gy"]).to(DEV))
    assert_close(x.grad, t(g["dx"]), 3 * TOL, "dx")
    for k, q in mod.named_parameters():
        assert_close(q.grad, t(g["grad/" + k]), 3 * TOL, "grad " + k)
    for k, v in mod.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            assert_close(v, t(g["after/" + k]), TOL, k)


CONV_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "blocks_conv2d_*.npz")))


@pytest.mark.parametrize("name", CONV_FIXTURES)
def test_blocks_conv2d_variants(name):
    from vae_play_amd import blocks
    g = load_golden(name)
    cin, cout, ks, stride = (int(v) for v in g["meta"])
    _, _, kstr, bn, act = name.split("_")
    bn = None if bn == "None" else bn
    act = None if act == "None" else act
    _check(blocks.Conv2d(cin, cout, ks, stride, bn, act), g)


def test_blocks_conv2d_fixture_coverage():
    combos = {tuple(n.split("_")[3:]) for n in CONV_FIXTURES}
    assert len(combos) == 12, "all 12 norm x activation variants must be covered"


@pytest.mark.parametrize("name,args", [("blocks_up_coord", (6, 4, True)), ("blocks_up", (6, 8, False))])
def test_blocks_up(name, args):
    from vae_play_amd import blocks
    _check(blocks.Up(*args), load_golden(name))


def test_blocks_down():
    from vae_play_amd import blocks
    _check(blocks.Down(6, 8, 3, True), load_golden("blocks_down_coord"))


@pytest.mark.parametrize("act", ["relu", "lrelu", "tanh", None])
def test_blocks_linear(act):
    from vae_play_amd import blocks
    _check(blocks.Linear(12, 7, True, act), load_golden(f"blocks_linear_{act}"))


@pytest.mark.parametrize("norm", [0, 1])
def test_blocks_add_coords(norm):
    from vae_play_amd import blocks
    g = load_golden(f"blocks_addcoords_{norm}")
    x = t(g["x"]).to(DEV).requires_grad_(True)
    y = blocks.AddCoords(bool(norm))(x)
    assert_close(y, t(g["y"]), 1e-6, "AddCoords")
    y.sum().backward()
    assert torch.equal(x.grad.cpu(), torch.ones_like(t(g["x"])))


def test_blocks_eval_mode_uses_running_stats():
    import torch.nn.functional as F
    from vae_play_amd import blocks
    torch.manual_seed(0)
    m = blocks.Conv2d(4, 6, 3, 1, "batch", "lrelu").to(DEV)
    x = torch.randn(2, 4, 9, 9, device=DEV)
    m.train(); m(x); m.eval()
    y = m(x)
    c = m.conv[0]; b = m.conv[1]
    ref = F.leaky_relu(F.batch_norm(F.conv2d(x.cpu(), c.weight.detach().cpu(), None, 1, 1), b.running_mean.cpu(), b.running_var.cpu(),
                                    b.weight.detach().cpu(), b.bias.detach().cpu(), False, 0.1, 1e-5), 0.02)
    assert_close(y, ref, TOL, "eval forward")
