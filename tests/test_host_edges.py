"""Host-side edges: circle dataset, PNG grid writer, state_dict checkpoints.  CPU only, except
the optimiser round trip which needs the HIP arena."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch


def test_circle_dataset_matches_the_reference_formulas():
    from vae_play_amd import data
    np.random.seed(3)
    ds = data.CDataset(64, min_radius=5, data_size=16, ifGen=True)
    assert len(ds) == 16
    img, p = ds[5]
    assert tuple(img.shape) == (1, 64, 64) and img.dtype == torch.float32 and set(img.unique().tolist()) <= {0.0, 1.0}
    # tools/utils.py:13-22: 5 <= r < 27, the circle stays inside the image
    assert 5 <= p["radius"] < 27 and p["radius"] <= p["x"] <= 64 - p["radius"] and p["radius"] <= p["y"] <= 64 - p["radius"]
    yy, xx = np.mgrid[0:64, 0:64]
    ref = ((xx - p["x"]) ** 2 + (yy - p["y"]) ** 2 <= p["radius"] ** 2).astype(np.float32)
    assert np.array_equal(img[0].numpy(), ref)
    imgs, tg = data.CDataset.train_collate_fn([ds[i] for i in range(4)])
    assert tuple(imgs.shape) == (4, 1, 64, 64) and tuple(tg.shape) == (4, 3)
    q = ds.params[0]
    assert abs(tg[0, 0].item() - np.log(q["radius"] / 64)) < 1e-6 and abs(tg[0, 1].item() - (q["x"] - 32) / 32) < 1e-6
    dec = data.decode_circle_param(64, tg[:, 0], tg[:, 1], tg[:, 2])
    assert torch.allclose(dec["radius"], torch.tensor([float(ds.params[i]["radius"]) for i in range(4)]), atol=1e-4)
    rgb = data.generate_circle_img(32, 10, 12, 6, channel_size=3)
    assert rgb.shape == (32, 32, 3) and rgb.dtype == np.uint8 and rgb.max() == 255


def test_png_grid_writer_layout_and_encoding(tmp_path):
    from vae_play_amd import imageio
    x = torch.zeros(5, 1, 4, 6)
    for i in range(5):
        x[i] = (i + 1) / 5
    path = str(tmp_path / "g.png")
    imageio.save_image(x, path, nrow=3, padding=2, pad_value=1)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, color = struct.unpack(">IIBB", raw[16:26])
    assert (w, h, depth, color) == (3 * 8 + 2, 2 * 6 + 2, 8, 2)          # torchvision's grid geometry
    # decode the single IDAT chunk by hand
    pos, idat = 8, b""
    while pos < len(raw):
        n, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xFFFFFFFF)
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = zlib.decompress(idat)
    img = np.frombuffer(rows, np.uint8).reshape(h, 1 + w * 3)[:, 1:].reshape(h, w, 3)
    assert (img[0, 0] == 255).all()                                        # padding value 1.0
    assert (img[2, 2] == int(0.2 * 255 + 0.5)).all() and (img[2, 2 + 8] == int(0.4 * 255 + 0.5)).all()
    assert (img[2 + 6, 2] == int(0.8 * 255 + 0.5)).all()                   # second row starts with image 3
    assert (img[2 + 6, 2 + 16] == 255).all()                               # empty sixth cell keeps the pad value
    try:
        from PIL import Image
        assert Image.open(path).size == (w, h)
    except ImportError:
        pass


def test_module_checkpoint_round_trip(tmp_path):
    import vae_play_amd as V
    from vae_play_amd import checkpoint
    torch.manual_seed(1)
    a = V.VaeGan(32, 16)
    path = str(tmp_path / "c.ckpt")
    checkpoint.save_checkpoint(path, {"VAE": a}, None, epoch=7)
    torch.manual_seed(2)
    b = V.VaeGan(32, 16)
    assert checkpoint.load_checkpoint(path, {"VAE": b}) == 7
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(u, v), k
    blob = torch.load(path, weights_only=True)                             # nothing but tensors and numbers inside
    assert blob["format"] == "vae_play_amd/1" and "encoder.conv.0.conv.weight" in blob["networks"]["VAE"]


@pytest.mark.gpu
def test_optimizer_state_round_trip_on_device(tmp_path):
    import vae_play_amd as V
    from vae_play_amd import checkpoint, optim
    torch.manual_seed(1)
    net = V.DirectDecoder(16).cuda()
    opt = optim.Adam(net.parameters(), lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        net(torch.randn(4, 16, device="cuda")).sum().backward()
        opt.step()
    path = str(tmp_path / "o.ckpt")
    checkpoint.save_checkpoint(path, {"AUX": net}, {"AUX": opt}, epoch=1)
    net2 = V.DirectDecoder(16).cuda()
    opt2 = optim.Adam(net2.parameters(), lr=5e-2)
    checkpoint.load_checkpoint(path, {"AUX": net2}, {"AUX": opt2})
    assert opt2.step_count == 3 and opt2.lr == 1e-3
    assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    x = torch.randn(4, 16, device="cuda")
    for o, n in ((opt, net), (opt2, net2)):
        o.zero_grad()
        n(x).sum().backward()
        o.step()
    for p, q in zip(net.parameters(), net2.parameters()):
        assert torch.equal(p, q)


@pytest.mark.gpu
def test_optimizer_state_round_trip_with_foreign_parameters(tmp_path):
    """Two optimisers over overlapping parameters (train_BE_font.py:280-282: Adam over the whole generator and a second Adam
    over its style encoder): the second one owns no arena slice at all, its moments and per-tensor step counts are "foreign"
    state and must survive a checkpoint, as the reference's pickled optimisers do (train.py:157)."""
    import vae_play_amd as V
    from vae_play_amd import checkpoint, optim

    def build(seed):
        torch.manual_seed(seed)
        net = V.DirectDecoder(16).cuda()
        sub = [p for n, p in net.named_parameters() if n.startswith("r_fc")]
        return net, optim.Adam(net.parameters(), lr=1e-3), optim.Adam(sub, lr=2e-3, betas=(0.8, 0.95)), optim.RMSprop(sub, lr=1e-3, alpha=0.9)

    net, oa, ob, oc = build(1)
    assert ob.arena.numel == 0 and len(ob.arena.foreign) > 0
    for i in range(3):
        for o in (oa, ob, oc):
            o.zero_grad()
        net(torch.randn(4, 16, device="cuda")).sum().backward()
        (oa, ob, oc)[i % 3].step()
        ob.step()
    path = str(tmp_path / "f.ckpt")
    checkpoint.save_checkpoint(path, {"AUX": net}, {"a": oa, "b": ob, "c": oc}, epoch=2)
    net2, oa2, ob2, oc2 = build(5)
    checkpoint.load_checkpoint(path, {"AUX": net2}, {"a": oa2, "b": ob2, "c": oc2})
    assert ob2.betas == (0.8, 0.95) and oc2.alpha == 0.9
    for (_, m, v, c), (_, m2, v2, c2) in zip(ob._fstate, ob2._fstate):
        assert torch.equal(m, m2) and torch.equal(v, v2) and c == c2 and c[0] == 4
    x = torch.randn(4, 16, device="cuda")
    for n, a, b, c in ((net, oa, ob, oc), (net2, oa2, ob2, oc2)):
        for o in (a, b, c):
            o.zero_grad()
        n(x).sum().backward()
        b.step(); c.step(); a.step()
    for p, q in zip(net.parameters(), net2.parameters()):
        assert torch.equal(p, q)
    bad = oa.state_dict()
    with pytest.raises(ValueError):
        ob2.load_state_dict(bad)          # kind matches, arena size and foreign entries do not


@pytest.mark.gpu
def test_fused_step_counters_reach_the_checkpoint(tmp_path):
    """FusedVAEStep advances BatchNorm num_batches_tracked lazily; state_dict() (hence save_checkpoint) must see the true value."""
    import vae_play_amd as V
    from vae_play_amd import engine, optim
    vae = V.VAE(32, 16, 1).cuda().train()
    opt = optim.Adam(vae.parameters(), lr=1e-4)
    fused = engine.FusedVAEStep(vae, opt, 4, 32, 1)
    x, eps = torch.rand(4, 1, 32, 32, device="cuda"), torch.randn(4, 16, device="cuda")
    for _ in range(3):
        fused.step(x, eps)
    assert int(vae.state_dict()["encoder.conv.0.bn.num_batches_tracked"]) == 3
    bn = vae.encoder.conv[0].bn
    assert fused._bn_momentum_eps[id(bn)] == (bn.momentum, bn.eps)


def test_every_script_compiles():
    """tools/, tests/diag/, examples/, profiles/*.py and the two root entry points are scripts that only run on a GPU box:
    keep at least their syntax under the CPU suite."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    for sub in ("tools", os.path.join("tests", "diag"), "examples", "profiles"):
        files += sorted(glob.glob(os.path.join(root, sub, "*.py")))
    assert len(files) > 20
    for f in files:
        with open(f, "rb") as fh:
            compile(fh.read(), f, "exec")          # syntax only: nothing is executed or written
