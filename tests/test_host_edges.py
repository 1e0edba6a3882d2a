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
