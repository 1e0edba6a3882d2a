"""CPU tests of the font U-Net row: oracle/ref_font.py against vectors produced by the reference's
own blocks classes (tests/golden/font_*.npz) and the drop-in classes' state_dict keys.  No GPU compute."""
import torch

from tests.util import load_golden, t


def test_font_oracle_matches_golden_forward():
    from oracle import ref_font as FN
    import vae_play_amd.networks_BE_font as N
    g = load_golden("font_compose16_b2")
    S, B = int(g["meta_S"]), int(g["meta_B"])
    net = N.ComposeNet(S)
    p = FN.seeded_weights(net.state_dict(), int(g["weight_seed"]))
    assert list(p.keys()) == list(net.state_dict().keys())
    imgs, masks, edges, labels, y = FN.synthetic_batch(B, S)
    with torch.no_grad():
        out = FN.compose_forward(p, imgs, y, S)
    assert torch.allclose(out["masks"], t(g["embed/masks"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(out["edges"], t(g["embed/edges"]), rtol=1e-4, atol=1e-5)


def test_font_drop_in_keys_and_shapes():
    import vae_play_amd.networks_BE_font as N
    net = N.ComposeNet(32)
    sd = net.state_dict()
    for k in ("down.0.conv.0.weight", "down.3.0.conv.1.running_mean", "embeding_block.label_encode_block.attention.2.gamma",
              "style_encoder.style_encode_block.convs.0.conv.0.weight", "relay_convs.0.fc.0.weight", "up.2.conv.1.conv.1.weight",
              "skip.0.conv.0.weight", "cat.1.conv.0.weight", "mask_net.predictor.2.conv.0.bias", "edge_net.predictor.0.conv.0.weight"):
        assert k in sd, k
    assert tuple(sd["relay_convs.0.fc.0.weight"].shape) == (8192, 8192 + 512)
    assert tuple(sd["embeding_block.label_encode_block.convs_first.0.fc.0.weight"].shape) == (256, 143)
    d = N.Discriminator(32, 2, 143)
    dsd = d.state_dict()
    assert tuple(dsd["aux_convs.cls_convs.2.fc.0.weight"].shape) == (143, 256) and tuple(dsd["adv_convs.cls_convs.2.fc.0.weight"].shape) == (1, 256)
    assert "adv_convs.backbone.3.conv.1.running_var" in dsd
