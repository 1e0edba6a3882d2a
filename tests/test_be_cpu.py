"""CPU tests of the networks_BE row: oracle/ref_be.py against the vectors produced by the
reference's own blocks classes (tests/golden/be_*.npz), drop-in state_dict keys, the restated dice / init rules
against torch primitives.  No GPU compute."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from tests.util import load_golden, t


def test_be_heads_oracle_matches_golden():
    from oracle import ref_be as BE
    from oracle import ref_cpu as O
    import vae_play_amd.networks_BE as N
    g = load_golden("be_heads_c32_b2_h16")
    C = int(g["meta_C"])
    holder = torch.nn.Module()
    holder.mask_net, holder.edge_net = N.MaskNet(C), N.EdgeNet(C)
    p = BE.seeded_weights(holder.state_dict(), int(g["weight_seed"]))
    assert list(p.keys()) == list(holder.state_dict().keys())
    O.require_grad(p)
    opt = O.make_optimizer(p, "adam", 1e-4)
    out = BE.heads_step(p, opt, t(g["feature"]), t(g["bimgs"]), t(g["eimgs"]))
    assert torch.allclose(out["masks"], t(g["masks"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(out["edges"], t(g["edges"]), rtol=1e-5, atol=1e-6)
    assert abs(out["loss_edge"].item() - g["loss_edge1"][0]) <= 1e-6 * abs(g["loss_edge1"][0])
    for n in O.trainable_names(p):
        assert torch.allclose(p[n].detach(), t(g[f"param1/{n}"]), rtol=1e-5, atol=1e-7), n


def test_compose_net_keys_and_shapes():
    import vae_play_amd.networks_BE as N
    net = N.ComposeNet(N.FeatureNet(None, in_channels=128, target_out_channels=32))
    keys = list(net.state_dict().keys())
    assert "feature_net.aux_convs.0.conv.0.weight" in keys and "feature_net.aux_convs.0.conv.1.running_mean" in keys
    assert "mask_net.conv1.conv.0.conv.0.weight" in keys and "edge_net.predictor.2.conv.0.bias" in keys
    sd = net.state_dict()
    assert tuple(sd["mask_net.conv1.conv.0.conv.0.weight"].shape) == (8, 34, 3, 3)      # 32 + 2 coordinate channels
    assert tuple(sd["mask_net.predictor.2.conv.0.weight"].shape) == (1, 4, 3, 3)
    assert net.feature_net.out_channels == 32 and net.mask_net.out_channels == 1


def test_restated_dice_and_init_rules_against_torch_primitives():
    """compute_dice_loss / initialize_model cannot be executed from the reference here (tools/ops.py needs cv2):
    the restatements are at least held to torch's own primitives."""
    from oracle import ref_be as BE
    import vae_play_amd.networks_BE as N
    g = torch.Generator().manual_seed(1)
    p = torch.rand(3, 1, 8, 8, generator=g)
    tt = (torch.rand(3, 1, 8, 8, generator=g) > 0.5).float()
    ref = 1 - sum((2 * (p[i] * tt[i]).sum() + 1) / (p[i].sum() + tt[i].sum() + 1) for i in range(3)) / 3
    assert abs(BE.dice_loss(p, tt).item() - ref.item()) < 1e-6
    net = N.MaskNet(32)
    torch.manual_seed(3)
    N.initialize_model(net)
    torch.manual_seed(3)
    ws = []
    for m in net.modules():
        if hasattr(m, "weight") and isinstance(m.weight, torch.nn.Parameter) and m.weight.dim() == 4:
            w = torch.empty_like(m.weight)
            torch.nn.init.kaiming_uniform_(w, mode="fan_in", nonlinearity="relu")
            ws.append((m.weight, w))
    for a, b in ws:
        assert torch.equal(a.detach(), b)
        fan_in = a.shape[1] * 9
        assert a.abs().max().item() <= math.sqrt(6.0 / fan_in) + 1e-7
    for m in net.modules():
        if getattr(m, "bias", None) is not None:
            assert float(m.bias.detach().abs().max()) == 0.0
