"""Generate tests/golden/font_*.npz: the font U-Net / AC-GAN discriminator rebuilt from the REAL reference's
models/blocks.py classes (models/networks_BE_font.py itself needs tkinter + torchvision and cannot be imported here)
pin oracle/ref_font.py bit-for-bit.  See oracle/ref_font.py for what is and is not pinned.

    python oracle/gen_golden_font.py
"""
from __future__ import annotations

import math
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_be as BE  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
from oracle import ref_font as FN  # noqa: E402
from oracle.gen_golden import bit_equal, import_reference, np_  # noqa: E402

L, S = FN.LABEL_EMBED, FN.STYLE_EMBED


# ---- the networks of models/networks_BE_font.py assembled from the reference's own blocks -----------------------
def embeding(bk, cin, cout):
    m = nn.Module()
    m.convs_first = nn.Sequential(bk.Linear(cin, cout, activate=None), bk.Linear(cout, cout, activate=None))
    m.attention = nn.Sequential(bk.SelfAttentionBlock(cout), bk.SelfAttentionBlock(cout), bk.SelfAttentionBlock(cout))
    m.embeding = nn.Sequential(bk.Linear(cout, cout, activate="lrelu"), bk.Linear(cout, cout, activate="lrelu"))

    def run(x):
        x = m.convs_first(x)
        x = m.attention(x.reshape(x.size(0), x.size(1), 1, 1))
        return m.embeding(x.reshape(x.size(0), -1))
    m.run = run
    return m


def style_encode(bk, cin, cout, in_size):
    m = nn.Module()
    convs = [bk.Conv2d(cin, 64, 3, stride=2, bn="instance")]
    c, nxt = 64, min(128, cout)
    for _ in range(int(math.log2(in_size)) - 3):
        convs.append(bk.Conv2d(c, nxt, 3, stride=2, bn="instance"))
        c, nxt = nxt, min(nxt * 2, cout)
    convs.append(bk.Conv2d(c, cout, 1, stride=1, bn="instance"))
    convs.append(nn.AdaptiveAvgPool2d((1, 1)))
    m.convs = nn.Sequential(*convs)
    m.run = lambda x: m.convs(x).reshape(x.size(0), -1)
    return m


def param_net(bk, kind, in_size):
    m = nn.Module()
    if kind == "embed":
        m.label_encode_block, m.style_encode_block = embeding(bk, 143, L), embeding(bk, 5, S)
    else:
        m.label_encode_block, m.style_encode_block = style_encode(bk, 3, L, in_size), style_encode(bk, 3, S, in_size)
    m.run = lambda a, b: (m.label_encode_block.run(a), m.style_encode_block.run(b))
    return m


def font_masknet(bk, c):
    m = nn.Module()
    m.predictor = nn.Sequential(bk.Conv2d(c, c, 3, stride=1, bn="instance"), bk.Conv2d(c, c, 3, stride=1, bn="instance"),
                                bk.Conv2d(c, 1, 3, stride=1, bn=None, activate=None))
    m.run = lambda x: m.predictor(x)
    return m


def compose_net(bk, in_size):
    m = nn.Module()
    repeat = int(math.log2(in_size // 4))
    m.down = nn.ModuleList([bk.Conv2d(3, 64, 3, stride=1, bn="instance")])
    c, nxt = 64, 128
    for _ in range(repeat):
        m.down.append(nn.Sequential(bk.Conv2d(c, nxt, 3, stride=2, bn="batch"), bk.Conv2d(nxt, nxt, 3, stride=1, bn="instance")))
        c, nxt = nxt, min(nxt * 2, 512)
    m.embeding_block = param_net(bk, "embed", in_size)
    m.style_encoder = param_net(bk, "image", in_size)
    relay = c * 16
    m.relay_convs = nn.Sequential(bk.Linear(relay + L + S, relay), bk.Linear(relay, relay))
    m.up, m.skip, m.cat = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
    c, nxt = 64, 128
    for _ in range(repeat):
        m.up.append(bk.Up(nxt, c))
        m.skip.append(bk.Conv2d(c, c, 3, stride=1, bn="instance"))
        m.cat.append(bk.Conv2d(c * 2, c, 3, stride=1, bn="instance"))
        c, nxt = nxt, min(nxt * 2, 512)
    m.mask_net, m.edge_net = font_masknet(bk, 64), font_masknet(bk, 64)

    def run(x, y=None):
        y_cls, y_sty = m.embeding_block.run(y["cls"], y["cnt_style"]) if y is not None else m.style_encoder.run(x, x)
        feats = []
        for d in m.down:
            x = d(x)
            feats.append(x)
        b, cc, h, w = x.shape
        x = m.relay_convs(torch.cat([x.reshape(b, -1), y_cls, y_sty], dim=1)).reshape(b, cc, h, w)
        for i in range(len(m.up)):
            idx = len(m.up) - 1 - i
            x = m.cat[idx](torch.cat([m.up[idx](x), m.skip[idx](feats[len(feats) - 2 - i])], dim=1))
        return {"edges": m.edge_net.run(x), "masks": m.mask_net.run(x)}
    m.run = run
    return m


def classifier(bk, in_size, cin, ncls):
    m = nn.Module()
    m.conv_first = bk.Conv2d(cin, 64, 3, stride=2, bn="instance", activate="lrelu")
    m.backbone = nn.Sequential(bk.Conv2d(64, 128, 3, stride=2, bn="instance", activate="lrelu"),
                               bk.Conv2d(128, 256, 3, stride=2, bn="instance", activate="lrelu"),
                               bk.Conv2d(256, 512, 3, stride=2, bn="batch", activate="lrelu"),
                               bk.Conv2d(512, 1024, 3, stride=2, bn="batch", activate="lrelu"))
    m.embeding_block = param_net(bk, "embed", in_size)
    n = 1024 * (in_size // 32) ** 2
    m.cls_convs = nn.Sequential(bk.Linear(n + L + S, n // 2, activate="lrelu"), bk.Linear(n // 2, n // 4, activate="lrelu"),
                                bk.Linear(n // 4, ncls, activate=None))

    def run(x, y):
        x = m.backbone(m.conv_first(x))
        a, b = m.embeding_block.run(y["cls"], y["cnt_style"])
        return m.cls_convs(torch.cat([x.reshape(x.size(0), -1), a, b], dim=1))
    m.run = run
    return m


def discriminator(bk, in_size, cin, ncls):
    m = nn.Module()
    m.adv_convs, m.aux_convs = classifier(bk, in_size, cin, 1), classifier(bk, in_size, cin, ncls)
    m.run = lambda x, y: (m.adv_convs.run(x, y).sigmoid(), m.aux_convs.run(x, y))
    return m


def checks(fx, tag, t):
    cs = O.checksum(t)
    fx[f"{tag}/sum"], fx[f"{tag}/l2"], fx[f"{tag}/samples"] = np_(cs["sum"]), np_(cs["l2"]), np_(cs["samples"])


def fwd_bwd_fixture(bk, in_size, B):
    """ComposeNet(in_size): forward + backward through both conditioning branches, seeded weights (gamma != 0)."""
    net = compose_net(bk, in_size)
    net.load_state_dict(FN.seeded_weights(net.state_dict(), 55))
    net.train()
    imgs, masks, edges, labels, y = FN.synthetic_batch(B, in_size)
    p = {k: v.detach().clone() for k, v in net.state_dict().items()}
    O.require_grad(p)
    fx = {"meta_S": in_size, "meta_B": B, "weight_seed": np.array(55)}
    g = torch.Generator().manual_seed(77)
    for branch, yy in (("embed", y), ("image", None)):
        out = net.run(imgs, yy)
        gm, ge = torch.randn(out["masks"].shape, generator=g), torch.randn(out["edges"].shape, generator=g)
        net.zero_grad()
        (out["masks"] * gm).sum().add((out["edges"] * ge).sum()).backward()
        oo = FN.compose_forward(p, imgs, yy, in_size)
        for n in O.trainable_names(p):
            p[n].grad = None
        (oo["masks"] * gm).sum().add((oo["edges"] * ge).sum()).backward()
        bit_equal(oo["masks"].detach(), out["masks"].detach(), f"{branch} masks")
        bit_equal(oo["edges"].detach(), out["edges"].detach(), f"{branch} edges")
        fx[f"{branch}/masks"], fx[f"{branch}/edges"] = np_(out["masks"]), np_(out["edges"])
        fx[f"{branch}/gm"], fx[f"{branch}/ge"] = np_(gm), np_(ge)
        rsd = net.state_dict(keep_vars=True)
        for n in O.trainable_names(p):
            rg, og = rsd[n].grad, p[n].grad
            if rg is None:
                assert og is None, n
                continue
            bit_equal(og, rg, f"{branch} grad {n}")
            checks(fx, f"{branch}/grad/{n}", rg)
    return fx


def disc_fixture(bk, in_size, B):
    d = discriminator(bk, in_size, 2, 143)
    d.load_state_dict(FN.seeded_weights(d.state_dict(), 66))
    d.train()
    _, masks, edges, labels, y = FN.synthetic_batch(B, in_size)
    x = torch.cat([masks, edges], dim=1).requires_grad_(True)
    adv, aux = d.run(x, y)
    g = torch.Generator().manual_seed(78)
    ga, gx = torch.randn(adv.shape, generator=g), torch.randn(aux.shape, generator=g)
    (adv * ga).sum().add((aux * gx).sum()).backward()
    p = {k: v.detach().clone() for k, v in d.state_dict().items()}
    for k in p:      # rewind the BN buffers the reference's forward advanced
        if k.endswith("running_mean"):
            p[k] = torch.zeros_like(p[k])
        elif k.endswith("running_var"):
            p[k] = torch.ones_like(p[k])
        elif k.endswith("num_batches_tracked"):
            p[k] = torch.zeros_like(p[k])
    O.require_grad(p)
    xo = x.detach().clone().requires_grad_(True)
    oadv, oaux = FN.discriminator_forward(p, xo, y)
    (oadv * ga).sum().add((oaux * gx).sum()).backward()
    bit_equal(oadv.detach(), adv.detach(), "disc adv")
    bit_equal(oaux.detach(), aux.detach(), "disc aux")
    bit_equal(xo.grad, x.grad, "disc dx")
    fx = {"meta_S": in_size, "meta_B": B, "weight_seed": np.array(66), "adv": np_(adv), "aux": np_(aux), "ga": np_(ga), "gx": np_(gx),
          "dx": np_(x.grad)}
    rsd = d.state_dict(keep_vars=True)
    for n in O.trainable_names(p):
        bit_equal(p[n].grad, rsd[n].grad, f"disc grad {n}")
        checks(fx, f"grad/{n}", rsd[n].grad)
    for n in p:
        if "running" in n:
            bit_equal(p[n], rsd[n], f"disc bn {n}")
            fx[f"bn/{n}"] = np_(p[n])
    return fx


def train_fixture(bk, in_size, B, iters):
    """train_BE_font.py:97-170 for `iters` iterations with the reference's blocks and torch.optim.Adam."""
    net, disc = compose_net(bk, in_size), discriminator(bk, in_size, 2, 143)
    net.load_state_dict(FN.seeded_weights(net.state_dict(), 55))
    disc.load_state_dict(FN.seeded_weights(disc.state_dict(), 66))
    net.train(); disc.train()
    lr = 1e-4
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    opt_style = torch.optim.Adam(net.style_encoder.parameters(), lr=lr)
    opt_disc = torch.optim.Adam(disc.parameters(), lr=lr)
    pn = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pd = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    O.require_grad(pn); O.require_grad(pd)
    oopt = torch.optim.Adam([pn[n] for n in O.trainable_names(pn)], lr=lr)
    oopt_style = torch.optim.Adam([pn[n] for n in FN.style_encoder_names(pn)], lr=lr)
    oopt_disc = torch.optim.Adam([pd[n] for n in O.trainable_names(pd)], lr=lr)
    imgs, masks, edges, labels, y = FN.synthetic_batch(B, in_size)
    fx = {"meta_S": in_size, "meta_B": B, "meta_iters": iters, "lr": np.array(lr)}
    for it in range(1, iters + 1):
        b = B
        # reference side, phase by phase as in train_BE_font.py
        gt = torch.cat([masks, edges], dim=1)
        with torch.no_grad():
            pr = net.run(imgs, y)
            pm = torch.cat([pr["masks"].detach(), pr["edges"].detach()], dim=1)
        d_gt_adv, d_gt_aux = disc.run(gt, y)
        d_pr_adv, _ = disc.run(pm, y)
        opt_disc.zero_grad()
        d_real = F.binary_cross_entropy(d_gt_adv, torch.ones((b, 1)))
        d_aux = F.cross_entropy(d_gt_aux, labels)
        d_fake = F.binary_cross_entropy(d_pr_adv, torch.zeros((b, 1)))
        ((d_real + d_fake) * 0.5 + d_aux).backward()
        opt_disc.step()
        pr = net.run(imgs, y)
        g_adv, g_aux = disc.run(torch.cat([pr["masks"], pr["edges"]], dim=1), y)
        opt.zero_grad()
        l_mask = BE.be_loss(pr["masks"], masks) * 10
        l_edge = BE.be_loss(pr["edges"], edges) * 10
        l_gadv = F.binary_cross_entropy(g_adv, torch.ones((b, 1))) * 2
        l_gaux = F.cross_entropy(g_aux, labels)
        l_gaux = l_gadv * 5
        (l_edge + l_mask + l_gadv + l_gaux).backward()
        opt.step()
        with torch.no_grad():
            ref = net.run(imgs, y)
        pr_ = net.run(imgs)
        opt_style.zero_grad()
        l_embed = (F.l1_loss(pr_["masks"], ref["masks"]) + F.l1_loss(pr_["edges"], ref["edges"])) * 2.0
        (BE.be_loss(pr_["masks"], masks) + BE.be_loss(pr_["edges"], edges) + l_embed).backward()
        opt_style.step()
        # oracle side
        o = FN.train_iteration(pn, pd, oopt, oopt_disc, oopt_style, imgs, masks, edges, labels, y, in_size)
        for k, v in (("d_adv_real", d_real), ("d_aux_real", d_aux), ("d_adv_fake", d_fake), ("loss_mask", l_mask), ("loss_edge", l_edge),
                     ("loss_g_adv", l_gadv), ("loss_embed", l_embed)):
            bit_equal(o[k], v.detach(), f"iter{it} {k}")
            fx[f"it{it}/{k}"] = np_(v.detach().double().reshape(1))
        bit_equal(o["masks"], pr["masks"].detach(), f"iter{it} masks")
        if it == 1:
            fx["it1/masks"], fx["it1/edges"] = np_(pr["masks"]), np_(pr["edges"])
        for tag, mod, pp in (("net", net, pn), ("disc", disc, pd)):
            rsd = mod.state_dict()
            for n in pp:
                bit_equal(pp[n].detach(), rsd[n], f"iter{it} {tag} {n}")
            for n in O.trainable_names(pp):
                checks(fx, f"it{it}/{tag}/{n}", pp[n])
    return fx


def main():
    _, bk = import_reference()
    outdir = os.path.join(ROOT, "tests", "golden")

    def save(name, fx):
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")

    save("font_compose16_b2", fwd_bwd_fixture(bk, 16, 2))
    save("font_disc32_b2", disc_fixture(bk, 32, 2))
    save("font_train32_b2", train_fixture(bk, 32, 2, 2))
    print("oracle == reference blocks composition (bit-exact) on every font fixture")


if __name__ == "__main__":
    main()
