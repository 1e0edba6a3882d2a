"""Generate tests/golden/be_*.npz: the networks_BE heads composed from the REAL reference's models/blocks.py classes
(models/networks_BE.py itself needs torchvision and cannot be imported here) pin oracle/ref_be.py bit-for-bit.
The loss terms come from torch (BCEWithLogits) and from the oracle's restatement of compute_dice_loss, whose
parity is unpinned (tools/ops.py needs cv2) -- see oracle/ref_be.py.

    python oracle/gen_golden_be.py
"""
from __future__ import annotations

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
from oracle.gen_golden import bit_equal, import_reference, np_  # noqa: E402


def ref_masknet(blocks, in_channel):
    """models/networks_BE.py:39-58 written with the reference's own blocks (attribute names = state_dict keys)."""
    m = nn.Module()
    m.conv1 = blocks.Up(in_channel, in_channel // 4, if_add_coord=True)
    m.conv2 = blocks.Up(in_channel // 4, in_channel // 8, if_add_coord=True)
    m.predictor = nn.Sequential(
        blocks.Conv2d(in_channel // 8, in_channel // 4, 3, stride=1, bn=None, activate=None),
        blocks.Conv2d(in_channel // 4, in_channel // 8, 3, stride=1, bn=None, activate=None),
        blocks.Conv2d(in_channel // 8, 1, 3, stride=1, bn=None, activate=None))
    m.run = lambda x: m.predictor(m.conv2(m.conv1(x)))
    return m


def heads_fixture(blocks, C, B, H, steps):
    holder = nn.Module()
    holder.mask_net = ref_masknet(blocks, C)
    holder.edge_net = ref_masknet(blocks, C)
    holder.load_state_dict(BE.seeded_weights(holder.state_dict(), 123))
    holder.train()
    g = torch.Generator().manual_seed(9)
    feature = torch.randn(B, C, H, H, generator=g)
    bimgs = (torch.rand(B, 1, 4 * H, 4 * H, generator=g) > 0.5).float()
    eimgs = (torch.rand(B, 1, 4 * H, 4 * H, generator=g) > 0.8).float()
    ropt = torch.optim.Adam(holder.parameters(), lr=1e-4)             # train_BE.py:131
    p = {k: v.detach().clone() for k, v in holder.state_dict().items()}
    O.require_grad(p)
    oopt = O.make_optimizer(p, "adam", 1e-4)
    fx = {"meta_C": C, "meta_B": B, "meta_H": H, "meta_steps": steps, "weight_seed": np.array(123),
          "feature": np_(feature), "bimgs": np_(bimgs), "eimgs": np_(eimgs)}
    for step in range(1, steps + 1):
        ropt.zero_grad()
        masks = holder.mask_net.run(feature)
        edges = holder.edge_net.run(feature)
        loss_edge = BE.be_loss(edges, eimgs)
        loss_mask = BE.be_loss(masks, bimgs)
        (loss_edge + loss_mask).backward()
        ropt.step()
        o = BE.heads_step(p, oopt, feature, bimgs, eimgs)
        bit_equal(o["masks"], masks.detach(), f"masks step{step}")
        bit_equal(o["edges"], edges.detach(), f"edges step{step}")
        rsd = holder.state_dict(keep_vars=True)
        for n in p:
            bit_equal(p[n].detach(), rsd[n].detach(), f"param {n} step{step}")
        if step == 1:
            for n in O.trainable_names(p):
                bit_equal(p[n].grad, rsd[n].grad, f"grad {n}")
                fx[f"grad/{n}"] = np_(rsd[n].grad)
            fx["masks"], fx["edges"] = np_(masks), np_(edges)
            for n in p:
                if "running" in n:
                    fx[f"bn/{n}"] = np_(p[n])
        fx[f"loss_edge{step}"] = np_(loss_edge.detach().double().reshape(1))
        fx[f"loss_mask{step}"] = np_(loss_mask.detach().double().reshape(1))
        for n in O.trainable_names(p):
            fx[f"param{step}/{n}"] = np_(p[n])
    return fx


def aux_fixture(blocks, Cin, target, B, H):
    layers = []
    c = Cin
    for _ in range(int(np.log2(Cin // target))):
        layers.append(blocks.Conv2d(c, c // 2, 1, stride=1, bn="batch"))
        layers.append(blocks.Conv2d(c // 2, c // 2, 3, stride=1, bn="batch"))
        c //= 2
    holder = nn.Module()
    holder.aux_convs = nn.Sequential(*layers)
    holder.load_state_dict(BE.seeded_weights(holder.state_dict(), 321))
    holder.train()
    g = torch.Generator().manual_seed(19)
    x = torch.randn(B, Cin, H, H, generator=g, requires_grad=True)
    y = holder.aux_convs(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    p = {k: v.detach().clone() for k, v in holder.state_dict().items()}
    # the reference's forward already advanced its BN buffers: rewind ours to the initial state
    for k in p:
        if k.endswith("running_mean"):
            p[k] = torch.zeros_like(p[k])
        elif k.endswith("running_var"):
            p[k] = torch.ones_like(p[k])
        elif k.endswith("num_batches_tracked"):
            p[k] = torch.zeros_like(p[k])
    O.require_grad(p)
    xo = x.detach().clone().requires_grad_(True)
    yo = BE.aux_convs_forward(p, xo, Cin, target, True)
    yo.backward(gy)
    bit_equal(yo.detach(), y.detach(), "aux y")
    bit_equal(xo.grad, x.grad, "aux dx")
    rsd = holder.state_dict(keep_vars=True)
    fx = {"meta_Cin": Cin, "meta_target": target, "weight_seed": np.array(321), "x": np_(x), "y": np_(y), "gy": np_(gy),
          "dx": np_(x.grad)}
    for n in O.trainable_names(p):
        bit_equal(p[n].grad, rsd[n].grad, f"aux grad {n}")
        fx[f"grad/{n}"] = np_(rsd[n].grad)
    for n in p:
        if "running" in n:
            bit_equal(p[n], rsd[n], f"aux bn {n}")
            fx[f"bn/{n}"] = np_(p[n])
    return fx


def gan_disc_fixture(blocks, in_size, B):
    """models/networks_BE_GAN.py:74-139 assembled from the reference's own blocks; forward + backward."""
    import math

    def mapper(cin, max_channel):
        m = nn.Module()
        m.convs = nn.Sequential(blocks.Conv2d(cin, 16, 3, 2, bn=None, activate="lrelu"), blocks.Conv2d(16, 32, 3, 2, bn=None, activate="lrelu"))
        c, nxt = 32, min(64, max_channel)
        m.feat_modules = nn.ModuleList()
        for _ in range(int(math.log2(in_size // 16)) - 2):
            m.feat_modules.append(nn.Sequential(blocks.Conv2d(c, nxt, 3, 2, bn="batch", activate="lrelu"),
                                                blocks.Conv2d(nxt, nxt, 3, 1, bn="batch", activate="lrelu")))
            c, nxt = nxt, min(nxt * 2, max_channel)
        m.pooler = nn.Sequential(blocks.Conv2d(c, max_channel, 1, 1, bn=None, activate=None), nn.AdaptiveAvgPool2d((1, 1)))

        def run(x, msk):
            x = m.convs(torch.cat([x, msk], dim=1))
            feats = []
            for idx, mod in enumerate(m.feat_modules):
                x = mod(x)
                feats.append(x.reshape(x.size(0), -1) * (idx // 2 + 1))
            x = m.pooler(x)
            return x.reshape(x.size(0), -1), torch.cat(feats, dim=1)
        m.run = run
        return m

    d = nn.Module()
    d.content_disc, d.boundary_disc = mapper(2, 64), mapper(2, 64)
    d.predictor = nn.Sequential(blocks.Linear(128, 128, bias=True, activate="lrelu"), blocks.Linear(128, 64, bias=True, activate="lrelu"),
                                blocks.Linear(64, 5, bias=False, activate=None))
    d.load_state_dict(BE.seeded_weights(d.state_dict(), 444))
    d.train()
    g = torch.Generator().manual_seed(29)
    x = torch.rand(B, 3, in_size, in_size, generator=g)
    m1 = torch.rand(B, 1, in_size, in_size, generator=g).requires_grad_(True)
    m2 = torch.rand(B, 1, in_size, in_size, generator=g).requires_grad_(True)
    x0 = x[:, 0, :, :].reshape(B, 1, in_size, in_size)
    a, fa = d.content_disc.run(x0, m1)
    b, fb = d.boundary_disc.run(x0, m2)
    logits, feats = d.predictor(torch.cat([a, b], dim=1)), torch.cat([fa, fb], dim=1)
    gl, gf = torch.randn(logits.shape, generator=g), torch.randn(feats.shape, generator=g) * 0.01
    ((logits * gl).sum() + (feats * gf).sum()).backward()
    p = {k: v.detach().clone() for k, v in d.state_dict().items()}
    for k in p:
        if k.endswith("running_mean"):
            p[k] = torch.zeros_like(p[k])
        elif k.endswith("running_var"):
            p[k] = torch.ones_like(p[k])
        elif k.endswith("num_batches_tracked"):
            p[k] = torch.zeros_like(p[k])
    O.require_grad(p)
    m1o, m2o = m1.detach().clone().requires_grad_(True), m2.detach().clone().requires_grad_(True)
    lo, fo = BE.gan_discriminator_forward(p, x, m1o, m2o, in_size, True)
    ((lo * gl).sum() + (fo * gf).sum()).backward()
    bit_equal(lo.detach(), logits.detach(), "gan disc logits")
    bit_equal(fo.detach(), feats.detach(), "gan disc feats")
    bit_equal(m1o.grad, m1.grad, "gan disc dm1")
    bit_equal(m2o.grad, m2.grad, "gan disc dm2")
    fx = {"meta_S": in_size, "meta_B": B, "weight_seed": np.array(444), "x": np_(x), "m1": np_(m1), "m2": np_(m2), "logits": np_(logits),
          "feats_stride7": np_(feats.detach().flatten()[::7][:8192]), "gl": np_(gl), "gf": np_(gf), "dm1": np_(m1.grad), "dm2": np_(m2.grad)}
    rsd = d.state_dict(keep_vars=True)
    for n in O.trainable_names(p):
        bit_equal(p[n].grad, rsd[n].grad, f"gan disc grad {n}")
        fx[f"grad/{n}"] = np_(rsd[n].grad)
    for n in p:
        if "running" in n:
            bit_equal(p[n], rsd[n], f"gan disc bn {n}")
            fx[f"bn/{n}"] = np_(p[n])
    return fx


def ref_gan_discriminator(blocks, in_size, num_classes=5):
    """models/networks_BE_GAN.py:74-139 from the reference's own blocks (as in gan_disc_fixture)."""
    import math

    def mapper(cin, max_channel):
        m = nn.Module()
        m.convs = nn.Sequential(blocks.Conv2d(cin, 16, 3, 2, bn=None, activate="lrelu"), blocks.Conv2d(16, 32, 3, 2, bn=None, activate="lrelu"))
        c, nxt = 32, min(64, max_channel)
        m.feat_modules = nn.ModuleList()
        for _ in range(int(math.log2(in_size // 16)) - 2):
            m.feat_modules.append(nn.Sequential(blocks.Conv2d(c, nxt, 3, 2, bn="batch", activate="lrelu"),
                                                blocks.Conv2d(nxt, nxt, 3, 1, bn="batch", activate="lrelu")))
            c, nxt = nxt, min(nxt * 2, max_channel)
        m.pooler = nn.Sequential(blocks.Conv2d(c, max_channel, 1, 1, bn=None, activate=None), nn.AdaptiveAvgPool2d((1, 1)))

        def run(x, msk):
            x = m.convs(torch.cat([x, msk], dim=1))
            feats = []
            for idx, mod in enumerate(m.feat_modules):
                x = mod(x)
                feats.append(x.reshape(x.size(0), -1) * (idx // 2 + 1))
            x = m.pooler(x)
            return x.reshape(x.size(0), -1), torch.cat(feats, dim=1)
        m.run = run
        return m

    d = nn.Module()
    d.content_disc, d.boundary_disc = mapper(2, 64), mapper(2, 64)
    d.predictor = nn.Sequential(blocks.Linear(128, 128, bias=True, activate="lrelu"), blocks.Linear(128, 64, bias=True, activate="lrelu"),
                                blocks.Linear(64, num_classes, bias=False, activate=None))

    def run(x, m1, m2):
        x0 = x[:, 0, :, :].reshape(x.size(0), 1, x.size(2), x.size(3))
        a, fa = d.content_disc.run(x0, m1)
        b, fb = d.boundary_disc.run(x0, m2)
        return d.predictor(torch.cat([a, b], dim=1)), torch.cat([fa, fb], dim=1)
    d.run = run
    return d


def gan_train_fixture(blocks, in_size, B, feat_ch, iters):
    """train_BE_GAN.py:131-165 with the reference's blocks modules below the backbone, torch's own cross-entropy /
    BCE-with-logits / Adam, next to oracle/ref_be.gan_train_iteration: bit equality of all seven losses and of every
    parameter after every iteration."""
    gen = nn.Module()
    layers, c = [], feat_ch
    for _ in range(int(np.log2(feat_ch // 64))):
        layers.append(blocks.Conv2d(c, c // 2, 1, stride=1, bn="batch"))
        layers.append(blocks.Conv2d(c // 2, c // 2, 3, stride=1, bn="batch"))
        c //= 2
    gen.aux_convs = nn.Sequential(*layers)
    gen.mask_net, gen.edge_net = ref_masknet(blocks, 64), ref_masknet(blocks, 64)
    gen.load_state_dict(BE.seeded_weights(gen.state_dict(), 555))
    disc = ref_gan_discriminator(blocks, in_size)
    disc.load_state_dict(BE.seeded_weights(disc.state_dict(), 666))
    gen.train(); disc.train()
    g = torch.Generator().manual_seed(41)
    H = in_size // 4
    feature = torch.randn(B, feat_ch, H, H, generator=g)
    imgs = torch.rand(B, 3, in_size, in_size, generator=g)
    bimgs = (torch.rand(B, 1, in_size, in_size, generator=g) > 0.5).float()
    eimgs = (torch.rand(B, 1, in_size, in_size, generator=g) > 0.8).float()
    labels = torch.randint(0, 5, (B,), generator=g)
    lr = 1e-4
    g_ropt = torch.optim.Adam(gen.parameters(), lr=lr, betas=(0.5, 0.999))          # train_BE_GAN.py:236
    d_ropt = torch.optim.Adam(disc.parameters(), lr=lr * 0.1, betas=(0.5, 0.999))   # :237
    pg = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    pd = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    O.require_grad(pg); O.require_grad(pd)
    g_oopt, d_oopt = BE.gan_make_optimizers(pg, pd, lr)

    def gen_run(f):
        h = gen.aux_convs(f)
        return {"masks": gen.mask_net.run(h), "edges": gen.edge_net.run(h)}
    fx = {"meta_S": in_size, "meta_B": B, "meta_C": feat_ch, "meta_iters": iters, "gen_seed": np.array(555), "disc_seed": np.array(666),
          "feature": np_(feature), "imgs": np_(imgs), "bimgs": np_(bimgs), "eimgs": np_(eimgs), "labels": labels.numpy().copy()}
    names = ("d_adv_loss", "d_type_loss", "loss_edge", "loss_mask", "g_adv_loss", "g_type_loss", "loss_cnt")
    for it in range(1, iters + 1):
        # ---- the reference's loop body, verbatim in structure (train_BE_GAN.py:131-165) ----
        with torch.no_grad():
            preds = gen_run(feature)
            pred_masks = preds["masks"].sigmoid()
            pred_edges = preds["edges"].sigmoid()
        d_real_type, d_real_feats = disc.run(imgs, bimgs, eimgs)
        d_fake_type, d_fake_feats = disc.run(imgs, pred_masks, pred_edges)
        d_adv_loss = 1 - torch.mean(torch.abs(d_fake_feats - d_real_feats))
        d_type_loss = F.cross_entropy(d_real_type, labels)
        d_losses = d_adv_loss + d_type_loss
        d_ropt.zero_grad()
        d_losses.backward()
        d_ropt.step()
        preds = gen_run(feature)
        pred_masks, pred_edges = preds["masks"], preds["edges"]
        with torch.no_grad():
            _, g_real_feats = disc.run(imgs, bimgs, eimgs)
        g_pred_type, g_pred_feats = disc.run(imgs, pred_masks.sigmoid(), pred_edges.sigmoid())
        loss_mask = 0.5 * F.binary_cross_entropy_with_logits(pred_masks, bimgs) + BE.dice_loss(pred_masks.sigmoid(), bimgs)
        loss_egde = 0.5 * F.binary_cross_entropy_with_logits(pred_edges, eimgs) + BE.dice_loss(pred_edges.sigmoid(), eimgs)
        g_adv_loss = torch.mean(torch.abs(g_pred_feats - g_real_feats))
        g_type_loss = F.cross_entropy(g_pred_type, labels)
        loss_cnt = BE.edge_loss(pred_masks.sigmoid(), bimgs) + BE.edge_loss(pred_edges.sigmoid(), eimgs)
        losses = loss_mask * 2 + loss_egde * 2 + g_adv_loss + g_type_loss + loss_cnt * 0.5
        g_ropt.zero_grad()
        losses.backward()
        g_ropt.step()
        ref = {"d_adv_loss": d_adv_loss, "d_type_loss": d_type_loss, "loss_edge": loss_egde, "loss_mask": loss_mask,
               "g_adv_loss": g_adv_loss, "g_type_loss": g_type_loss, "loss_cnt": loss_cnt}
        # ---- the oracle ----
        o = BE.gan_train_iteration(pg, pd, g_oopt, d_oopt, feature, imgs, bimgs, eimgs, labels, in_size, feat_ch)
        for k in names:
            bit_equal(o[k], ref[k].detach(), f"{k} iteration {it}")
            fx[f"{k}{it}"] = np_(ref[k].detach().double().reshape(1))
        bit_equal(o["masks"], pred_masks.detach(), f"masks iteration {it}")
        for net, pp, tag in ((gen, pg, "g"), (disc, pd, "d")):
            rsd = net.state_dict()
            for n in pp:
                bit_equal(pp[n].detach(), rsd[n].detach(), f"{tag} param {n} iteration {it}")
        if it == 1:
            fx["masks1"], fx["edges1"] = np_(pred_masks.detach()), np_(pred_edges.detach())
        for tag, pp in (("g", pg), ("d", pd)):
            for n in O.trainable_names(pp):
                t_ = pp[n].detach()
                fx[f"param{it}/{tag}/{n}"] = np_(torch.stack([t_.double().sum(), t_.double().pow(2).sum()]))
                fx[f"psample{it}/{tag}/{n}"] = np_(t_.flatten()[:: max(1, t_.numel() // 64)][:64])
    for tag, pp in (("g", pg), ("d", pd)):
        for n in pp:
            if "running" in n:
                fx[f"bn/{tag}/{n}"] = np_(pp[n])
    return fx


def main():
    _, blocks = import_reference()
    outdir = os.path.join(ROOT, "tests", "golden")

    def save(name, fx):
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **fx)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")

    save("be_heads_c32_b2_h16", heads_fixture(blocks, 32, 2, 16, 2))
    save("be_aux_c128_to32", aux_fixture(blocks, 128, 32, 2, 12))
    save("be_gan_disc128_b2", gan_disc_fixture(blocks, 128, 2))
    save("be_gan_train128_b2", gan_train_fixture(blocks, 128, 2, 128, 2))
    print("oracle == reference blocks composition (bit-exact) on every BE fixture")


if __name__ == "__main__":
    main()
