"""The loop body of kungyao/vae-play's ``train_BE_GAN.py:131-165`` (the "alt discriminator on blocks") on the
HIP modules: a discriminator step -- feature-matching loss ``1 - mean|D(fake) - D(real)|`` + cross-entropy of the real type logits --
and a generator step -- ``2 loss_mask + 2 loss_edge + g_adv + g_type + 0.5 loss_cnt`` -- with the reference's two Adams
(``:236-237``: betas (0.5, 0.999), the discriminator at a tenth of the learning rate).

Everything below the backbone: ``networks_BE_GAN.ComposeNet`` is built with ``backbone=None`` and takes the backbone's stride-4 feature
map (torchvision's pretrained ResNet-50-FPN is out of scope).  Every loss is a HIP kernel behind the C ABI
(``functional.cross_entropy`` -> vp_cross_entropy_*, ``l1_loss`` -> vp_l1_mean_*, ``be_loss`` -> vp_be_loss_*, ``edge_loss`` ->
vp_dice_loss_* over the HIP 3x3 convolution); what stays on ATen is glue on tensors the kernels already produced: ``sigmoid`` of the
two predicted maps, ``torch.cat`` inside the discriminator, and scalar arithmetic on 0-d loss tensors.

Several ranks: the two optimisers are owned by one ``parallel.DataParallelGroup`` (one SUM all-reduce of the stepping optimiser's
gradient arena per phase, 1/W folded into the fused Adam) -- the reference has no distributed code.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import functional as Fh
from . import optim, parallel


class BEGanStep:
    """``step(feature, imgs, bimgs, eimgs, labels)`` = one iteration of train_BE_GAN.py:131-165; returns the seven scalars the
    reference logs (:167-175) as 0-d device tensors (no host sync)."""

    def __init__(self, generator: torch.nn.Module, discriminator: torch.nn.Module, lr: float = 1e-4, group=None):
        self.G, self.D = generator, discriminator
        self.g_opt = optim.Adam(generator.parameters(), lr=lr, betas=(0.5, 0.999))                 # train_BE_GAN.py:236
        self.d_opt = optim.Adam(discriminator.parameters(), lr=lr * 0.1, betas=(0.5, 0.999))       # :237
        self.group = group
        self.dp: Optional[parallel.DataParallelGroup] = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
            self.dp = parallel.DataParallelGroup([self.g_opt, self.d_opt], group)

    def _apply(self, opt) -> None:
        if self.dp is not None:
            self.dp.step(subset=[opt])
        else:
            opt.step()

    def step(self, feature: torch.Tensor, imgs: torch.Tensor, bimgs: torch.Tensor, eimgs: torch.Tensor, labels: torch.Tensor) -> Dict[str, torch.Tensor]:
        G, D = self.G, self.D
        # ---- D (:131-144) ----
        with torch.no_grad():
            preds = G(feature)
            pred_masks, pred_edges = preds["masks"].sigmoid(), preds["edges"].sigmoid()
        d_real_type, d_real_feats = D(imgs, bimgs, eimgs)
        d_fake_type, d_fake_feats = D(imgs, pred_masks, pred_edges)
        d_adv_loss = 1 - Fh.l1_loss(d_fake_feats, d_real_feats)
        d_type_loss = Fh.cross_entropy(d_real_type, labels)
        d_losses = d_adv_loss + d_type_loss
        self.d_opt.zero_grad()
        d_losses.backward()
        self._apply(self.d_opt)
        # ---- G (:147-165) ----
        preds = G(feature)
        pred_masks, pred_edges = preds["masks"], preds["edges"]
        with torch.no_grad():
            _, g_real_feats = D(imgs, bimgs, eimgs)
        pm, pe = pred_masks.sigmoid(), pred_edges.sigmoid()
        g_pred_type, g_pred_feats = D(imgs, pm, pe)
        loss_mask = Fh.be_loss(pred_masks, bimgs)
        loss_edge = Fh.be_loss(pred_edges, eimgs)
        g_adv_loss = Fh.l1_loss(g_pred_feats, g_real_feats)
        g_type_loss = Fh.cross_entropy(g_pred_type, labels)
        loss_cnt = Fh.edge_loss(pm, bimgs) + Fh.edge_loss(pe, eimgs)
        losses = loss_mask * 2 + loss_edge * 2 + g_adv_loss + g_type_loss + loss_cnt * 0.5
        self.g_opt.zero_grad()
        losses.backward()
        self._apply(self.g_opt)
        return {"d_adv_loss": d_adv_loss.detach(), "d_type_loss": d_type_loss.detach(), "loss_edge": loss_edge.detach(),
                "loss_mask": loss_mask.detach(), "g_adv_loss": g_adv_loss.detach(), "g_type_loss": g_type_loss.detach(),
                "loss_cnt": loss_cnt.detach(), "masks": pred_masks.detach(), "edges": pred_edges.detach()}
