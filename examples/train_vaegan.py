"""End-to-end VAE-GAN training on MI355X with the drop-in back end: the loop of kungyao/vae-play's ``train.py``
(:18-106,109-161) on ``vae_play_amd`` -- generated circle dataset, VaeGan forward, VaeGan.loss, the five losses,
RMSprop per sub-network, PNG grids and ``state_dict`` checkpoints.  No network, no torchvision, no cv2.

    python examples/train_vaegan.py --epoch 1 --iters 50 --img_size 64 --batchsize 16 --res_output /tmp/vg_res --model_output /tmp/vg_ckpt
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import vae_play_amd as V  # noqa: E402
from vae_play_amd import checkpoint, imageio, optim  # noqa: E402
from vae_play_amd.data import CHANNEL_SIZE, CDataset, decode_circle_param, generate_batch_circle  # noqa: E402

LAMBDA_MSE = 1e-6   # train.py:15


def train(args, epoch, net, optims, loader, dev):
    avg = {k: 0.0 for k in ("loss_recon", "loss_encoder", "loss_discriminator", "loss_decoder", "loss_aux")}
    count = 0
    net.train()
    for i, (imgs, targets) in enumerate(loader):
        if args.iters and i >= args.iters:
            break
        B = imgs.size(0)
        imgs, targets = imgs.to(dev), targets.to(dev)
        x_tilde, disc_class, disc_layer, mus, log_variances, params = net(imgs)
        dl = (disc_layer[:B], disc_layer[B:-B], disc_layer[-B:])
        dc = (disc_class[:B], disc_class[B:-B], disc_class[-B:])
        nle, kl, mse, bo, bp, bs, l1 = V.VaeGan.loss(imgs, x_tilde, *dl, *dc, mus, log_variances, targets, params)
        losses = {"loss_recon": F.mse_loss(imgs, x_tilde), "loss_encoder": torch.sum(kl) + torch.sum(mse),
                  "loss_discriminator": torch.sum(bo) + torch.sum(bp) + torch.sum(bs)}
        losses["loss_decoder"] = torch.sum(LAMBDA_MSE * mse) - (1.0 - LAMBDA_MSE) * losses["loss_discriminator"]
        losses["loss_aux"] = l1
        for o in optims.values():
            o.zero_grad(set_to_none=True)
        V.VaeGan.backward_all(losses["loss_recon"], losses["loss_encoder"], losses["loss_decoder"], losses["loss_discriminator"],
                              losses["loss_aux"])
        for o in optims.values():
            o.step()
        nxt = count + B
        for k in avg:
            avg[k] = (avg[k] * count + losses[k].item()) / nxt
        count = nxt
        if (i + 1) % args.viz_freq == 0:
            print(f"epoch {epoch} iter {i + 1}: " + "; ".join(f"{k}: {v:.6f}" for k, v in avg.items()), flush=True)
            with torch.no_grad():
                x_tilde, _, _, _, _, params = net(imgs)
            rs, xs, ys = torch.unbind(params.cpu(), dim=-1)
            dec = decode_circle_param(args.img_size, rs, xs, ys)
            r = dec["radius"].clamp(1, args.img_size).round()
            from_params = generate_batch_circle(args.img_size, r, dec["x"].round(), dec["y"].round(), channel_size=CHANNEL_SIZE)
            imageio.save_image(torch.cat([imgs.cpu(), x_tilde.cpu(), from_params], dim=0),
                               os.path.join(args.res_output, f"{epoch}_{i}.png"), nrow=B, padding=2, pad_value=1)
    return avg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epoch", type=int, dest="epochs", default=20)
    ap.add_argument("--batchsize", type=int, default=16)
    ap.add_argument("--img_size", type=int, default=128)
    ap.add_argument("--zdim", type=int, default=128)
    ap.add_argument("--res_output", type=str, default="./results")
    ap.add_argument("--model_output", type=str, default="./logs")
    ap.add_argument("--viz_freq", type=int, default=16)
    ap.add_argument("--data_size", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=0, help="stop an epoch after this many iterations (0 = whole epoch)")
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default="bf16x3")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    os.makedirs(args.res_output, exist_ok=True)
    os.makedirs(args.model_output, exist_ok=True)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    dev = torch.device("cuda")
    V.set_conv_precision(args.precision)
    net = V.VaeGan(args.img_size, args.zdim, num_of_param=3).to(dev)
    optims = {"ENCODER": optim.RMSprop(net.encoder.parameters(), lr=1e-4), "DECODER": optim.RMSprop(net.decoder.parameters(), lr=1e-4),
              "DISCRIMINATOR": optim.RMSprop(net.discriminator.parameters(), lr=1e-4),
              "AUX": optim.RMSprop(net.param_encoder.parameters(), lr=1e-4)}
    data = CDataset(args.img_size, min_radius=max(2, args.img_size // 12), data_size=args.data_size, ifGen=True)
    loader = DataLoader(data, batch_size=args.batchsize, shuffle=True, num_workers=0, drop_last=True, collate_fn=CDataset.train_collate_fn)
    first = last = None
    for epoch in range(args.epochs):
        avg = train(args, epoch, net, optims, loader, dev)
        first = first or dict(avg)
        last = avg
        checkpoint.save_checkpoint(os.path.join(args.model_output, f"{epoch}.ckpt"), {"VAE": net}, optims, epoch)
    print("first epoch:", first)
    print("last epoch :", last)


if __name__ == "__main__":
    main()
