"""Throughput of the train_BE_GAN.py loop body (alt discriminator on blocks; train_BE_GAN.py:131-165) below the
backbone: generator = aux_convs (256 -> 64 channels at stride 4) + MaskNet + EdgeNet, discriminator = two MaskMappers + dense head,
D step + G step with the reference's two Adams, on synthetic tensors.  Prints one JSON line with `roofline` (the dominant kernel
family of the iteration by HIP-event time) and `cpu_baseline` (oracle/ref_be.gan_train_iteration on the host cores).
usage: python tools/bench_be_gan.py [--img 256] [--batch 16] [--steps 10] [--precision bf16x3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--feat", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default="bf16x3")
    a = ap.parse_args()
    import vae_play_amd as V
    import vae_play_amd.networks_BE as N
    import vae_play_amd.networks_BE_GAN as NG
    from vae_play_amd.train_be_gan import BEGanStep
    V.set_conv_precision(a.precision)
    dev = "cuda"
    torch.manual_seed(0)
    G = N.initialize_model(NG.ComposeNet(3, a.img, backbone=None, feature_channels=a.feat)).to(dev).train()
    D = N.initialize_model(NG.Discriminator(3, a.img, 5)).to(dev).train()
    step = BEGanStep(G, D, lr=1e-4)
    g = torch.Generator().manual_seed(1)
    H = a.img // 4
    feature = torch.randn(a.batch, a.feat, H, H, generator=g).to(dev)
    imgs = torch.rand(a.batch, 3, a.img, a.img, generator=g).to(dev)
    bimgs = (torch.rand(a.batch, 1, a.img, a.img, generator=g) > 0.5).float().to(dev)
    eimgs = (torch.rand(a.batch, 1, a.img, a.img, generator=g) > 0.8).float().to(dev)
    labels = torch.randint(0, 5, (a.batch,), generator=g).to(dev)
    for _ in range(a.warmup):
        step.step(feature, imgs, bimgs, eimgs, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step.step(feature, imgs, bimgs, eimgs, labels)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import op_roofline
    roof = op_roofline.roofline(op_roofline.trace(lambda: step.step(feature, imgs, bimgs, eimgs, labels), 2))
    out_rec = {"metric": "images/sec (train_BE_GAN.py:131-165 iteration below the backbone)", "value": round(a.batch / dt, 1), "unit": "images/sec",
               "ms_per_step": round(dt * 1e3, 3), "n_gpus": 1, "higher_is_better": True, "dtype": a.precision, "data": "synthetic",
               "config": {"workload": f"ComposeNet heads ({a.feat} -> 64 ch) + Discriminator, {a.img}x{a.img}, batch {a.batch}, D step + G step",
                          "path": f"autograd modules on HIP kernels ({a.precision} convolutions), train_be_gan.BEGanStep"},
               "losses": {k: round(float(v), 6) for k, v in out.items() if v.dim() == 0},
               "roofline": roof}
    if a.cpu_steps > 0:
        from oracle import ref_be as BE
        from oracle import ref_cpu as O
        threads = min(len(os.sched_getaffinity(0)), 16)
        torch.set_num_threads(threads)
        pg = {k: v.detach().cpu().clone() for k, v in G.state_dict().items()}
        pd = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
        O.require_grad(pg); O.require_grad(pd)
        go, do = BE.gan_make_optimizers(pg, pd, 1e-4)
        args = (feature.cpu(), imgs.cpu(), bimgs.cpu(), eimgs.cpu(), labels.cpu(), a.img, a.feat)
        BE.gan_train_iteration(pg, pd, go, do, *args)
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            BE.gan_train_iteration(pg, pd, go, do, *args)
        ct = (time.perf_counter() - t0) / a.cpu_steps
        out_rec["cpu_baseline"] = {"value": round(a.batch / ct, 2), "unit": "images/sec", "cores": threads, "kind": "port",
                                   "sample": f"{a.cpu_steps} iteration(s) after 1 warm-up, same shapes, torch CPU fp32"}
    print(json.dumps(out_rec))


if __name__ == "__main__":
    main()
