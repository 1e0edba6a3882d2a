"""Throughput of the networks_BE heads step: aux_convs (256 -> 32 channels at stride 4) + MaskNet
+ EdgeNet + 0.5*BCEWithLogits + dice for both heads + backward + Adam, on a synthetic stride-4 feature map (the
torchvision backbone is out of scope).  usage: python tools/bench_be_heads.py [--img 256] [--batch 16] [--steps 10]"""
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
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default="f32", help="k x k convolutions: exact-f32 MFMA or split-bf16")
    a = ap.parse_args()
    import vae_play_amd.networks_BE as N
    import vae_play_amd as _V
    _V.set_conv_precision(a.precision)
    from vae_play_amd import optim
    dev = "cuda"
    torch.manual_seed(0)
    net = N.initialize_model(N.ComposeNet(N.FeatureNet(None, in_channels=256, target_out_channels=32))).to(dev).train()
    opt = optim.Adam(net.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1)
    H = a.img // 4
    feat = torch.randn(a.batch, 256, H, H, generator=g).to(dev)
    bimgs = (torch.rand(a.batch, 1, a.img, a.img, generator=g) > 0.5).float().to(dev)
    eimgs = (torch.rand(a.batch, 1, a.img, a.img, generator=g) > 0.8).float().to(dev)

    def step():
        out = net(feat)
        loss = N.be_loss(out["edges"], eimgs) + N.be_loss(out["masks"], bimgs)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    out = {"metric": "images/sec (networks_BE heads step, train_BE.py:54-64 below the backbone)", "value": round(a.batch / dt, 1),
           "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3),
           "config": {"workload": f"aux_convs 256->32 + MaskNet + EdgeNet, {a.img}x{a.img} targets, batch {a.batch}",
                      "path": f"autograd modules on HIP kernels ({a.precision} convolutions)"}, "loss": float(loss)}
    if True:      # per-entry-point roofline of the iteration (tools/op_roofline.py)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import op_roofline
        out["roofline"] = op_roofline.roofline(op_roofline.trace(step, 2))
    if a.cpu_steps > 0:
        from oracle import ref_be as BE
        from oracle import ref_cpu as O
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
        p = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        O.require_grad(p)
        oopt = O.make_optimizer(p, "adam", 1e-4)
        fc, bc, ec = feat.cpu(), bimgs.cpu(), eimgs.cpu()

        def cstep():
            for n in O.trainable_names(p):
                p[n].grad = None
            f = BE.aux_convs_forward(p, fc, 256, 32, True, "feature_net.aux_convs.")
            m = BE.masknet_forward(p, f, True, "mask_net.")
            e = BE.masknet_forward(p, f, True, "edge_net.")
            (BE.be_loss(e, ec) + BE.be_loss(m, bc)).backward()
            oopt.step()

        cstep()
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            cstep()
        ct = (time.perf_counter() - t0) / a.cpu_steps
        out["cpu_baseline"] = {"value": round(a.batch / ct, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{a.cpu_steps} step(s) after 1 warm-up"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
