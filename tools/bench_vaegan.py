"""Throughput of the VAE-GAN training step: VaeGan forward, VaeGan.loss, the five losses of train.py:61-66,
their backward passes, four flat-arena RMSprop steps -- as the pre-planned launch list of engine_gan.FusedVAEGANStep (--path fused,
default) or on the drop-in modules with loss.backward() (--path autograd).
usage: python tools/bench_vaegan.py [--img 128] [--z 128] [--batch 16] [--steps 10] [--cpu-steps 1]
Prints one JSON line; `cpu_baseline` is the oracle restatement on this host's CPU (test infrastructure)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def step(net, opts, x, targets, eps, z_p, V, lam, fused=True, dp=None):
    B = x.size(0)
    x_tilde, disc_class, disc_layer, mus, logvar, params = net(x, eps=eps, z_p=z_p)
    dl = (disc_layer[:B], disc_layer[B:-B], disc_layer[-B:])
    dc = (disc_class[:B], disc_class[B:-B], disc_class[-B:])
    nle, kl, mse, bo, bp, bs, l1 = V.VaeGan.loss(x, x_tilde, *dl, *dc, mus, logvar, targets, params)
    loss_recon = F.mse_loss(x, x_tilde)
    loss_encoder = torch.sum(kl) + torch.sum(mse)
    loss_disc = torch.sum(bo) + torch.sum(bp) + torch.sum(bs)
    loss_decoder = torch.sum(lam * mse) - (1.0 - lam) * loss_disc
    for o in opts:
        o.zero_grad(set_to_none=True)      # gradients are then written straight into the arena slices (functional._grad_out)
    if fused:
        V.VaeGan.backward_all(loss_recon, loss_encoder, loss_decoder, loss_disc, l1)
    else:
        loss_recon.backward(retain_graph=True)
        loss_encoder.backward(retain_graph=True)
        loss_decoder.backward(retain_graph=True)
        loss_disc.backward(retain_graph=True)
        l1.backward()
    if dp is not None:
        dp.step()           # under torchrun: one all-reduce per gradient arena, then the fused updates
    else:
        for o in opts:
            o.step()
    return loss_encoder


def step_gflop_per_image(img: int, z: int) -> float:
    """Algorithmic GFLOP of one VAE-GAN training step per image (2 x MACs; transposed convolutions without zero insertion;
    backward = 2 x forward; the discriminator sees three images per input image; dense layers included)."""
    import math
    L = int(math.log2(img // 8))
    mac = 0.0
    ch, s = 1, img                                  # encoder (1 image channel)
    for i in range(L):
        co = 64 if i == 0 else ch * 2
        s //= 2
        mac += s * s * 25 * ch * co
        ch = co
    size = ch
    mac += 64 * size * 1024 + 2 * 1024 * z
    dec = z * 64 * size                             # decoder, run twice (x_tilde and x_p)
    c, s = size, 8
    for i in range(L):
        co = size if i == 0 else c // 2
        dec += s * s * 25 * c * co
        c, s = co, s * 2
    dec += s * s * 25 * c * 1
    mac += 2 * dec
    d = img * img * 25 * 1 * 32                     # discriminator on (x, x_tilde, x_p)
    c, s = 32, img
    for i in range(L):
        s //= 2
        d += s * s * 25 * c * (2 * c)
        c *= 2
    d += 64 * c * 512 + 512
    mac += 3 * d
    mac += z * 512 + 512 * 256 + 256 * 128 + 128 * 64 + 2 * 64 * 32 + 32 * 3     # param_encoder
    return 3 * 2 * mac * 1e-9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img", type=int, default=128)
    ap.add_argument("--z", type=int, default=128)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default="bf16x3")
    ap.add_argument("--five-pass", action="store_true", help="the reference's five backward(retain_graph=True) calls instead of one")
    ap.add_argument("--path", choices=["fused", "autograd"], default="fused",
                    help="fused: engine_gan.FusedVAEGANStep (pre-planned launch list, bf16x3); autograd: the drop-in modules + loss.backward()")
    a = ap.parse_args()
    import vae_play_amd as V
    from vae_play_amd import optim, parallel
    # one process per GPU under torchrun (BASELINE config 4: 4 x MI355X): --batch images PER RANK, the four gradient arenas
    # all-reduced (averaged) before their updates; VP_BENCH_BACKEND=gloo rehearses it with several ranks on one GPU
    rank, world, local = parallel.init_from_env(os.environ.get("VP_BENCH_BACKEND"))
    if os.environ.get("VP_BENCH_BACKEND", "nccl") == "nccl":
        torch.cuda.set_device(local)
    dev = "cuda"
    V.set_conv_precision(a.precision)
    torch.manual_seed(0)
    net = V.VaeGan(a.img, a.z).to(dev).train()
    opts = [optim.RMSprop(m.parameters(), lr=1e-4) for m in (net.encoder, net.decoder, net.discriminator, net.param_encoder)]
    dp = parallel.DataParallelGroup(opts) if world > 1 else None
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.rand(a.batch, 1, a.img, a.img, generator=g).to(dev)
    targets = torch.rand(a.batch, 3, generator=g).to(dev)
    eps = torch.randn(a.batch, a.z, generator=g).to(dev)
    z_p = torch.randn(a.batch, a.z, generator=g).to(dev)
    fused = None
    if a.path == "fused":
        if a.precision != "bf16x3":
            raise SystemExit("--path fused runs the bf16x3 plan")
        from vae_play_amd.engine_gan import FusedVAEGANStep
        fused = FusedVAEGANStep(net, opts, a.batch, a.img, lambda_mse=1e-6)

        def do_step(net, opts, x, targets, eps, z_p, V, lam, one_pass=True, dp=None):      # same call shape as the autograd step
            fused.step(x, targets, eps, z_p)        # (several ranks: it all-reduces the four arenas itself)
            return fused.kl
    else:
        do_step = step
    for _ in range(a.warmup):
        do_step(net, opts, x, targets, eps, z_p, V, 1e-6, not a.five_pass, dp)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = do_step(net, opts, x, targets, eps, z_p, V, 1e-6, not a.five_pass, dp)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / a.steps
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    out = {"metric": "images/sec (VAE-GAN train step, train.py:43-78)", "value": round(world * a.batch / dt, 1), "unit": "images/sec",
           "n_gpus": world, "scaling": "weak",
           "ms_per_step": round(dt * 1e3, 3), "config": {"workload": f"VaeGan {a.img}x{a.img}x1 z={a.z} batch {a.batch} per rank",
                                                         "path": f"autograd modules on HIP kernels ({a.precision} convolutions)",
                                                         "backward": "five passes (train.py:69-73)" if a.five_pass else "one pass over the summed losses"},
           "loss_encoder": fused.losses()["loss_encoder"] if fused is not None else float(loss.detach())}
    if fused is not None:
        out["config"]["path"] = "engine_gan.FusedVAEGANStep: pre-planned launch list, weight gradients on a side stream (bf16x3 convolutions)"
        out["config"]["launches_per_step"] = len(fused._fwd.calls) + len(fused._fwd_disc.calls) + len(fused._bwd.calls)
    gf = step_gflop_per_image(a.img, a.z)
    ach = world * a.batch / dt * gf / 1e3 / world          # algorithmic TFLOP/s per GPU over the whole step
    peak = 2500.0 if a.precision == "bf16x3" else 157.3
    out["dtype"] = a.precision
    out["roofline"] = {"bound": "mfma", "kernel": ("whole step (fused plan; per-kernel table: profiles/*_vaegan_fused_summary*.md)" if fused is not None else
                                                   "whole step (autograd front end; per-kernel table: profiles/*_vaegan_summary.md)"),
                       "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                       "mfma_issue_frac": round(ach * (3 if a.precision == "bf16x3" else 1) / peak, 4), "traffic": None,
                       "step_gflop_per_image": round(gf, 3)}
    if world > 1:
        if rank == 0:
            print(json.dumps(out))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        return
    if a.cpu_steps > 0:
        from oracle import ref_cpu as O
        from oracle import ref_vaegan as G
        quota = None
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q, per = f.read().split()[:2]
                quota = None if q == "max" else max(1, int(float(q) / float(per) + 0.5))
        except (OSError, ValueError):
            pass
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), quota or 1 << 30))
        p = G.init_vaegan_params(a.img, a.z, seed=0)
        O.require_grad(p)
        o_opts = G.make_optimizers(p)
        xc, tc, ec, zc = x.cpu(), targets.cpu(), eps.cpu(), z_p.cpu()
        G.train_step(p, o_opts, xc, tc, ec, zc, a.img)
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            G.train_step(p, o_opts, xc, tc, ec, zc, a.img)
        ct = (time.perf_counter() - t0) / a.cpu_steps
        out["cpu_baseline"] = {"value": round(a.batch / ct, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{a.cpu_steps} step(s) after 1 warm-up"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
