"""Throughput of one font-GAN training iteration (train_BE_font.py:97-170): discriminator phase,
generator phase, style-encoder phase on the drop-in ComposeNet / Discriminator with three flat-arena Adam optimisers.
usage: python tools/bench_font.py [--img 64] [--batch 8] [--steps 5] [--cpu-steps 1]"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--img", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--precision", choices=["f32", "bf16x3"], default="f32", help="k x k convolutions: exact-f32 MFMA or split-bf16")
    a = ap.parse_args()
    import vae_play_amd.networks_BE as NB
    import vae_play_amd as _V
    _V.set_conv_precision(a.precision)
    import vae_play_amd.networks_BE_font as N
    from vae_play_amd import functional as Fh
    from vae_play_amd import optim, parallel
    # one process per GPU under torchrun (BASELINE config 5: 8 x MI355X): --batch images PER RANK; every phase all-reduces
    # (averages) the gradients it is about to apply; VP_BENCH_BACKEND=gloo rehearses it with several ranks on one GPU
    rank, world, local = parallel.init_from_env(os.environ.get("VP_BENCH_BACKEND"))
    if os.environ.get("VP_BENCH_BACKEND", "nccl") == "nccl":
        torch.cuda.set_device(local)
    dev = "cuda"
    torch.manual_seed(0)
    net = NB.initialize_model(N.ComposeNet(a.img))
    disc = NB.initialize_model(N.Discriminator(a.img, 2, 143))
    pn = {k: v.detach().clone() for k, v in net.state_dict().items()}
    pd = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    net, disc = net.to(dev).train(), disc.to(dev).train()
    opt, opt_style, opt_disc = optim.Adam(net.parameters(), lr=1e-4), optim.Adam(net.style_encoder.parameters(), lr=1e-4), \
        optim.Adam(disc.parameters(), lr=1e-4)
    dp = parallel.DataParallelGroup([opt, opt_style, opt_disc]) if world > 1 else None

    def apply(o):
        if dp is not None:
            dp.step(subset=[o])
        else:
            o.step()

    # synthetic inputs of the shapes train_BE_font.py feeds (generated here: the oracle is only the cpu_baseline leg below)
    gen = torch.Generator().manual_seed(8642 + rank)
    imgs = torch.rand(a.batch, 3, a.img, a.img, generator=gen)
    masks = (torch.rand(a.batch, 1, a.img, a.img, generator=gen) > 0.5).float()
    edges = (torch.rand(a.batch, 1, a.img, a.img, generator=gen) > 0.8).float()
    labels = torch.randint(0, 143, (a.batch,), generator=gen)
    cls = torch.zeros(a.batch, 143)
    cls[torch.arange(a.batch), labels] = 1
    y = {"cls": cls, "cnt_style": torch.rand(a.batch, 5, generator=gen)}
    dimgs, dmasks, dedges, dlabels = imgs.to(dev), masks.to(dev), edges.to(dev), labels.to(dev)
    dy = {k: v.to(dev) for k, v in y.items()}
    ones, zeros = torch.ones((a.batch, 1), device=dev), torch.zeros((a.batch, 1), device=dev)

    def iteration():
        with torch.no_grad():
            pr = net(dimgs, dy)
            pm = torch.cat([pr["masks"], pr["edges"]], dim=1)
        d_gt_adv, d_gt_aux = disc(torch.cat([dmasks, dedges], dim=1), dy)
        d_pr_adv, _ = disc(pm, dy)
        opt_disc.zero_grad(set_to_none=True)
        ((F.binary_cross_entropy(d_gt_adv, ones) + F.binary_cross_entropy(d_pr_adv, zeros)) * 0.5 + F.cross_entropy(d_gt_aux, dlabels)).backward()
        apply(opt_disc)
        pr = net(dimgs, dy)
        g_adv, g_aux = disc(torch.cat([pr["masks"], pr["edges"]], dim=1), dy)
        opt.zero_grad(set_to_none=True)
        l_gadv = F.binary_cross_entropy(g_adv, ones) * 2
        (NB.be_loss(pr["edges"], dedges) * 10 + NB.be_loss(pr["masks"], dmasks) * 10 + l_gadv + l_gadv * 5).backward()
        apply(opt)
        with torch.no_grad():
            ref = net(dimgs, dy)
        pr_ = net(dimgs)
        opt_style.zero_grad(set_to_none=True)
        l_embed = (Fh.l1_loss(pr_["masks"], ref["masks"]) + Fh.l1_loss(pr_["edges"], ref["edges"])) * 2.0
        (NB.be_loss(pr_["masks"], dmasks) + NB.be_loss(pr_["edges"], dedges) + l_embed).backward()
        apply(opt_style)
        return l_embed

    for _ in range(a.warmup):
        iteration()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        le = iteration()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / a.steps
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    out = {"metric": "images/sec (font GAN iteration, train_BE_font.py:97-170)", "value": round(world * a.batch / dt, 1), "unit": "images/sec",
           "n_gpus": world, "scaling": "weak",
           "ms_per_step": round(dt * 1e3, 2), "config": {"workload": f"ComposeNet({a.img}) + Discriminator({a.img}, 2, 143), batch {a.batch}",
                                                         "path": f"autograd modules on HIP kernels ({a.precision} convolutions)"}, "loss_embed": float(le.detach())}
    if world > 1:
        if rank == 0:
            print(json.dumps(out))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        return
    if True:      # per-entry-point roofline of the iteration (tools/op_roofline.py)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import op_roofline
        out["roofline"] = op_roofline.roofline(op_roofline.trace(iteration, 2))
    if a.cpu_steps > 0:
        from oracle import ref_cpu as O      # checker / CPU baseline only
        from oracle import ref_font as FN
        torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
        O.require_grad(pn); O.require_grad(pd)
        o1 = torch.optim.Adam([pn[n] for n in O.trainable_names(pn)], lr=1e-4)
        o2 = torch.optim.Adam([pn[n] for n in FN.style_encoder_names(pn)], lr=1e-4)
        o3 = torch.optim.Adam([pd[n] for n in O.trainable_names(pd)], lr=1e-4)
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            FN.train_iteration(pn, pd, o1, o3, o2, imgs, masks, edges, labels, y, a.img)
        ct = (time.perf_counter() - t0) / a.cpu_steps
        out["cpu_baseline"] = {"value": round(a.batch / ct, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{a.cpu_steps} iteration(s), no warm-up"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
