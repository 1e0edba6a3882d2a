#!/usr/bin/env python3
"""bench.py -- images/sec of the full VAE training step (fwd + BCE/KL loss + bwd + gradient
all-reduce + Adam) on N MI355X GPUs, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json config 3): 128x128x3 images, latent 128, 32 images per GPU
(global batch 256 at 8 GPUs -> weak scaling), synthetic uniform[0,1) pixels, random-init weights
(VaeGan.init_parameters rule, seed 0), fp32 arithmetic on the f32 MFMA path, Adam(lr=1e-4).
Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel family of the step (the
5x5 implicit-GEMM convolutions, MFMA-bound): algorithmic FLOPs of its launches / their summed
HIP-event durations inside the timed region.  `cpu_baseline` times the CPU oracle
(oracle/ref_cpu.py = the reference's torch-CPU algorithm) on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic GFLOP per image for one training step (fwd + dgrad + wgrad, transposed
# convs without zero insertion)
STEP_GFLOP = {32: 0.6141, 64: 4.0301, 128: 22.2088, 256: 112.0225}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
CONV_CALLS = {"vp_conv5_gather_f32", "vp_conv5_scatter_f32", "vp_conv5_wgrad_f32",
              "vp_conv5_gather_bf16x3", "vp_conv5_scatter_bf16x3", "vp_conv5_wgrad_bf16x3"}


def measured_traffic(api_name):
    """HBM bytes per launch of a conv family from the committed PMC passes (profiles/*_traffic.json, newest round);
    None when no measurement is on file for that kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    return d.get(api_name), os.path.basename(files[-1])


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--img", type=int, default=128)
    ap.add_argument("--z", type=int, default=128)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured hipGraph "
                    "(roofline events are then taken in a separate instrumented pass)")
    ap.add_argument("--precision", choices=["bf16x3", "f32"], default="bf16x3",
                    help="bf16x3: split-bf16 MFMA (3 bf16 MFMAs per product, fp32 accumulate, ~1e-5 parity); "
                         "f32: exact fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-settle", action="store_true", help="do exactly --warmup untimed steps (no power-state settling)")
    ap.add_argument("--tags-out", type=str, default="", help="write per-layer conv timings (JSON) to this file")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (= reference algorithm on torch CPU) on the host cores: bounded sample."""
    from oracle import ref_cpu as O
    # threads = the CPU share this process may actually use (affinity), capped by VAEPLAY_CPU_THREADS
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("VAEPLAY_CPU_THREADS", min(avail, 16)))
    torch.set_num_threads(cores)
    B, C, S, z = args.batch_per_gpu, args.channels, args.img, args.z
    L = O.iter_level_for(S)
    p = O.init_params(C, z, L, seed=0)
    O.require_grad(p)
    opt = O.make_optimizer(p, "adam", 1e-4)
    x, eps = O.synthetic_batch(B, C, S, z)
    O.train_step(p, opt, x, eps, L)  # warm-up
    best = float("inf")
    for _ in range(args.cpu_steps):
        t0 = time.perf_counter()
        O.train_step(p, opt, x, eps, L)
        best = min(best, time.perf_counter() - t0)
    return {"value": round(B / best, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{args.cpu_steps} timed steps (best) after 1 warm-up, batch {B}, {S}x{S}x{C}, z={z}, "
                      f"torch {torch.__version__} CPU fp32, {cores} threads"}


SETTLE_STEPS = 200


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("VP_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the multi-rank path on one GPU
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import vae_play_amd as V
    from vae_play_amd import engine, optim, parallel

    B, C, S, z = args.batch_per_gpu, args.channels, args.img, args.z
    torch.manual_seed(0)
    vae = V.VAE(S, z, C, init_rule=True).to("cuda").train()     # same seed on every rank = replicated weights
    opt = optim.Adam(vae.parameters(), lr=1e-4)
    parallel.broadcast_flat_params(opt.flat_param, 0)
    fused = engine.FusedVAEStep(vae, opt, B, S, C, precision=args.precision)
    gx = torch.Generator().manual_seed(1234 + rank)
    ge = torch.Generator().manual_seed(4321 + rank)
    x = torch.rand(B, C, S, S, generator=gx).cuda()               # inputs resident in HBM before timing
    eps = torch.randn(B, z, generator=ge).cuda()
    if args.graph:
        fused.capture()

    # W untimed warm-up steps, then untimed "settle" steps up to SETTLE_STEPS in total: a fresh process starts with the
    # GPU in a low power state and a 0.2-s timed region taken right after a short warm-up measured 4.8-6.6 ms/step
    # for the same binary; after ~1 s of load successive runs agree within 4 % (profiles/r01_c_notes.md).
    # The count is fixed (not time-based) so that every rank of a multi-GPU run executes the same collectives.
    n_settle = max(0, SETTLE_STEPS - args.warmup) if not args.no_settle else 0
    for _ in range(args.warmup + n_settle):
        fused.step(x, eps)

    timers = None if args.graph else {"names": CONV_CALLS, "events": [], "pool": {}, "slot": 0}
    N_INST = 2      # instrumented steps inside the timed region (first and middle)
    if timers is not None:    # create the events (hipEventCreate is not free) outside the timed region
        for sl in range(N_INST):
            timers["slot"] = sl
            fused.step(x, eps, timers)
        torch.cuda.synchronize()
        timers["events"].clear()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP events bracket the convolution launches of TWO timed steps only (the first and the middle one), with
    # pre-created events: an event record is a barrier packet on the stream and 52 of them per step cost 0.4-2 ms of
    # pipeline bubbles, which made the figure depend on how many steps were instrumented (measured: 4.7 ms/step bare,
    # 5.1-6.6 ms with every step or every 5th step instrumented and events created on the fly).  The two instrumented
    # steps also run the SERIAL schedule (weight gradients on the main stream instead of the side stream) so that an
    # event pair times one kernel alone; the other K-2 steps run the concurrent schedule.
    inst = sorted({0, args.steps // 2})[:N_INST] if timers is not None else []
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in inst:
            timers["slot"] = inst.index(i)
            loss, recon, kl = fused.step(x, eps, timers)
        else:
            loss, recon, kl = fused.step(x, eps, None)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    final_loss = loss.item()

    measured = "timed-region"
    if timers is None:  # graph replay cannot carry events: instrumented eager pass of the same K steps
        timers = {"names": CONV_CALLS, "events": []}
        for _ in range(args.steps):
            fused.forward_backward(x, eps, timers)
        torch.cuda.synchronize()
        measured = "instrumented-pass-after-timed-region"

    if rank == 0:
        fam = {}
        for name, tag, flops, e0, e1 in timers["events"]:
            d = fam.setdefault(name, [0.0, 0.0, 0])
            d[0] += flops
            d[1] += e0.elapsed_time(e1) * 1e-3
            d[2] += 1
        if args.tags_out:
            per = {}
            for name, tag, flops, e0, e1 in timers["events"]:
                d = per.setdefault(tag, [0.0, 0.0, 0])
                d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1
            with open(args.tags_out, "w") as f:
                json.dump({k: {"ms": round(v[1] / v[2] * 1e3, 4), "gflop": round(v[0] / v[2] / 1e9, 3),
                               "tflops": round(v[0] / v[1] / 1e12, 2)} for k, v in per.items()}, f, indent=1)
        n_inst = len(inst) if measured == "timed-region" else args.steps
        dom = max(fam, key=lambda k: fam[k][1])
        tot_f = sum(v[0] for v in fam.values())
        tot_t = sum(v[1] for v in fam.values())
        ach = fam[dom][0] / fam[dom][1] / 1e12
        is16 = dom.endswith("bf16x3")
        peak = PEAK_BF16_MFMA_TFLOPS if is16 else PEAK_FP32_MFMA_TFLOPS
        kdesc = ("igemm16_kernel, 3 x v_mfma_f32_32x32x16_bf16 per product" if is16
                 else "igemm_kernel, v_mfma_f32_32x32x2_f32")
        # the committed PMC passes were taken on the default workload only
        traffic, traffic_src = measured_traffic(dom) if (S, C, z, B) == (128, 3, 128, 32) else (None, None)
        ips = world * B * args.steps / elapsed
        out = {
            "metric": f"images/sec (train step, {S}x{S} VAE)", "value": round(ips, 1), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": n_settle,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3" if args.precision == "bf16x3" else "f32", "data": "synthetic",
            "config": {"workload": f"networks VAE {S}x{S}x{C} latent={z} train step (fwd+BCE/KL+bwd+allreduce+Adam), "
                                   f"{B} images/GPU, global batch {B * world}", "parallelism": f"dp{world}",
                       "per_gpu_batch": B, "global_batch": B * world, "graph": bool(args.graph),
                       "precision": args.precision},
            "images_per_sec_per_gpu": round(ips / world, 1),
            "step_algorithmic_tflops_per_gpu": round(ips / world * STEP_GFLOP.get(S, 0.0) / 1e3, 1),
            "loss": round(final_loss, 4),
            "roofline": {"bound": "mfma", "kernel": f"{dom} ({kdesc})",
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": traffic, "traffic_unit": "HBM-side bytes per launch (PMC)",
                         "traffic_source": traffic_src,
                         "mfma_issue_frac": round(ach * (3 if is16 else 1) / peak, 4),
                         "launches": fam[dom][2], "avg_launch_ms": round(fam[dom][1] / fam[dom][2] * 1e3, 4),
                         "all_conv_families_tflops": round(tot_f / tot_t / 1e12, 2),
                         "conv_share_of_step_time": round(tot_t / n_inst / (elapsed / args.steps), 3),
                         "instrumented_steps": n_inst,
                         "measured": measured},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
