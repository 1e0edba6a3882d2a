"""Coordinate search over the workgroup tile of each split-bf16 conv launch shape of the FUSED VAE-GAN step (128x128, 16 images) or,
with --vae B, of the fused VAE step at B images per GPU (bench.py's batch sweep), in one process: VP_TILE_OVERRIDE is read per
launch, candidates alternate with the incumbent (tools/search_tiles.py is the subprocess form for the VAE step at 32 images).  A
candidate that a launch rejects (statistics workspace sized for another tile) is skipped.
usage: python tools/search_tiles_gan.py [rounds] [steps] [--vae B]"""
import io
import os
import re
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VP_ENV_DYNAMIC", "1")      # the library then re-reads its per-launch knobs on every launch (csrc/env.h)


def main():
    argv = list(sys.argv[1:])
    vae_b = 0
    if "--vae" in argv:
        i = argv.index("--vae")
        vae_b = int(argv[i + 1])
        del argv[i:i + 2]
    rounds = int(argv[0]) if len(argv) > 0 else 4
    steps = int(argv[1]) if len(argv) > 1 else 20
    import vae_play_amd as V
    from vae_play_amd import optim
    torch.manual_seed(0)
    if vae_b:
        from vae_play_amd.engine import FusedVAEStep
        vae = V.VAE(128, 128, 3).cuda()
        st = FusedVAEStep(vae, optim.Adam(vae.parameters(), lr=1e-4), vae_b, 128, 3)
        args = (torch.rand(vae_b, 3, 128, 128, device="cuda"), torch.randn(vae_b, 128, device="cuda"))
    else:
        from vae_play_amd.engine_gan import FusedVAEGANStep
        net = V.VaeGan(128, 128).cuda().train()
        opts = [optim.RMSprop(m.parameters(), lr=1e-4) for m in (net.encoder, net.decoder, net.discriminator, net.param_encoder)]
        st = FusedVAEGANStep(net, opts, 16, 128, lambda_mse=1e-6)
        args = (torch.rand(16, 1, 128, 128, device="cuda"), torch.rand(16, 3, device="cuda"), torch.randn(16, 128, device="cuda"),
                torch.randn(16, 128, device="cuda"))
    for _ in range(5):
        st.step(*args)
    torch.cuda.synchronize()
    # launch shapes: the library logs them to stderr (fd 2) under VP_TILE_LOG
    r, w = os.pipe()
    saved = os.dup(2)
    os.dup2(w, 2)
    os.environ["VP_TILE_LOG"] = "1"
    st.step(*args)
    torch.cuda.synchronize()
    os.environ.pop("VP_TILE_LOG")
    os.dup2(saved, 2)
    os.close(w)
    log = os.read(r, 1 << 20).decode()
    shapes = sorted({m for m in re.findall(r"tile16 (\d+x\d+x\d+)", log)})
    print("launch shapes:", shapes, flush=True)

    def run(ov):
        if ov:
            os.environ["VP_TILE_OVERRIDE"] = ov
        else:
            os.environ.pop("VP_TILE_OVERRIDE", None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            st.step(*args)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    best = {}
    for key in shapes:
        M, N, gz = (int(v) for v in key.split("x"))
        if gz >= 13:          # weight-gradient family: tiles follow the channel counts
            continue
        for cand in ("128x128", "128x64", "64x64"):
            if (cand == "128x128" and N < 128) or (cand.startswith("128") and M < 128):
                continue
            ref = ",".join(f"{k}:{v}" for k, v in best.items())
            cfg = ",".join([f"{k}:{v}" for k, v in best.items()] + [f"{key}:{cand}"])
            try:
                run(cfg)
                a, b = [], []
                for _ in range(rounds):
                    a.append(run(ref))
                    b.append(run(cfg))
            except Exception as ex:  # noqa: BLE001
                print(f"{key:18s} {cand:8s} rejected: {str(ex)[:100]}", flush=True)
                continue
            a, b = sorted(a)[len(a) // 2], sorted(b)[len(b) // 2]
            flag = "<-- better" if b < a * 0.997 else ""
            print(f"{key:18s} {cand:8s} ref {a:.3f} cand {b:.3f} {flag}", flush=True)
            if flag:
                best[key] = cand
    print("best overrides:", best)


if __name__ == "__main__":
    main()
