"""Like tools/ab_build.py for SEVERAL build-time switches at once: every argument is one configuration "VAR=val,VAR=val";
one FusedVAEStep (or FusedVAEGANStep with --gan) per configuration in the same process, timed alternately.
usage: python tools/ab_multi.py [--gan] [--rounds 6] [--steps 20] CONFIG CONFIG ..."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VP_ENV_DYNAMIC", "1")      # the library then re-reads its per-launch knobs on every launch (csrc/env.h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gan", action="store_true")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    import vae_play_amd as V
    from vae_play_amd import optim
    runs = {}
    for cfg in a.configs:
        env = dict(kv.split("=", 1) for kv in cfg.split(",") if kv)
        os.environ.update(env)
        torch.manual_seed(0)
        if a.gan:
            from vae_play_amd.engine_gan import FusedVAEGANStep
            net = V.VaeGan(128, 128).cuda().train()
            opts = [optim.RMSprop(m.parameters(), lr=1e-4) for m in (net.encoder, net.decoder, net.discriminator, net.param_encoder)]
            st = FusedVAEGANStep(net, opts, 16, 128, lambda_mse=1e-6)
            args = (torch.rand(16, 1, 128, 128, device="cuda"), torch.rand(16, 3, device="cuda"), torch.randn(16, 128, device="cuda"),
                    torch.randn(16, 128, device="cuda"))
        else:
            from vae_play_amd.engine import FusedVAEStep
            vae = V.VAE(128, 128, 3).cuda()
            st = FusedVAEStep(vae, optim.Adam(vae.parameters(), lr=1e-4), 32, 128, 3)
            args = (torch.rand(32, 3, 128, 128, device="cuda"), torch.randn(32, 128, device="cuda"))
        for _ in range(5):
            st.step(*args)
        runs[cfg] = (st, args, env, [])
        for k in env:
            os.environ.pop(k, None)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for cfg, (st, args, env, res) in runs.items():
            os.environ.update(env)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                st.step(*args)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / a.steps * 1e3)
            for k in env:
                os.environ.pop(k, None)
    for cfg, (_, _, _, res) in runs.items():
        r = sorted(res)
        print(f"{cfg}: median {r[len(r) // 2]:.3f} ms  min {r[0]:.3f}  max {r[-1]:.3f}")


if __name__ == "__main__":
    main()
