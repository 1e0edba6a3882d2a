"""In-process A/B of the fused VAE step under two values of an environment switch that the engine reads per step
(bench.py runs in separate processes differ by +-10 % on a shared box, which hides small effects).
usage: python tools/ab_env.py VAR valueA valueB [rounds] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VP_ENV_DYNAMIC", "1")      # the library then re-reads its per-launch knobs on every launch (csrc/env.h)


def main():
    var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
    import vae_play_amd as V
    from vae_play_amd import optim
    from vae_play_amd.engine import FusedVAEStep
    torch.manual_seed(0)
    vae = V.VAE(128, 128, 3).cuda()
    opt = optim.Adam(vae.parameters(), lr=1e-4)
    st = FusedVAEStep(vae, opt, 32, 128, 3)
    x, eps = torch.rand(32, 3, 128, 128, device="cuda"), torch.randn(32, 128, device="cuda")
    for v in (va, vb):
        os.environ[var] = v
        st.reload_switches()
        for _ in range(5):
            st.step(x, eps)
    torch.cuda.synchronize()
    res = {va: [], vb: []}
    for _ in range(rounds):
        for v in (va, vb):
            os.environ[var] = v
            st.reload_switches()          # the engine resolves its step-level switches when built, not per step
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st.step(x, eps)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / steps * 1e3)
    for v in (va, vb):
        r = sorted(res[v])
        print(f"{var}={v}: median {r[len(r) // 2]:.3f} ms  min {r[0]:.3f}  max {r[-1]:.3f}")


if __name__ == "__main__":
    main()
