"""Where does the concurrent schedule wait?  Times (HIP events) of the fused step's phases on the main stream and the
moment the side stream drains, relative to the start of the step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import vae_play_amd as V
    from vae_play_amd import optim
    from vae_play_amd.engine import FusedVAEStep
    torch.manual_seed(0)
    vae = V.VAE(128, 128, 3).cuda()
    opt = optim.Adam(vae.parameters(), lr=1e-4)
    st = FusedVAEStep(vae, opt, 32, 128, 3)
    x, eps = torch.rand(32, 3, 128, 128, device="cuda"), torch.randn(32, 128, device="cuda")
    for _ in range(30):
        st.step(x, eps)
    torch.cuda.synchronize()
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ("t0", "fwd", "dec", "a", "b_main", "side_done", "joined", "adam")}
    acc = {k: 0.0 for k in ev}
    N = 20
    for _ in range(N):
        s = torch.cuda.current_stream().cuda_stream
        side = st._side_ctx()
        st.x_nchw.copy_(x); st.eps.copy_(eps)
        ev["t0"].record()
        st._fwd.run(s, None, side=side); ev["fwd"].record()
        st._bwd_dec.run(s, None, side=side); ev["dec"].record()
        st._bwd_a.run(s, None)
        st._dhb[0].add_(st._dhb[1]); ev["a"].record()
        st._bwd_b.run(s, None, 0, None, side=side); ev["b_main"].record()
        ev["side_done"].record(side[0])
        torch.cuda.current_stream().wait_stream(side[0]); ev["joined"].record()
        st.opt.step(); ev["adam"].record()
        torch.cuda.synchronize()
        for k in ev:
            acc[k] += ev["t0"].elapsed_time(ev[k])
    for k in ev:
        print(f"{k:10s} {acc[k] / N:7.3f} ms")


if __name__ == "__main__":
    main()
