"""Child process of tests/test_gpu_capture_guard.py: hipGraph capture of the AUTOGRAD front end must be refused with a Python
exception before anything is launched into the capture (round 2 left a core dump here: gpurun_out/be_graph.err).  A refused capture
must leave the process usable: the same step runs eagerly afterwards, and the pre-planned step still captures and replays."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import vae_play_amd as V
    from vae_play_amd import optim
    from vae_play_amd._lib import VaePlayHipError
    from vae_play_amd.engine import FusedVAEStep
    torch.manual_seed(0)
    vae = V.VAE(32, 16, 1).cuda().train()
    opt = optim.Adam(vae.parameters(), lr=1e-4)
    x, eps = torch.rand(4, 1, 32, 32, device="cuda"), torch.randn(4, 16, device="cuda")

    def autograd_step():
        opt.zero_grad()
        mu, logvar = vae.encoder(x)
        z = V.reparameterize(mu, logvar, eps=eps)
        xt = vae.decoder(z)
        loss = (V.binary_cross_entropy(xt, x, reduction="sum") + V.kl_divergence(mu, logvar).sum()) / x.shape[0]
        loss.backward()
        opt.step()
        return loss

    autograd_step()                       # eager: AccumulateGrad nodes now live on the default stream
    torch.cuda.synchronize()
    refused = False
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            autograd_step()
    except VaePlayHipError as e:
        refused = "capture of the autograd front end is not supported" in str(e)
    torch.cuda.synchronize()
    print("refused", refused)
    l1 = float(autograd_step())           # the process is intact: eager still works
    torch.cuda.synchronize()
    fused = FusedVAEStep(vae, opt, 4, 32, 1)
    fused.capture()                        # the pre-planned step is the capturable front end
    a = float(fused.forward_backward(x, eps)[0])
    b = float(fused.forward_backward(x, eps)[0])
    print("eager_loss", l1, "graph_replay_losses", a, b)
    assert refused and a == b and abs(a - l1) < 0.05 * abs(l1)
    print("OK")


if __name__ == "__main__":
    main()
