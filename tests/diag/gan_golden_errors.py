"""Diagnostic (not a test): per-tensor gradient errors of the fused VAE-GAN step against the reference's golden vectors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.util import load_golden, t
from tests.test_gpu_engine_gan import _build
from oracle import ref_cpu as O, ref_vaegan as G
from vae_play_amd.engine_gan import FusedVAEGANStep
for stats in ("1", "0"):
    os.environ["VP_FUSE_BN_STATS"] = stats
    for name in ("vaegan_32x32_z16_b4", "vaegan_64x64_z32_b4"):
        g = load_golden(name)
        S, z, B = (int(g[k]) for k in ("meta_S", "meta_z", "meta_B"))
        net, opts = _build(S, z)
        x, targets, eps, z_p = (t(g[k]).cuda() for k in ("x", "targets", "eps", "z_p"))
        fused = FusedVAEGANStep(net, opts, B, S, lambda_mse=G.LAMBDA_MSE)
        fused.forward_backward(x, targets, eps, z_p)
        worst = []
        for n, p in net.named_parameters():
            if n.startswith("discriminator."):
                continue
            gr = p._vp_arena.grad_view(p).detach().cpu().contiguous()
            if f"grad/{n}" in g:
                ref = t(g[f"grad/{n}"]).double()
                e = ((gr.double() - ref).pow(2).sum() / ref.pow(2).sum().clamp_min(1e-300)).sqrt().item()
                worst.append((e, 0.0, n, "full"))
            else:
                l2 = g[f"grad_l2/{n}"][0]
                idx = O.sample_indices(gr.numel())
                d = (gr.flatten()[idx].double() - t(g[f"grad_samples/{n}"])).abs().max().item()
                scale = max(l2 / gr.numel() ** 0.5, 1e-12)
                rel = abs(gr.double().pow(2).sum().sqrt().item() - l2) / (l2 + 1e-30)
                worst.append((rel, d / scale, n, "l2/sample"))
        worst.sort(reverse=True)
        print(f"stats={stats} {name}")
        for w in worst[:6]:
            print("   %.3e  sample/rms %.3e  %s (%s)" % w)
        print("   worst sample/rms:", max(w[1] for w in worst))
