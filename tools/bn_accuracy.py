"""BatchNorm+ReLU forward/backward of the HIP kernels against an fp64 evaluation (and torch CPU fp32 for scale).
usage: python tools/bn_accuracy.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_play_amd import functional as FH  # noqa: E402


def ref(x, gamma, beta, gy, dtype):
    x = x.detach().clone().to(dtype).requires_grad_(True)
    g = gamma.detach().clone().to(dtype).requires_grad_(True)
    b = beta.detach().clone().to(dtype).requires_grad_(True)
    C = x.shape[1]
    y = F.relu(F.batch_norm(x, torch.zeros(C, dtype=dtype), torch.ones(C, dtype=dtype), g, b, True, 0.9, 1e-5))
    y.backward(gy.to(dtype))
    return [t.detach().double() for t in (y, x.grad, g.grad, b.grad)]


gen = torch.Generator().manual_seed(3)
for shape, mean_scale, tag in (((4, 32, 64, 64), 0.0, "zero-mean"), ((4, 32, 64, 64), 5.0, "channel means ~5 sigma"),
                               ((4, 1024), 30.0, "BatchNorm1d batch 4, means ~30 sigma"), ((12, 512), 30.0, "BatchNorm1d batch 12"),
                               ((4, 64, 32, 32), 2.0, "means ~2 sigma")):
    C = shape[1]
    bshape = [1, C] + [1] * (len(shape) - 2)
    x = torch.randn(shape, generator=gen) + mean_scale * torch.randn(bshape, generator=gen)
    gamma, beta = torch.rand(C, generator=gen) + 0.5, torch.randn(C, generator=gen) * 0.2
    gy = torch.randn(shape, generator=gen) * 1e-5 + 3e-6           # small gradients with a common offset
    r64, r32 = ref(x, gamma, beta, gy, torch.float64), ref(x, gamma, beta, gy, torch.float32)
    xd = x.cuda().requires_grad_(True); gd = gamma.cuda().requires_grad_(True); bd = beta.cuda().requires_grad_(True)
    yd = FH.batch_norm_act(xd, gd, bd, torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), True, 0.9, 1e-5, "relu", 0.0)
    yd.backward(gy.cuda())
    ours = [t.detach().cpu().double() for t in (yd, xd.grad, gd.grad, bd.grad)]
    err = lambda a, b: ((a - b).norm() / b.norm()).item()
    print(f"{tag:40s} {str(shape):18s} rel-l2 vs fp64 (y, dx, dgamma, dbeta):  hip " + " ".join(f"{err(a, b):.1e}" for a, b in zip(ours, r64))
          + "   torch-cpu-fp32 " + " ".join(f"{err(a, b):.1e}" for a, b in zip(r32, r64)))
