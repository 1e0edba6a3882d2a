"""Accuracy of the dense-layer kernels against an fp64 product (VP_GEMM_SKINNY=0|1). usage: python tools/skinny_accuracy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VP_ENV_DYNAMIC", "1")      # the library then re-reads its per-launch knobs on every launch (csrc/env.h)
from vae_play_amd import ops  # noqa: E402

for M, N, K in ((4, 1024, 8192), (32, 1024, 32768), (4, 8192, 64), (4, 64, 1024), (12, 512, 16384), (4, 1024, 16384), (12, 512, 4096),
                (4, 1024, 4096), (12, 256, 512), (4, 16384, 32), (12, 1024, 1024), (4, 4096, 1024)):
    g = torch.Generator().manual_seed(1)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    ref_f = (x.double() @ W.double().t() + b.double())
    ref_d = dy.double() @ W.double()
    for mode in ("0", "1"):
        os.environ["VP_GEMM_SKINNY"] = mode
        yf = ops.linear_fwd(x.cuda(), W.cuda(), b.cuda()).cpu().double()
        yd = ops.linear_dgrad(dy.cuda(), W.cuda()).cpu().double()
        ef = ((yf - ref_f).norm() / ref_f.norm()).item()
        ed = ((yd - ref_d).norm() / ref_d.norm()).item()
        print(f"M={M} N={N} K={K} skinny={mode}: fwd rel-l2 {ef:.2e} max {((yf - ref_f).abs().max() / ref_f.abs().max()).item():.2e}   "
              f"dgrad rel-l2 {ed:.2e}")
    cf = (x @ W.t() + b).double()
    print(f"   torch CPU fp32 fwd rel-l2 {((cf - ref_f).norm() / ref_f.norm()).item():.2e}")
