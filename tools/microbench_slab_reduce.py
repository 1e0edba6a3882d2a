"""Times vp_wgrad_slab_reduce_f32's two kernels on the weight-gradient shapes of the benchmark step (Cs, Cb, splits).
usage: python tools/microbench_slab_reduce.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_play_amd import ops  # noqa: E402

SHAPES = [(256, 256, 3), (256, 256, 6), (256, 128, 6), (128, 64, 12), (128, 64, 24), (64, 32, 32), (512, 256, 2)]
for Cs, Cb, ns in SHAPES:
    slab = torch.randn(ns, 25, Cs, Cb, device="cuda")
    out = torch.empty(Cs, Cb, 25, device="cuda")
    line = f"{Cs}x{Cb} x{ns} ({slab.numel() * 4 / 1e6:.1f} MB):"
    for v in (0, 1):
        for _ in range(5):
            ops.wgrad_slab_reduce(slab, Cs, Cb, 25, v, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.wgrad_slab_reduce(slab, Cs, Cb, 25, v, out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        line += f"  variant {v}: {us:6.1f} us ({(slab.numel() + out.numel()) * 4 / us / 1e6:.2f} TB/s)"
    print(line)
