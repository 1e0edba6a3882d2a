"""Kernel times: weight-gradient GEMM + Adam on the materialised gradient vs vp_adam_outer_f32 (encoder.fc.0 at config 3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_play_amd import ops  # noqa: E402

K, R, Cn = 32, 1024, 32768
dev = "cuda"
A, Bm = torch.randn(K, R, device=dev), torch.randn(K, Cn, device=dev)
p, m, v = torch.randn(R, Cn, device=dev), torch.zeros(R, Cn, device=dev), torch.zeros(R, Cn, device=dev)
g = torch.empty(R, Cn, device=dev)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_gemm = timeit(lambda: ops.gemm(A, 1, R, Bm, 1, Cn, R, Cn, K, 2, out=g))
t_adam = timeit(lambda: ops.adam_step(p.view(-1), g.view(-1), m.view(-1), v.view(-1), 1e-4, 0.9, 0.999, 1e-8, 3, 1.0))
t_outer = timeit(lambda: ops.adam_outer_step(p, m, v, A, Bm, 1e-4, 0.9, 0.999, 1e-8, 3, 1.0))
print(f"gemm {t_gemm:.1f} us + adam {t_adam:.1f} us = {t_gemm + t_adam:.1f} us ; adam_outer {t_outer:.1f} us")
