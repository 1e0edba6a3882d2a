"""Cost of the factored fc.0 weight gradient dW[1024][32768] = dh_all^T flat_all for W = 1, 2, 4, 8 ranks (K = 32 W rows):
the exact-f32 dense kernel vs the split-bf16 weight-gradient kernel run as a 1x1 'convolution' over W*B pixels.
usage: python tools/microbench_fc_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_play_amd import ops  # noqa: E402


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


N, F0 = 1024, 32768
for W in (1, 2, 4, 8):
    K = 32 * W
    dh, flat = torch.randn(K, N, device="cuda"), torch.randn(K, F0, device="cuda").clamp_min(0)
    out = torch.empty(N, F0, device="cuda")
    t_f32 = timed(lambda: ops.gemm(dh, 1, N, flat, 1, F0, N, F0, K, 2, out=out))
    ref = out.clone()

    def x3():
        big = ops.split_f32(flat)          # [K pixels][F0 channels]
        small = ops.split_f32(dh)
        return ops.conv_wgrad_bf16x3(big, (K, F0, 1, 1), small, (K, N, 1, 1), 1, 1)
    t_x3 = timed(x3)
    got = x3().view(N, F0)
    err = ((got - ref).norm() / ref.norm()).item()
    print(f"W={W} K={K}: f32 dense {t_f32:7.1f} us   split-bf16 wgrad (incl. the two split passes) {t_x3:7.1f} us   rel diff {err:.1e}")
