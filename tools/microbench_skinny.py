"""Achieved weight-streaming rate of the skinny dense kernels (M = 32) for one matrix size at different row strides, cold
(a 512 MB buffer is written between launches so that neither L2 nor the Infinity Cache holds the weights).
usage: python tools/microbench_skinny.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, flush, reps=10):
    ts = []
    for _ in range(reps):
        flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    from vae_play_amd import ops
    flush = torch.zeros(128 * 1024 * 1024, device="cuda")
    M = 32
    for N, K in ((1024, 32768), (4096, 8192), (16384, 2048), (32768, 1024), (512, 65536), (8192, 4096)):
        x = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda")
        dy = torch.randn(M, N, device="cuda")
        mb = N * K * 4 / 1e6
        for name, fn in (("fwd", lambda: ops.linear_fwd(x, W, None)), ("dgrad", lambda: ops.linear_dgrad(dy, W)),
                         ("wgrad", lambda: ops.linear_wgrad(dy, x))):
            fn()
            us = timed(fn, flush)
            print(f"N={N:6d} K={K:6d} ({mb:.0f} MB) {name:6s} {us:7.1f} us  {mb / us:6.2f} TB/s (incl. split-K reduce and launch gaps)")


if __name__ == "__main__":
    main()
