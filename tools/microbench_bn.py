"""Achieved HBM GB/s of the BatchNorm kernels on the flagship layer shapes (R rows x C channels, NHWC).
usage: python tools/microbench_bn.py"""
import os
import sys
from ctypes import c_void_p

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_play_amd import _lib, ops  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    lib = _lib.load()
    P = lambda t: None if t is None else c_void_p(t.data_ptr())
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    for R, C in ((524288, 64), (131072, 128), (131072, 64), (32768, 256), (8192, 512)):
        n = R * C
        x, dy = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
        ys = ops.empty_split(n, x)
        mean, rstd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        ws = torch.empty(lib.vp_bn_workspace_bytes(R, C) // 4 + 4, device="cuda")
        t_stats = timed(lambda: lib.vp_bn_stats_f32(P(x), R, C, 1e-5, 0.9, P(mean), P(rstd), None, None, P(ws), ws.numel() * 4, st))
        mean.zero_(); rstd.fill_(1.0)
        t_fwd = timed(lambda: lib.vp_bn_act_fwd_split_f32(P(x), P(mean), P(rstd), P(gamma), P(beta), None, P(ys), R, C, 1, 0.0, st))
        t_bwd = timed(lambda: lib.vp_bn_act_bwd_split_f32(P(x), P(dy), P(mean), P(rstd), P(gamma), P(beta), None, P(ys), P(dg), P(db),
                                                          R, C, 1, 0.0, 1, P(ws), ws.numel() * 4, st))
        gb = n * 4 / 1e9
        print(f"R={R:7d} C={C:4d}: stats {t_stats:6.1f} us ({gb / t_stats * 1e6:5.0f} GB/s)  fwd {t_fwd:6.1f} us ({2 * gb / t_fwd * 1e6:5.0f} GB/s)"
              f"  bwd(partial+final+apply) {t_bwd:6.1f} us ({5 * gb / t_bwd * 1e6:5.0f} GB/s)")


if __name__ == "__main__":
    main()
