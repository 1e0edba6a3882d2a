"""Micro-benchmark of the flat-arena optimiser kernels (vp_adam_f32 / vp_rmsprop_f32): achieved HBM GB/s
against the 7 (Adam) / 5 (RMSprop) fp32 streams per element.  usage: python tools/microbench_adam.py [n]"""
import sys
from ctypes import c_void_p

import torch

sys.path.insert(0, ".")
from vae_play_amd import _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
    dev = "cuda"
    p = torch.randn(n, device=dev)
    g = torch.randn(n, device=dev) * 0.01
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: c_void_p(t.data_ptr())

    def run(fn, streams):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        return ms, streams * 4 * n / ms / 1e6

    ms, gbs = run(lambda: _lib.call("vp_adam_f32", ptr(p), ptr(g), ptr(m), ptr(v), n, 1e-4, 0.9, 0.999, 1e-8, 3, 1.0, st), 7)
    print(f"adam    n={n}: {ms * 1e3:.1f} us  {gbs:.0f} GB/s")
    ms, gbs = run(lambda: _lib.call("vp_rmsprop_f32", ptr(p), ptr(g), ptr(v), n, 1e-4, 0.99, 1e-8, 1.0, st), 5)
    print(f"rmsprop n={n}: {ms * 1e3:.1f} us  {gbs:.0f} GB/s")
    q = torch.empty_like(p)
    ms, gbs = run(lambda: q.copy_(p), 2)
    print(f"copy    n={n}: {ms * 1e3:.1f} us  {gbs:.0f} GB/s")


if __name__ == "__main__":
    main()
