"""Timing of the edge-layer kernels at the benchmark shape (B x 128 x 128): final conv forward with the tap-in-N MFMA kernel
(default) against the VALU kernel (VP_TAPN=0 in a second process).  usage: python tools/microbench_edge.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_play_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = ops.channels_last(torch.rand(B, 64, 128, 128, device="cuda"))
w = (torch.rand(3, 64, 5, 5, device="cuda") - 0.5) * 0.1
b = torch.zeros(3, device="cuda")
p0, _ = ops.pack_w5(w, True, False)
from vae_play_amd import _lib  # noqa: E402
y = ops.empty_cl(B, 3, 128, 128, x)


def tapn():
    _lib.call("vp_conv5_smallout_bf16x3", ops._p(x), ops._p(p0), ops._p(b), ops._p(y), B, 128, 128, 64, 3, 4, ops._stream())


for name, fn in (("tap-in-N MFMA (bf16x3)", tapn), ("VALU (exact f32)", lambda: ops.conv5_gather(x, p0, b, 1, 4))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"final conv fwd B={B} {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
