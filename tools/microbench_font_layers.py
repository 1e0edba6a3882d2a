"""Per-layer throughput of the font U-Net's 3x3 convolutions (BASELINE config 5: ComposeNet(256), 64 images) on the split-bf16
kernels: forward (gather), input gradient (scatter), weight gradient, each in algorithmic TFLOP/s.
usage: python tools/microbench_font_layers.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_play_amd import ops  # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


# (name, Cin, Cout, H of the input, stride)
LAYERS = [("skip.0 / up.0.1 / heads", 64, 64, 256, 1), ("cat.0 / up.0.0", 128, 64, 256, 1), ("skip.1 / up.1.1", 128, 128, 128, 1),
          ("cat.1 / up.1.0", 256, 128, 128, 1), ("skip.2 / up.2.1", 256, 256, 64, 1), ("cat.2 / up.2.0", 512, 256, 64, 1),
          ("skip.3 / up.3.*", 512, 512, 32, 1), ("cat.3", 1024, 512, 32, 1), ("style.1", 64, 128, 128, 2), ("style.2", 128, 256, 64, 2)]
for name, Ci, Co, H, st in LAYERS:
    Ho = H // st
    w = torch.randn(Co, Ci, 3, 3, device=DEV) * 0.05
    p0, p1 = ops.pack_w_split(w, True, True)
    xs = ops.split_f32(torch.randn(B, Ci, H, H, device=DEV).contiguous(memory_format=torch.channels_last))
    dys = ops.split_f32(torch.randn(B, Co, Ho, Ho, device=DEV).contiguous(memory_format=torch.channels_last))
    gf = 2.0 * B * Ho * Ho * 9 * Ci * Co * 1e-9
    t_f = timeit(lambda: ops.conv_gather_bf16x3(xs, (B, Ci, H, H), p0, Co, None, 3, st))
    t_d = timeit(lambda: ops.conv_scatter_bf16x3(dys, (B, Co, Ho, Ho), p1, Ci, 3, st, H, H))
    t_w = timeit(lambda: ops.conv_wgrad_bf16x3(xs, (B, Ci, H, H), dys, (B, Co, Ho, Ho), 3, st))
    print(f"{name:24s} {Ci:4d}->{Co:4d} @{H:3d} s{st} {gf:6.1f} GF: fwd {t_f:7.0f} us {gf / t_f * 1e3:4.0f} TF | dgrad {t_d:7.0f} us {gf / t_d * 1e3:4.0f} TF"
          f" | wgrad {t_w:7.0f} us {gf / t_w * 1e3:4.0f} TF", flush=True)
    del xs, dys
