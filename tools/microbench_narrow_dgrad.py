"""Input gradient of a first conv (1 input channel, 32 outputs, 5x5 stride 1) at the VAE-GAN discriminator's size (48 images of
128 x 128): exact-f32 scatter vs the split-bf16 scatter with the input channel zero-padded to 8.
usage: python tools/microbench_narrow_dgrad.py"""
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


B, H, Cs, Cb = 48, 128, 32, 1
dy = torch.randn(B, Cs, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(Cs, Cb, 5, 5, device="cuda") * 0.1
_, p1 = ops.pack_w5(w, False, True)
t_f32 = timed(lambda: ops.conv5_scatter(dy, p1, 1))
ref = ops.conv5_scatter(dy, p1, 1)
wpad = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 8 - Cb))
_, p1s = ops.pack_w5_split(wpad, False, True)
dys = ops.split_f32(dy)
t_split = timed(lambda: ops.split_f32(dy))
t_sc = timed(lambda: ops.conv5_scatter_bf16x3(dys, dy.shape, p1s, 8, 1))
got = ops.conv5_scatter_bf16x3(dys, dy.shape, p1s, 8, 1)[:, :Cb]
t_slice = timed(lambda: ops.conv5_scatter_bf16x3(dys, dy.shape, p1s, 8, 1)[:, :Cb].contiguous(memory_format=torch.channels_last))
print(f"f32 scatter {t_f32:.1f} us | split(dy) {t_split:.1f} us + bf16x3 scatter to 8 channels {t_sc:.1f} us (+ slice copy: {t_slice:.1f} us)"
      f" | rel diff {((got - ref).norm() / ref.norm()).item():.1e}")
