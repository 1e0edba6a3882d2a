"""Tile choice for split-bf16 layers with <= 32 output columns (igemm16.h choose_tile16): times the default (256x32 / 128x32 tiles,
32-deep K-tiles) against 64-column tiles forced with VP_TILE_OVERRIDE on scatter / gather shapes of the VAE-GAN, BE and font rows.
usage: python tools/microbench_narrow_n.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VP_ENV_DYNAMIC", "1")      # the library then re-reads its per-launch knobs on every launch (csrc/env.h)
from vae_play_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


# (family, ks, stride, B, Hs (small side), Csmall, Cbig)
CASES = [("scatter", 5, 2, 48, 64, 64, 32), ("scatter", 5, 2, 16, 64, 64, 32), ("scatter", 3, 1, 16, 128, 64, 32), ("scatter", 3, 1, 16, 256, 32, 32),
         ("scatter", 3, 1, 16, 256, 32, 8), ("gather", 3, 1, 16, 128, 32, 64), ("gather", 3, 1, 16, 256, 32, 32), ("gather", 3, 1, 16, 256, 8, 40),
         ("gather", 5, 2, 32, 64, 32, 64), ("gather", 3, 2, 16, 64, 32, 128)]
for fam, ks, st, B, Hs, Cs, Cb in CASES:
    Hb = Hs * st
    w = torch.randn(Cs, Cb, ks, ks, device=DEV) * 0.05
    p0, p1 = ops.pack_w_split(w, True, True)
    if fam == "scatter":
        xs = ops.split_f32(torch.randn(B, Cs, Hs, Hs, device=DEV).contiguous(memory_format=torch.channels_last))
        fn = lambda: ops.conv_scatter_bf16x3(xs, (B, Cs, Hs, Hs), p1, Cb, ks, st, Hb, Hb)
        M, N, gz = B * Hs * Hs, Cb, st * st
    else:
        xs = ops.split_f32(torch.randn(B, Cb, Hb, Hb, device=DEV).contiguous(memory_format=torch.channels_last))
        fn = lambda: ops.conv_gather_bf16x3(xs, (B, Cb, Hb, Hb), p0, Cs, None, ks, st)
        M, N, gz = B * Hs * Hs, Cs, 1
    gf = 2.0 * B * Hs * Hs * ks * ks * Cs * Cb * 1e-9
    line = f"{fam} k{ks} s{st} B{B} Hs{Hs} Cs{Cs} Cb{Cb} (M={M} N={N} gz={gz}, {gf:.1f} GF):"
    for ov in ("", "128x64", "64x64"):
        if ov:
            os.environ["VP_TILE_OVERRIDE"] = f"{M}x{N}x{gz}:{ov}"
        else:
            os.environ.pop("VP_TILE_OVERRIDE", None)
        us = timeit(fn)
        line += f"  {ov or 'default'} {us:7.1f} us ({gf / us * 1e3:5.0f} TF)"
    os.environ.pop("VP_TILE_OVERRIDE", None)
    print(line, flush=True)
