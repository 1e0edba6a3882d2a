"""Bounded experiment on product count: would Winograd F(2x2, 3x3) pay on the (even, even) phase of dec2's transposed
convolution?  That phase is a 3 x 3 stride-1 convolution of the small image (32 images x 32 x 32 pixels, 256 -> 128 channels): 9 products per
output against 4 for F(2x2, 3x3) (16 element-wise products per 2 x 2 output block).

The experiment prices the UNFUSED form out of kernels the library already has, which is an upper bound on the GEMM part of any fused form
(full 128 x 128 tiles, both operands reused as in the direct kernel):
  * input transform  V[xi][tile][c] = (B^T d B)[xi]: fp32, then split planes -- a bandwidth pass that writes 4x the activation;
  * 16 GEMMs [tiles x 256] x [256 x 128] on the split-bf16 1 x 1 gather kernel (timed as ONE launch over 16 x tiles rows: same work, same
    operand traffic; the 16 launches with their own weights are what the accuracy leg runs);
  * output transform  Y = A^T M A: reads 16 fp32 planes, writes the 2 x 2 blocks.
and compares it with the direct kernels on the same shape (the generic 3 x 3 gather, and 9/25 of the layer's phase-decomposed launch).
Accuracy leg: fp32 transforms, split after the input transform, against an fp64 direct convolution.

usage (GPU box): python tools/winograd_probe.py [--batch 32] [--reps 20]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def input_transform(x):
    """x: [B, C, H, W] fp32 (H, W even) -> V [16, B * H/2 * W/2, C]: 4 x 4 patches at stride 2 of the zero-padded image."""
    B, C, H, W = x.shape
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [B, C, H/2, W/2, 4, 4]
    bt = BT.to(x)
    v = torch.einsum("ij,bchwjk,lk->bhwilc", bt, d, bt)          # [B, H/2, W/2, 4, 4, C]
    return v.reshape(B * (H // 2) * (W // 2), 16, C).permute(1, 0, 2).contiguous()


def weight_transform(w):
    """w: [N, C, 3, 3] -> U [16, N, C] (G g G^T)."""
    g = G.to(w)
    u = torch.einsum("ij,ncjk,lk->ilnc", g, w, g)
    return u.reshape(16, w.shape[0], w.shape[1]).contiguous()


def output_transform(m, B, H, W):
    """m: [16, tiles, N] -> y [B, N, H, W]."""
    N = m.shape[2]
    at = AT.to(m)
    mm = m.reshape(4, 4, B, H // 2, W // 2, N)
    y = torch.einsum("ij,jkbhwn,lk->bnhiwl", at, mm, at)         # [B, N, H/2, 2, W/2, 2]
    return y.reshape(B, N, H, W)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--hw", type=int, default=32)
    a = ap.parse_args()
    from vae_play_amd import ops
    dev = "cuda"
    B, C, N, H = a.batch, a.cin, a.cout, a.hw
    g = torch.Generator().manual_seed(7)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    res = {"shape": f"{B} x {H} x {H}, {C} -> {N} channels, 3 x 3 stride 1 (the (even, even) phase of dec2.fwd)"}

    # ---- accuracy: 4 images, fp64 direct convolution as the ground truth ------------------------------------------------------------
    Ba = 4
    x = torch.relu(torch.randn(Ba, C, H, H, generator=g))       # post-ReLU activations, as in the step
    w = torch.randn(N, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    rms = ref.pow(2).mean().sqrt().item()
    xs, wd = cl(x), w.to(dev)
    # direct split-bf16 3 x 3
    p0, _ = ops.pack_w_split(wd, True, False)
    y_dir = ops.conv_gather_bf16x3(ops.split_f32(xs), xs.shape, p0, N, None, 3, 1)
    e_dir = (y_dir.double().cpu() - ref).abs().max().item() / rms
    # Winograd: fp32 transforms on the device, 16 split-bf16 GEMMs
    V = input_transform(xs.contiguous())                         # [16, tiles, C] fp32
    U = weight_transform(wd)                                     # [16, N, C]
    tiles = V.shape[1]
    M = torch.empty(16, tiles, N, device=dev)
    for xi in range(16):
        v_img = V[xi].reshape(1, tiles, 1, C).permute(0, 3, 1, 2)   # NHWC storage [1, tiles, 1, C]
        pu, _ = ops.pack_w_split(U[xi].reshape(N, C, 1, 1).contiguous(), True, False)
        out = ops.conv_gather_bf16x3(ops.split_f32(v_img), (1, C, tiles, 1), pu, N, None, 1, 1)
        M[xi] = out.permute(0, 2, 3, 1).reshape(tiles, N)
    y_win = output_transform(M, Ba, H, H)
    e_win = (y_win.double().cpu() - ref).abs().max().item() / rms
    # the same Winograd algebra in fp64 end to end: separates the transform's conditioning from the split arithmetic
    y64 = output_transform(torch.einsum("xtc,xnc->xtn", input_transform(x.double()), weight_transform(w.double())), Ba, H, H)
    e_alg = (y64 - ref).abs().max().item() / rms
    # ... and in plain fp32 (what an exact-f32 Winograd would give)
    y32 = output_transform(torch.einsum("xtc,xnc->xtn", input_transform(x), weight_transform(w)), Ba, H, H)
    e_f32 = (y32.double() - ref).abs().max().item() / rms
    res["max_err_over_rms"] = {"direct_bf16x3": e_dir, "winograd_bf16x3_split_after_input_transform": e_win,
                               "winograd_fp32_everything": e_f32, "winograd_fp64_everything": e_alg}

    # ---- timing at the benchmark shard ------------------------------------------------------------------------------------------
    x = cl(torch.relu(torch.randn(B, C, H, H, generator=g)))
    xsplit = ops.split_f32(x)
    t_direct3 = timed(lambda: ops.conv_gather_bf16x3(xsplit, x.shape, p0, N, None, 3, 1), a.reps)
    # the layer as the step runs it: 5 x 5 stride-2 transposed convolution, all four phases in one launch; (even, even) = 9 of 25 taps
    w5 = (torch.randn(C, N, 5, 5, generator=g) * 0.02).to(dev)
    _, p1 = ops.pack_w5_split(w5, False, True)
    t_layer = timed(lambda: ops.conv5_scatter_bf16x3(xsplit, x.shape, p1, N, 2), a.reps)
    tiles = B * (H // 2) * (H // 2)
    rows = 16 * tiles
    # 16 GEMMs [tiles x C] x [C x N] as one 1 x 1 launch over 16 x tiles rows
    vbig = torch.randn(1, C, rows, 1, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    vsplit = ops.split_f32(vbig)
    pu, _ = ops.pack_w_split((torch.randn(N, C, 1, 1, generator=g) * 0.05).to(dev), True, False)
    t_gemm = timed(lambda: ops.conv_gather_bf16x3(vsplit, (1, C, rows, 1), pu, N, None, 1, 1), a.reps)
    # bandwidth passes, priced with the library's own streaming kernels on buffers of the transforms' sizes:
    #   input transform: read the activation (B*H*H*C fp32) once, write 16 x tiles x C split planes = 4x the activation
    vf = torch.empty(rows * C, device=dev)
    t_split4 = timed(lambda: ops.split_f32(vf), a.reps)        # reads 4x + writes 4x: an upper bound (the transform reads 1x)
    xin = torch.empty(B * H * H * C, device=dev)
    t_split1 = timed(lambda: ops.split_f32(xin), a.reps)       # reads 1x + writes 1x
    t_in = t_split1 + (t_split4 - t_split1) * 0.5              # 1x read + 4x write, linear in bytes
    #   output transform: read 16 x tiles x N fp32, write 4 x tiles x N
    mf, mg = torch.empty(rows * N, device=dev), torch.empty(rows * N, device=dev)
    t_copy = timed(lambda: mg.copy_(mf), a.reps)               # reads + writes 16 x tiles x N: upper bound (the transform writes a quarter)
    t_out = t_copy * (1.0 + 0.25) / 2.0
    t_ee = t_layer * 9.0 / 25.0
    t_win = t_in + t_gemm + t_out
    res["us"] = {"direct_3x3_gather": round(t_direct3, 1), "layer_all_phases": round(t_layer, 1), "even_even_share_9_of_25": round(t_ee, 1),
                 "winograd_16_gemms": round(t_gemm, 1), "input_transform_pass": round(t_in, 1), "output_transform_pass": round(t_out, 1),
                 "winograd_unfused_total": round(t_win, 1)}
    best_direct = min(t_direct3, t_ee)
    res["speedup"] = {"gemm_part_alone_vs_direct": round(best_direct / t_gemm, 3), "unfused_total_vs_direct": round(best_direct / t_win, 3)}
    res["gflop"] = {"direct": 2.0 * B * H * H * N * 9 * C / 1e9, "winograd_gemms": 2.0 * rows * N * C / 1e9}
    res["criterion"] = "keep if >= 1.3x faster including transforms at <= 5e-5 output error (VERDICT r2 item 9)"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
