"""Per-entry-point roofline of an autograd-front-end step (bench tools of the widened rows: tools/bench_be_heads.py, bench_be_gan.py,
bench_font.py).  ``trace(fn, n)`` runs ``fn`` n times with every C-ABI call bracketed by a HIP event pair on its launch stream
(vae_play_amd._lib.TRACE) and returns, per entry point, launches, device time and the ALGORITHMIC work of its launches -- FLOPs for
the contractions (2 MACs), bytes for the bandwidth-bound kernels (every operand read once, every result written once) -- from the
call's own shape arguments.  ``roofline(rows)`` picks the family with the most device time and prices it against the roof that
bounds it: 8 TB/s of HBM, 2.5 PFLOP/s of dense bf16 MFMA (split-bf16 convolutions: 3 MFMAs per product, so the algorithmic
fraction cannot exceed 1/3) or 157.3 TFLOP/s of fp32 MFMA (MI355X_MICROARCH.md).  Event pairs serialise nothing but add a
barrier packet each: the traced iterations are slower than the timed ones; shares, not totals, are what the table is for."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK_HBM, PEAK_BF16, PEAK_F32 = 8000.0, 2500.0, 157.3      # GB/s, TFLOP/s, TFLOP/s


def _conv(a, i):      # (B, Hs, Ws, Hb, Wb, Cbig, Csmall, ks, stride) starting at index i
    B, Hs, Ws, Hb, Wb, Cb, Cs, ks, st = a[i:i + 9]
    return 2.0 * B * Hs * Ws * ks * ks * Cb * Cs, 4.0 * (B * Hb * Wb * Cb + B * Hs * Ws * Cs + ks * ks * Cb * Cs)


def _conv5(a, i):     # (B, Hs, Ws, Cbig, Csmall, stride)
    B, Hs, Ws, Cb, Cs, st = a[i:i + 6]
    return 2.0 * B * Hs * Ws * 25 * Cb * Cs, 4.0 * (B * Hs * st * Ws * st * Cb + B * Hs * Ws * Cs + 25 * Cb * Cs)


def work(name, a):
    """(flops, bytes, bound) of one call; bound in {"mfma16", "mfma32", "hbm", None}."""
    n = name
    if n in ("vp_conv_gather_bf16x3", "vp_conv_gather_f32"):
        f, b = _conv(a, 4)
        return f, b, "mfma16" if n.endswith("bf16x3") else "mfma32"
    if n in ("vp_conv_scatter_bf16x3", "vp_conv_scatter_f32"):
        B, Hs, Ws, Hb, Wb, Cs, Cb, ks, st = a[3:12]
        return 2.0 * B * Hs * Ws * ks * ks * Cb * Cs, 4.0 * (B * Hb * Wb * Cb + B * Hs * Ws * Cs + ks * ks * Cb * Cs), "mfma16" if n.endswith("bf16x3") else "mfma32"
    if n in ("vp_conv_wgrad_bf16x3", "vp_conv_wgrad_f32"):
        f, b = _conv(a, 3)
        return f, b, "mfma16" if n.endswith("bf16x3") else "mfma32"
    if n in ("vp_conv5_gather_bf16x3", "vp_conv5_gather_f32"):
        f, b = _conv5(a, 4)
        return f, b, "mfma16" if n.endswith("bf16x3") else "mfma32"
    if n in ("vp_conv5_scatter_bf16x3", "vp_conv5_scatter_f32"):
        B, Hs, Ws, Cs, Cb, st = a[3:9]
        return 2.0 * B * Hs * Ws * 25 * Cb * Cs, 4.0 * (B * Hs * st * Ws * st * Cb + B * Hs * Ws * Cs + 25 * Cb * Cs), "mfma16" if n.endswith("bf16x3") else "mfma32"
    if n in ("vp_conv5_wgrad_bf16x3", "vp_conv5_wgrad_f32"):
        f, b = _conv5(a, 3)
        return f, b, "mfma16" if n.endswith("bf16x3") else "mfma32"
    if n == "vp_gemm_f32":
        M, N, K = a[9], a[10], a[11]
        f, b = 2.0 * M * N * K, 4.0 * (M * K + N * K + M * N)
        return f, b, "hbm" if min(M, N, K) <= 64 else "mfma32"       # a batch-sized dimension: streaming the big matrix bounds it
    if n == "vp_conv3_small_fwd_f32":
        B, H, W, Ci, Co = a[4:9]
        return 2.0 * B * H * W * 9 * Ci * Co, 4.0 * B * H * W * (Ci + Co), "hbm"
    if n == "vp_conv3_small_dgrad_f32":
        B, H, W, Ci, Co = a[3:8]
        return 2.0 * B * H * W * 9 * Ci * Co, 4.0 * B * H * W * (Ci + Co), "hbm"
    if n == "vp_conv3_small_wgrad_f32":
        B, H, W, Ci, Co = a[3:8]
        return 2.0 * B * H * W * 9 * Ci * Co, 4.0 * B * H * W * (Ci + Co), "hbm"
    if n in ("vp_adam_f32",):
        return 0.0, 28.0 * a[4], "hbm"
    if n in ("vp_rmsprop_f32",):
        return 0.0, 20.0 * a[3], "hbm"
    if n.startswith("vp_bn_act_fwd") or n.startswith("vp_instnorm_act_fwd"):
        ints = [v for v in a if isinstance(v, int) and not isinstance(v, bool)]
        R, C = (ints[0] * ints[1], ints[2]) if n.startswith("vp_instnorm") else (ints[0], ints[1])
        return 0.0, 4.0 * R * C * (3 if "split" in n else 2) + (4.0 * R * C * 2 if n.startswith("vp_instnorm") else 0.0), "hbm"
    if n.startswith("vp_bn_act_bwd") or n.startswith("vp_instnorm_act_bwd"):
        ints = [v for v in a if isinstance(v, int) and not isinstance(v, bool)]
        R, C = (ints[0] * ints[1], ints[2]) if n.startswith("vp_instnorm") else (ints[0], ints[1])
        return 0.0, 4.0 * R * C * 5, "hbm"          # statistics pass reads x, dy; apply pass reads x, dy, writes dx
    if n == "vp_bn_stats_f32":
        return 0.0, 4.0 * a[1] * a[2], "hbm"
    if n in ("vp_split_f32",):
        return 0.0, 8.0 * a[2], "hbm"
    if n.startswith("vp_upsample2x_bilinear"):
        B, H, W, C = a[2:6]
        return 0.0, 4.0 * B * H * W * C * 5, "hbm"
    if n in ("vp_act_fwd_f32",):
        return 0.0, 8.0 * a[2], "hbm"
    if n in ("vp_act_bwd_from_y_f32",):
        return 0.0, 12.0 * a[3], "hbm"
    return 0.0, 0.0, None


def trace(fn, n=2):
    import torch
    from vae_play_amd import _lib
    fn()
    torch.cuda.synchronize()
    _lib.TRACE = {"events": []}
    try:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ev = _lib.TRACE["events"]
    finally:
        _lib.TRACE = None
    rows = {}
    for name, args, e0, e1 in ev:
        vals = [None if hasattr(a, "value") or a is None else a for a in args]      # pointers -> None: only shape arguments are read
        f, b, bound = work(name, vals)
        r = rows.setdefault(name, {"name": name, "launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "bound": bound})
        r["launches"] += 1
        r["ms"] += e0.elapsed_time(e1)
        r["flops"] += f
        r["bytes"] += b
    for r in rows.values():
        r["launches"] /= n; r["ms"] /= n; r["flops"] /= n; r["bytes"] /= n
    return sorted(rows.values(), key=lambda r: -r["ms"])


def roofline(rows, top=6):
    tot = sum(r["ms"] for r in rows)
    fams = []
    for r in rows[:top]:
        d = {"entry": r["name"], "launches": round(r["launches"], 1), "ms": round(r["ms"], 4), "share": round(r["ms"] / tot, 3), "bound": r["bound"]}
        if r["bound"] in ("mfma16", "mfma32") and r["ms"] > 0:
            peak = PEAK_BF16 if r["bound"] == "mfma16" else PEAK_F32
            ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
            d.update(achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4))
        elif r["bound"] == "hbm" and r["ms"] > 0:
            ach = r["bytes"] / (r["ms"] * 1e-3) / 1e9
            d.update(achieved=round(ach, 1), peak=PEAK_HBM, unit="GB/s", frac=round(ach / PEAK_HBM, 4))
        fams.append(d)
    dom = next((d for d in fams if "achieved" in d), None)
    if dom is None:
        return {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None, "families": fams}
    return {"bound": "hbm" if dom["bound"] == "hbm" else "mfma", "kernel": dom["entry"], "achieved": dom["achieved"], "peak": dom["peak"],
            "unit": dom["unit"], "frac": dom["frac"], "traffic": None, "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
            "launches": dom["launches"], "device_ms_traced": round(tot, 3), "families": fams,
            "note": "dominant C-ABI entry point of the iteration by HIP-event time (events around every call of 2 traced iterations); "
                    "algorithmic FLOPs / bytes from the call's shape arguments"}
