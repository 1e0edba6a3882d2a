"""CPU check of the implicit-GEMM index math (problems.h) against torch.

tests/host_emul/emul.cpp compiles the *same* accessor structs the MFMA kernel uses and runs a
naive contraction through them; here the result is compared with torch.nn.functional on CPU.
Covers: F family (Conv2d fwd / ConvTranspose2d dgrad), T family (ConvTranspose2d fwd / Conv2d
dgrad, 9/6/6/4-tap phases), W family (both weight gradients, split-K), stride 1 and 2, channel
counts that take the scalar path (1, 3) and the vector path, and the three Linear GEMM forms.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_emul", "emul.cpp")
OUT = os.path.join(HERE, "host_emul", "libvp_host_emul.so")
FP = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def emul():
    hdr = os.path.join(HERE, "..", "vae_play_amd", "csrc", "problems.h")
    if (not os.path.exists(OUT)) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        cxx = "/opt/rocm/lib/llvm/bin/clang++"
        if not os.path.exists(cxx):
            cxx = "clang++"
        subprocess.check_call([cxx, "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", OUT])
    return ctypes.CDLL(OUT)


_KEEP = []


def ptr(t):
    """Raw pointer; the tensor is kept alive until the end of the test module."""
    if t is None:
        return None
    assert t.is_contiguous() and t.dtype == torch.float32
    _KEEP.append(t)
    return ctypes.cast(t.data_ptr(), FP)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def close(a, b, tol=2e-5):
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item() / scale
    assert err < tol, f"rel err {err}"


CASES = [  # (B, Hs, Cb, Cs, stride)
    (2, 4, 4, 8, 2), (2, 8, 3, 8, 2), (1, 8, 1, 64, 2), (2, 4, 64, 32, 2), (3, 5, 8, 4, 2),
    (2, 8, 8, 3, 1), (2, 6, 4, 4, 1), (1, 8, 64, 3, 1),
]


@pytest.mark.parametrize("B,Hs,Cb,Cs,stride", CASES)
def test_gather_is_conv2d(emul, B, Hs, Cb, Cs, stride):
    g = torch.Generator().manual_seed(1)
    Hb = Hs * stride
    big = torch.randn(B, Cb, Hb, Hb, generator=g)
    w = torch.randn(Cs, Cb, 5, 5, generator=g)
    bias = torch.randn(Cs, generator=g)
    ref = F.conv2d(big, w, bias, stride=stride, padding=2)
    wp0 = w.permute(0, 2, 3, 1).contiguous()
    out = torch.empty(B, Hs, Hs, Cs)
    emul.emul_conv5_gather(ptr(nhwc(big)), ptr(wp0), ptr(bias), ptr(out), B, Hs, Hs, Cb, Cs, stride, 0)
    close(out, nhwc(ref))
    # sigmoid epilogue
    emul.emul_conv5_gather(ptr(nhwc(big)), ptr(wp0), ptr(bias), ptr(out), B, Hs, Hs, Cb, Cs, stride, 4)
    close(out, nhwc(torch.sigmoid(ref)))


@pytest.mark.parametrize("B,Hs,Cb,Cs,stride", CASES)
def test_scatter_is_conv_transpose2d(emul, B, Hs, Cb, Cs, stride):
    g = torch.Generator().manual_seed(2)
    small = torch.randn(B, Cs, Hs, Hs, generator=g)
    w = torch.randn(Cs, Cb, 5, 5, generator=g)  # ConvTranspose2d layout (Cin=small, Cout=big)
    ref = F.conv_transpose2d(small, w, None, stride=stride, padding=2, output_padding=stride - 1)
    assert ref.shape[-1] == Hs * stride
    wp1 = w.permute(1, 2, 3, 0).contiguous()
    out = torch.full((B, Hs * stride, Hs * stride, Cb), float("nan"))
    emul.emul_conv5_scatter(ptr(nhwc(small)), ptr(wp1), ptr(out), B, Hs, Hs, Cs, Cb, stride)
    assert not torch.isnan(out).any(), "some output pixels never written"
    close(out, nhwc(ref))


@pytest.mark.parametrize("B,Hs,Cb,Cs,stride", CASES)
@pytest.mark.parametrize("split", [0, 1, 3])
def test_wgrad_is_autograd(emul, B, Hs, Cb, Cs, stride, split):
    g = torch.Generator().manual_seed(3)
    Hb = Hs * stride
    big = torch.randn(B, Cb, Hb, Hb, generator=g)
    w = torch.randn(Cs, Cb, 5, 5, generator=g, requires_grad=True)
    dsmall = torch.randn(B, Cs, Hs, Hs, generator=g)
    F.conv2d(big, w, None, stride=stride, padding=2).backward(dsmall)
    dw = torch.empty(Cs, Cb, 5, 5)
    emul.emul_conv5_wgrad(ptr(nhwc(big)), ptr(nhwc(dsmall)), ptr(dw), B, Hs, Hs, Cb, Cs, stride, split)
    close(dw, w.grad)
    # the same kernel is the ConvTranspose2d weight gradient with big = dy, small = x
    wt = torch.randn(Cs, Cb, 5, 5, generator=g, requires_grad=True)
    x = torch.randn(B, Cs, Hs, Hs, generator=g)
    dy = torch.randn(B, Cb, Hb, Hb, generator=g)
    F.conv_transpose2d(x, wt, None, stride=stride, padding=2, output_padding=stride - 1).backward(dy)
    emul.emul_conv5_wgrad(ptr(nhwc(dy)), ptr(nhwc(x)), ptr(dw), B, Hs, Hs, Cb, Cs, stride, split)
    close(dw, wt.grad)


def test_conv_dgrad_and_convT_dgrad(emul):
    g = torch.Generator().manual_seed(4)
    B, Hs, Cb, Cs = 2, 4, 8, 12
    # Conv2d dgrad = T family with P1 packing of the Conv2d weight
    x = torch.randn(B, Cb, 2 * Hs, 2 * Hs, generator=g, requires_grad=True)
    w = torch.randn(Cs, Cb, 5, 5, generator=g)
    dy = torch.randn(B, Cs, Hs, Hs, generator=g)
    F.conv2d(x, w, None, stride=2, padding=2).backward(dy)
    out = torch.empty(B, 2 * Hs, 2 * Hs, Cb)
    emul.emul_conv5_scatter(ptr(nhwc(dy)), ptr(w.permute(1, 2, 3, 0).contiguous()), ptr(out), B, Hs, Hs, Cs, Cb, 2)
    close(out, nhwc(x.grad))
    # ConvTranspose2d dgrad = F family with P0 packing of the ConvTranspose2d weight
    xs = torch.randn(B, Cs, Hs, Hs, generator=g, requires_grad=True)
    wt = torch.randn(Cs, Cb, 5, 5, generator=g)
    dbig = torch.randn(B, Cb, 2 * Hs, 2 * Hs, generator=g)
    F.conv_transpose2d(xs, wt, None, stride=2, padding=2, output_padding=1).backward(dbig)
    out2 = torch.empty(B, Hs, Hs, Cs)
    emul.emul_conv5_gather(ptr(nhwc(dbig)), ptr(wt.permute(0, 2, 3, 1).contiguous()), None, ptr(out2), B, Hs, Hs, Cb, Cs, 2, 0)
    close(out2, nhwc(xs.grad))


@pytest.mark.parametrize("M,N,K", [(4, 16, 64), (32, 1024, 4096), (5, 7, 9), (32, 128, 1024), (130, 70, 33)])
@pytest.mark.parametrize("split", [0, 1, 4])
def test_linear_gemm_forms(emul, M, N, K, split):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    y = torch.empty(M, N)
    emul.emul_gemm(ptr(x), K, 1, ptr(W), K, 1, ptr(y), N, ptr(b), M, N, K, 0, split)
    close(y, F.linear(x, W, b), 1e-5)
    dy = torch.randn(M, N, generator=g)
    dx = torch.empty(M, K)   # dx[m][k] = sum_n dy[m][n] W[n][k]  -> contraction over n: "K" = N
    emul.emul_gemm(ptr(dy), N, 1, ptr(W), 1, K, ptr(dx), K, None, M, K, N, 1, split)
    close(dx, dy @ W, 1e-5)
    dW = torch.empty(N, K)   # dW[n][k] = sum_m dy[m][n] x[m][k]  -> contraction over m
    emul.emul_gemm(ptr(dy), 1, N, ptr(x), 1, K, ptr(dW), K, None, N, K, M, 2, split)
    close(dW, dy.t() @ x, 1e-5)


# ---- k x k generalisation (models/blocks.py Conv2d: kernel 1/3/5, padding (k-1)//2, stride 1/2, odd sizes) ----
KCASES = [  # (B, Hb, Cb, Cs, ks, stride)
    (2, 8, 4, 8, 3, 1), (2, 8, 4, 8, 3, 2), (1, 9, 5, 6, 3, 2), (2, 7, 3, 4, 3, 1), (2, 8, 8, 4, 1, 1),
    (2, 8, 6, 4, 1, 2), (1, 11, 4, 4, 5, 2), (2, 10, 34, 8, 3, 1), (1, 6, 8, 1, 3, 1),
]


@pytest.mark.parametrize("B,Hb,Cb,Cs,ks,stride", KCASES)
def test_kxk_families_match_torch(emul, B, Hb, Cb, Cs, ks, stride):
    g = torch.Generator().manual_seed(ks * 100 + Hb + Cb)
    pad = (ks - 1) // 2
    Hs = (Hb + 2 * pad - ks) // stride + 1
    big = torch.randn(B, Cb, Hb, Hb, generator=g)
    w = torch.randn(Cs, Cb, ks, ks, generator=g, requires_grad=True)
    bias = torch.randn(Cs, generator=g)
    bigr = big.clone().requires_grad_(True)
    y = F.conv2d(bigr, w, bias, stride=stride, padding=pad)
    assert y.shape[-1] == Hs
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    wp0 = w.detach().permute(0, 2, 3, 1).contiguous()
    wp1 = w.detach().permute(1, 2, 3, 0).contiguous()
    out = torch.empty(B, Hs, Hs, Cs)
    emul.emul_conv_gather(ptr(nhwc(big)), ptr(wp0), ptr(bias), ptr(out), B, Hs, Hs, Hb, Hb, Cb, Cs, ks, stride, 0)
    close(out, nhwc(y.detach()))
    dx = torch.full((B, Hb, Hb, Cb), float("nan"))
    emul.emul_conv_scatter(ptr(nhwc(gy)), ptr(wp1), ptr(dx), B, Hs, Hs, Hb, Hb, Cs, Cb, ks, stride)
    assert not torch.isnan(dx).any(), "some input-gradient pixels never written"
    close(dx, nhwc(bigr.grad))
    dw = torch.empty(Cs, Cb, ks, ks)
    emul.emul_conv_wgrad(ptr(nhwc(big)), ptr(nhwc(gy)), ptr(dw), B, Hs, Hs, Hb, Hb, Cb, Cs, ks, stride, 0)
    close(dw, w.grad)
