"""Tensor-level wrappers over the C ABI (one function per entry point of include/vaeplay_hip.h).

Conventions: activations are 4-D tensors with *logical* NCHW shape stored channels_last
(= NHWC in memory, what the kernels index), or 2-D [R, C] matrices; everything is fp32 on
the current HIP device and stream.  These wrappers only marshal pointers -- all arithmetic
happens in libvaeplay_hip.so.
"""
from __future__ import annotations

from ctypes import c_void_p
from typing import Optional, Tuple

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
ACT_CODES = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "lrelu": ACT_LRELU, "tanh": ACT_TANH,
             "sigmoid": ACT_SIGMOID}


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.VaePlayHipError("vae_play_amd ops need tensors on the HIP device (no CPU path exists)")
    if t.dtype != torch.float32:
        raise _lib.VaePlayHipError(f"fp32 tensor expected, got {t.dtype}")
    return c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


CAPTURE_OK = [False]     # set by the fused plans (engine.FusedVAEStep.capture) around their own hipGraph capture


def _stream():
    """The current HIP stream of the current device as a raw pointer.  torch.cuda.current_stream() builds a Stream object per call
    (13 % of the host time of a launch-bound autograd step); torch's raw-stream accessor is one C call.

    hipGraph capture of the AUTOGRAD front end is refused here, before anything is launched into the capture: autograd's
    AccumulateGrad nodes live on the stream their parameters were first used on (normally the default stream), the engine
    synchronises that stream with the capturing one -- illegal inside a capture -- and every launch after that returns
    "capture invalidated" on an autograd worker thread, which ends the process (round 2: gpurun_out/be_graph.err, a core dump
    right after torch's AccumulateGrad-stream warning).  The pre-planned steps (engine.FusedVAEStep / engine_gan.FusedVAEGANStep)
    are the graph-capturable front ends.  The check costs nothing on the default stream (raw handle 0), where no capture can run."""
    if _raw_stream is not None:
        h = _raw_stream(torch.cuda.current_device())
        if h and not CAPTURE_OK[0] and torch.cuda.is_current_stream_capturing():
            raise _lib.VaePlayHipError(
                "hipGraph capture of the autograd front end is not supported (autograd's AccumulateGrad nodes synchronise with "
                "the stream their parameters were first used on, which invalidates the capture and aborts the process); "
                "capture engine.FusedVAEStep / engine_gan.FusedVAEGANStep instead, or run this step eagerly")
        return c_void_p(h)
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(max(4, (nbytes + 3) // 4), dtype=torch.float32, device=like.device)


def channels_last(x: torch.Tensor) -> torch.Tensor:
    """Logical NCHW tensor whose memory is NHWC: no copy when it already is, the HIP transpose
    for a standard-contiguous NCHW tensor, torch's strided copy only for exotic strides."""
    if x.is_contiguous(memory_format=torch.channels_last):
        return x
    if x.is_contiguous() and x.is_cuda and x.dtype == torch.float32:
        return nchw_to_nhwc(x)
    return x.contiguous(memory_format=torch.channels_last)


def empty_cl(B: int, C: int, H: int, W: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty((B, C, H, W), dtype=torch.float32, device=like.device, memory_format=torch.channels_last)


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Same shape and same memory order (strides of size-1 dims are irrelevant)."""
    if a.shape != b.shape:
        return False
    return all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


def _is_nhwc(x: torch.Tensor) -> bool:
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)


# ---- layout --------------------------------------------------------------------------------
def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """Copy a standard-contiguous NCHW tensor into a channels_last one with the HIP transpose."""
    B, C, H, W = x.shape
    x = x.contiguous()
    out = empty_cl(B, C, H, W, x)
    _lib.call("vp_nchw_to_nhwc_f32", _p(x), _p(out), B, C, H, W, _stream())
    return out


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    B, C, H, W = x.shape
    assert _is_nhwc(x)
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _lib.call("vp_nhwc_to_nchw_f32", _p(x), _p(out), B, C, H, W, _stream())
    return out


def pack_w5(w_ref: torch.Tensor, want_p0: bool, want_p1: bool) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """w_ref [Cs][Cb][5][5] -> p0 [Cs][25][Cb], p1 [Cb][25][Cs]."""
    Cs, Cb = w_ref.shape[0], w_ref.shape[1]
    assert w_ref.shape[2:] == (5, 5)
    w_ref = w_ref.contiguous()
    p0 = torch.empty((Cs, 25, Cb), dtype=torch.float32, device=w_ref.device) if want_p0 else None
    p1 = torch.empty((Cb, 25, Cs), dtype=torch.float32, device=w_ref.device) if want_p1 else None
    _lib.call("vp_pack_w5_f32", _p(w_ref), _p(p0), _p(p1), Cs, Cb, _stream())
    return p0, p1


# ---- 5x5 convolution families ----------------------------------------------------------------
def conv5_gather(big: torch.Tensor, w_p0: torch.Tensor, bias: Optional[torch.Tensor], stride: int,
                 act: int = ACT_NONE) -> torch.Tensor:
    """small = act(bias + conv5x5_pad2_stride(big)); big is (B,Cb,Hb,Wb) channels_last."""
    assert _is_nhwc(big)
    B, Cb, Hb, Wb = big.shape
    Cs = w_p0.shape[0]
    assert w_p0.shape == (Cs, 25, Cb) and Hb % stride == 0 and Wb % stride == 0
    Hs, Ws = Hb // stride, Wb // stride
    out = empty_cl(B, Cs, Hs, Ws, big)
    _lib.call("vp_conv5_gather_f32", _p(big), _p(w_p0), _p(bias), _p(out), B, Hs, Ws, Cb, Cs, stride, act, _stream())
    return out


def conv5_scatter(small: torch.Tensor, w_p1: torch.Tensor, stride: int) -> torch.Tensor:
    """big = convT5x5_pad2_stride(small) (output_padding = stride-1); small is (B,Cs,Hs,Ws) channels_last."""
    assert _is_nhwc(small)
    B, Cs, Hs, Ws = small.shape
    Cb = w_p1.shape[0]
    assert w_p1.shape == (Cb, 25, Cs)
    out = empty_cl(B, Cb, Hs * stride, Ws * stride, small)
    _lib.call("vp_conv5_scatter_f32", _p(small), _p(w_p1), _p(out), B, Hs, Ws, Cs, Cb, stride, _stream())
    return out


def _out(out: Optional[torch.Tensor], shape, device) -> torch.Tensor:
    """``out`` (a caller-owned fp32 buffer of that shape, e.g. a parameter's slice of the optimiser's gradient arena) or a new tensor"""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    assert tuple(out.shape) == tuple(shape) and out.is_contiguous() and out.dtype == torch.float32
    return out


def conv5_wgrad(big: torch.Tensor, small: torch.Tensor, stride: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[Cs][Cb][5][5] in the reference layout."""
    assert _is_nhwc(big) and _is_nhwc(small)
    B, Cb, Hb, Wb = big.shape
    _, Cs, Hs, Ws = small.shape
    assert Hb == Hs * stride and Wb == Ws * stride and small.shape[0] == B
    nbytes = _lib.load().vp_conv5_wgrad_workspace_bytes(B, Hs, Ws, Cb, Cs, stride)
    ws = _ws(nbytes, big)
    dw = _out(out, (Cs, Cb, 5, 5), big.device)
    _lib.call("vp_conv5_wgrad_f32", _p(big), _p(small), _p(dw), B, Hs, Ws, Cb, Cs, stride, _p(ws), ws.numel() * 4, _stream())
    return dw


# ---- dense ------------------------------------------------------------------------------------
def gemm(A: torch.Tensor, sam: int, sak: int, Bm: torch.Tensor, sbn: int, sbk: int, M: int, N: int, K: int, mode: int,
         bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    nbytes = _lib.load().vp_gemm_workspace_bytes(M, N, K)
    ws = _ws(nbytes, A) if nbytes else None
    _lib.call("vp_gemm_f32", _p(A), sam, sak, _p(Bm), sbn, sbk, _p(out), out.stride(0), _p(bias), M, N, K, mode,
              _p(ws), (ws.numel() * 4 if ws is not None else 0), _stream())
    return out


def linear_fwd(x: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """y[M,N] = x[M,K] W[N,K]^T + bias."""
    x, W = x.contiguous(), W.contiguous()
    M, K = x.shape
    N = W.shape[0]
    return gemm(x, K, 1, W, K, 1, M, N, K, 0, bias)


def linear_dgrad(dy: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """dx[M,K] = dy[M,N] W[N,K]."""
    dy, W = dy.contiguous(), W.contiguous()
    M, N = dy.shape
    K = W.shape[1]
    return gemm(dy, N, 1, W, 1, K, M, K, N, 1)


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[N,K] = dy[M,N]^T x[M,K]."""
    dy, x = dy.contiguous(), x.contiguous()
    M, N = dy.shape
    K = x.shape[1]
    return gemm(dy, 1, N, x, 1, K, N, K, M, 2, out=out)


def conv3_small_wgrad_applicable(B: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    return _lib.load().vp_conv3_small_wgrad_workspace_bytes(B, H, W, Cin, Cout) > 0


def conv3_small_wgrad(x: torch.Tensor, dy: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW (Cout, Cin, 3, 3) of a 3x3 / stride 1 / padding 1 convolution with few channels; x, dy: (B, C, H, W) channels_last fp32."""
    B, Ci, H, W = x.shape
    Co = dy.shape[1]
    assert _is_nhwc(x) and _is_nhwc(dy) and tuple(dy.shape) == (B, Co, H, W)
    nbytes = _lib.load().vp_conv3_small_wgrad_workspace_bytes(B, H, W, Ci, Co)
    if nbytes == 0:
        raise _lib.VaePlayHipError("vp_conv3_small_wgrad_f32 does not take this shape")
    ws = _ws(nbytes, x)
    dw = _out(out, (Co, Ci, 3, 3), x.device)
    _lib.call("vp_conv3_small_wgrad_f32", _p(x), _p(dy), _p(dw), B, H, W, Ci, Co, _p(ws), ws.numel() * 4, _stream())
    return dw


def conv3_small_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """y = bias + conv3x3(x) (stride 1, padding 1) for few channels; x (B, Ci, H, W) channels_last, weight (Co, Ci, 3, 3)."""
    B, Ci, H, W = x.shape
    Co = weight.shape[0]
    assert _is_nhwc(x) and weight.is_contiguous()
    y = empty_cl(B, Co, H, W, x)
    _lib.call("vp_conv3_small_fwd_f32", _p(x), _p(weight), _p(bias), _p(y), B, H, W, Ci, Co, _stream())
    return y


def conv3_small_dgrad(dy: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    B, Co, H, W = dy.shape
    Ci = weight.shape[1]
    assert _is_nhwc(dy) and weight.is_contiguous()
    dx = empty_cl(B, Ci, H, W, dy)
    _lib.call("vp_conv3_small_dgrad_f32", _p(dy), _p(weight), _p(dx), B, H, W, Ci, Co, _stream())
    return dx


def wgrad_slab_reduce(slab: torch.Tensor, Cs: int, Cb: int, nt: int, variant: int = -1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """slab[nsplit][nt][Cs][Cb] (the K-split weight-gradient launches' workspace) -> dw[Cs][Cb][nt], splits summed in a fixed order."""
    nsplit = slab.numel() // (nt * Cs * Cb)
    assert slab.is_contiguous() and slab.numel() == nsplit * nt * Cs * Cb
    if out is None:
        out = torch.empty((Cs, Cb, nt), device=slab.device, dtype=torch.float32)
    _lib.call("vp_wgrad_slab_reduce_f32", _p(slab), _p(out), Cs, Cb, nsplit, nt, variant, _stream())
    return out


def colsum(x2d: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    R, C = x2d.shape
    assert x2d.is_contiguous()
    ws = _ws(_lib.load().vp_colsum_workspace_bytes(R, C), x2d)
    out = _out(out, (C,), x2d.device)
    _lib.call("vp_colsum_f32", _p(x2d), _p(out), R, C, _p(ws), ws.numel() * 4, _stream())
    return out


# ---- BatchNorm + activation over an [R][C] view ---------------------------------------------------
def _rc(x: torch.Tensor) -> Tuple[int, int]:
    if x.dim() == 2:
        assert x.is_contiguous()
        return x.shape[0], x.shape[1]
    assert _is_nhwc(x)
    B, C, H, W = x.shape
    return B * H * W, C


def bn_stats(x, eps, momentum, running_mean=None, running_var=None):
    R, C = _rc(x)
    mean = torch.empty((C,), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = _ws(_lib.load().vp_bn_workspace_bytes(R, C), x)
    _lib.call("vp_bn_stats_f32", _p(x), R, C, float(eps), float(momentum), _p(mean), _p(rstd), _p(running_mean),
              _p(running_var), _p(ws), ws.numel() * 4, _stream())
    return mean, rstd


def bn_act_fwd(x, mean, rstd, gamma, beta, act: int, slope: float = 0.0, want_split: bool = False):
    """``want_split``: also emit the bf16 hi/lo planes of y from the same pass (C % 4 == 0); returns (y, y_split)."""
    R, C = _rc(x)
    y = torch.empty_like(x)  # preserves channels_last strides
    if want_split:
        ys = empty_split(x.numel(), x)
        _lib.call("vp_bn_act_fwd_split_f32", _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), _pv(ys), R, C, act, float(slope),
                  _stream())
        return y, ys
    _lib.call("vp_bn_act_fwd_f32", _p(x), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(y), R, C, act, float(slope), _stream())
    return y


def bn_act_bwd(x, dy, mean, rstd, gamma, beta, act: int, slope: float, batch_stats: bool, need_affine_grads: bool = True,
               out_dgamma: Optional[torch.Tensor] = None, out_dbeta: Optional[torch.Tensor] = None, want_split: bool = False):
    """``want_split``: also emit the bf16 hi/lo planes of dx from the same pass (C % 4 == 0); returns (dx, dgamma, dbeta, dx_split)."""
    R, C = _rc(x)
    assert _same_layout(dy, x)
    dx = torch.empty_like(x)
    dgamma = _out(out_dgamma, (C,), x.device) if need_affine_grads else None
    dbeta = _out(out_dbeta, (C,), x.device) if need_affine_grads else None
    ws = _ws(_lib.load().vp_bn_workspace_bytes(R, C), x)
    if want_split:
        dxs = empty_split(x.numel(), x)
        _lib.call("vp_bn_act_bwd_split_f32", _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx), _pv(dxs), _p(dgamma), _p(dbeta),
                  R, C, act, float(slope), int(batch_stats), _p(ws), ws.numel() * 4, _stream())
        return dx, dgamma, dbeta, dxs
    _lib.call("vp_bn_act_bwd_f32", _p(x), _p(dy), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx), _p(dgamma), _p(dbeta),
              R, C, act, float(slope), int(batch_stats), _p(ws), ws.numel() * 4, _stream())
    return dx, dgamma, dbeta


def instnorm_act_fwd(x, eps: float, act: int, slope: float = 0.0, want_split: bool = False):
    """x: (B, C, H, W) channels_last -> (y, mean[B][C], rstd[B][C]) (+ the bf16 hi/lo planes of y with ``want_split``, C % 4 == 0)."""
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((B, C), dtype=torch.float32, device=x.device)
    rstd = torch.empty((B, C), dtype=torch.float32, device=x.device)
    ws = _ws(_lib.load().vp_instnorm_workspace_bytes(B, H * W, C), x)
    if want_split:
        ys = empty_split(x.numel(), x)
        _lib.call("vp_instnorm_act_fwd_split_f32", _p(x), _p(y), _pv(ys), _p(mean), _p(rstd), B, H * W, C, eps, act, slope, _p(ws),
                  ws.numel() * 4, _stream())
        return y, mean, rstd, ys
    _lib.call("vp_instnorm_act_fwd_f32", _p(x), _p(y), _p(mean), _p(rstd), B, H * W, C, eps, act, slope, _p(ws), ws.numel() * 4, _stream())
    return y, mean, rstd


def instnorm_act_bwd(x, dy, mean, rstd, act: int, slope: float = 0.0, want_split: bool = False):
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    ws = _ws(_lib.load().vp_instnorm_workspace_bytes(B, H * W, C), x)
    if want_split:
        dxs = empty_split(x.numel(), x)
        _lib.call("vp_instnorm_act_bwd_split_f32", _p(x), _p(dy), _p(mean), _p(rstd), _p(dx), _pv(dxs), B, H * W, C, act, slope, _p(ws),
                  ws.numel() * 4, _stream())
        return dx, dxs
    _lib.call("vp_instnorm_act_bwd_f32", _p(x), _p(dy), _p(mean), _p(rstd), _p(dx), B, H * W, C, act, slope, _p(ws), ws.numel() * 4,
              _stream())
    return dx


def act_fwd(x, act: int, slope: float = 0.0):
    y = torch.empty_like(x)
    _lib.call("vp_act_fwd_f32", _p(x), _p(y), x.numel(), act, float(slope), _stream())
    return y


def act_bwd_from_y(y, dy, act: int, slope: float = 0.0):
    assert _same_layout(y, dy)
    dx = torch.empty_like(y)
    _lib.call("vp_act_bwd_from_y_f32", _p(y), _p(dy), _p(dx), y.numel(), act, float(slope), _stream())
    return dx


# ---- latent + loss ---------------------------------------------------------------------------------
def latent_fwd(mu, logvar, eps, want_kl: bool = True):
    B, Z = mu.shape
    mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
    z = torch.empty_like(mu)
    kl = torch.empty((B,), dtype=torch.float32, device=mu.device) if want_kl else None
    _lib.call("vp_latent_fwd_f32", _p(mu), _p(logvar), _p(eps), _p(z), _p(kl), B, Z, _stream())
    return z, kl


def latent_bwd(mu, logvar, eps, dz, gkl, gkl_scalar: float = 0.0):
    B, Z = mu.shape
    dmu = torch.empty_like(mu)
    dlv = torch.empty_like(mu)
    _lib.call("vp_latent_bwd_f32", _p(mu.contiguous()), _p(logvar.contiguous()), _p(eps.contiguous()),
              _p(dz.contiguous() if dz is not None else None), _p(gkl.contiguous() if gkl is not None else None),
              float(gkl_scalar), _p(dmu), _p(dlv), B, Z, _stream())
    return dmu, dlv


def bce_sum(p, t):
    assert _same_layout(p, t)
    n = p.numel()
    ws = _ws(_lib.load().vp_reduce_workspace_bytes(n), p)
    out = torch.empty((1,), dtype=torch.float32, device=p.device)
    _lib.call("vp_bce_sum_f32", _p(p), _p(t), n, _p(out), _p(ws), ws.numel() * 4, _stream())
    return out


def bce_bwd(p, t, g: Optional[torch.Tensor], gscale: float = 1.0):
    assert _same_layout(p, t)
    dp = torch.empty_like(p)
    _lib.call("vp_bce_bwd_f32", _p(p), _p(t), _p(g), float(gscale), _p(dp), p.numel(), _stream())
    return dp


def bce_sigmoid_bwd(p, t, gscale: float):
    assert _same_layout(p, t)
    dl = torch.empty_like(p)
    _lib.call("vp_bce_sigmoid_bwd_f32", _p(p), _p(t), float(gscale), _p(dl), p.numel(), _stream())
    return dl


def global_avgpool_fwd(x):
    """x: (B, C, H, W) channels_last -> (B, C) means."""
    B, C, H, W = x.shape
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    _lib.call("vp_global_avgpool_fwd_f32", _p(x), _p(out), B, H * W, C, _stream())
    return out


def global_avgpool_bwd(dy, shape):
    B, C, H, W = shape
    dx = empty_cl(B, C, H, W, dy)
    _lib.call("vp_global_avgpool_bwd_f32", _p(dy), _p(dx), B, H * W, C, _stream())
    return dx


def softmax_rows_fwd(x2d):
    R, n = x2d.shape
    y = torch.empty_like(x2d)
    _lib.call("vp_softmax_rows_fwd_f32", _p(x2d), _p(y), R, n, _stream())
    return y


def cross_entropy_fwd(logits2d: torch.Tensor, labels: torch.Tensor):
    """F.cross_entropy (mean) of (R, n) logits against int64 labels: (loss (1,), softmax rows (R, n))."""
    R, n = logits2d.shape
    if labels.dtype != torch.int64 or labels.shape != (R,) or not labels.is_cuda:
        raise _lib.VaePlayHipError("cross_entropy: labels must be an int64 device tensor of shape (rows,)")
    loss = torch.empty(1, dtype=torch.float32, device=logits2d.device)
    prob = torch.empty_like(logits2d)
    _lib.call("vp_cross_entropy_fwd_f32", _p(logits2d), c_void_p(labels.data_ptr()), _p(loss), _p(prob), R, n, _stream())
    return loss, prob


def cross_entropy_bwd(prob: torch.Tensor, labels: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    R, n = prob.shape
    dx = torch.empty_like(prob)
    _lib.call("vp_cross_entropy_bwd_f32", _p(prob), c_void_p(labels.data_ptr()), _p(g), _p(dx), R, n, _stream())
    return dx


def softmax_rows_bwd(y2d, dy2d):
    R, n = y2d.shape
    dx = torch.empty_like(y2d)
    _lib.call("vp_softmax_rows_bwd_f32", _p(y2d), _p(dy2d), _p(dx), R, n, _stream())
    return dx


def l1_mean(a, b):
    n = a.numel()
    ws = _ws(2 * _lib.load().vp_reduce_workspace_bytes(n), a)
    out = torch.empty((1,), dtype=torch.float32, device=a.device)
    _lib.call("vp_l1_mean_f32", _p(a), _p(b), n, _p(out), _p(ws), ws.numel() * 4, _stream())
    return out


def l1_mean_bwd(a, b, g, want_a: bool, want_b: bool):
    da = torch.empty_like(a) if want_a else None
    db = torch.empty_like(a) if want_b else None
    _lib.call("vp_l1_mean_bwd_f32", _p(a), _p(b), _p(g), _p(da), _p(db), a.numel(), _stream())
    return da, db


def be_loss_fwd(logits, targets, bce_weight: float, smooth: float):
    B, n = logits.shape[0], logits.numel() // logits.shape[0]
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    sums = torch.empty((B, 4), dtype=torch.float32, device=logits.device)
    ws = _ws(_lib.load().vp_be_loss_workspace_bytes(B, n), logits)
    _lib.call("vp_be_loss_fwd_f32", _p(logits), _p(targets), _p(loss), _p(sums), B, n, bce_weight, smooth, _p(ws), ws.numel() * 4,
              _stream())
    return loss, sums


def be_loss_bwd(logits, targets, sums, g, bce_weight: float, smooth: float):
    B, n = logits.shape[0], logits.numel() // logits.shape[0]
    dx = torch.empty_like(logits)
    _lib.call("vp_be_loss_bwd_f32", _p(logits), _p(targets), _p(sums), _p(g), _p(dx), B, n, bce_weight, smooth, _stream())
    return dx


def dice_loss_fwd(probs, targets, smooth: float):
    B, n = probs.shape[0], probs.numel() // probs.shape[0]
    loss = torch.empty(1, dtype=torch.float32, device=probs.device)
    sums = torch.empty((B, 4), dtype=torch.float32, device=probs.device)
    ws = _ws(_lib.load().vp_be_loss_workspace_bytes(B, n), probs)
    _lib.call("vp_dice_loss_fwd_f32", _p(probs), _p(targets), _p(loss), _p(sums), B, n, smooth, _p(ws), ws.numel() * 4, _stream())
    return loss, sums


def dice_loss_bwd(probs, targets, sums, g, smooth: float):
    B, n = probs.shape[0], probs.numel() // probs.shape[0]
    dp = torch.empty_like(probs)
    _lib.call("vp_dice_loss_bwd_f32", _p(probs), _p(targets), _p(sums), _p(g), _p(dp), B, n, smooth, _stream())
    return dp


def half_sqdiff(a, b):
    out = torch.empty_like(a)
    _lib.call("vp_half_sqdiff_f32", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def half_sqdiff_rowsum(a, b):
    R, n = a.shape[0], a.numel() // a.shape[0]
    out = torch.empty(R, dtype=torch.float32, device=a.device)
    _lib.call("vp_half_sqdiff_rowsum_f32", _p(a), _p(b), _p(out), R, n, _stream())
    return out


def half_sqdiff_bwd(a, b, g, per_row: bool, want_a: bool = True, want_b: bool = True):
    R, n = a.shape[0], a.numel() // a.shape[0]
    da = torch.empty_like(a) if want_a else None
    db = torch.empty_like(a) if want_b else None
    _lib.call("vp_half_sqdiff_bwd_f32", _p(a), _p(b), _p(g), _p(da), _p(db), R, n, 1 if per_row else 0, _stream())
    return da, db


def tensor_sum(x):
    n = x.numel()
    ws = _ws(_lib.load().vp_reduce_workspace_bytes(n), x)
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    _lib.call("vp_sum_f32", _p(x), n, _p(out), _p(ws), ws.numel() * 4, _stream())
    return out


# ---- optimiser -----------------------------------------------------------------------------------------
def adam_step(p, g, m, v, lr, beta1, beta2, eps, step: int, grad_scale: float = 1.0):
    _lib.call("vp_adam_f32", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
              int(step), float(grad_scale), _stream())


def adam_outer_step(p2d, m2d, v2d, A, Bm, lr, beta1, beta2, eps, step: int, grad_scale: float = 1.0):
    """Adam on p2d [R][Cn] with the gradient grad_scale * A^T Bm (A [K][R], Bm [K][Cn]) contracted inside the update."""
    R, Cn = p2d.shape
    K = A.shape[0]
    assert A.shape == (K, R) and Bm.shape == (K, Cn) and A.is_contiguous() and Bm.is_contiguous() and p2d.is_contiguous()
    _lib.call("vp_adam_outer_f32", _p(p2d), _p(m2d), _p(v2d), _p(A), _p(Bm), K, R, Cn, lr, beta1, beta2, eps, step, grad_scale, _stream())


def rmsprop_step(p, g, sq, lr, alpha, eps, grad_scale: float = 1.0):
    _lib.call("vp_rmsprop_f32", _p(p), _p(g), _p(sq), p.numel(), float(lr), float(alpha), float(eps), float(grad_scale), _stream())


# ---- split-bf16 ("bf16x3") operands ----------------------------------------------------------------
# A split tensor is an int16 tensor of shape (2, n): plane 0 = bf16(x), plane 1 = bf16(x - plane0).
def _pv(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.VaePlayHipError("vae_play_amd ops need tensors on the HIP device (no CPU path exists)")
    return c_void_p(t.data_ptr())


def empty_split(n: int, like: torch.Tensor) -> torch.Tensor:
    return torch.empty((2, n), dtype=torch.int16, device=like.device)


PARAM_EPOCH = [0]        # bumped by every flat-arena optimiser step: the arena kernels change weights without touching torch's version counters

SPLIT_BF16, SPLIT_F16 = 0, 1     # VP_SPLIT_* of include/vaeplay_hip.h: bf16 pair ("bf16x3" kernels) | fp16 pair ("f16x2" kernels)


def split_f32(x: torch.Tensor, fmt: int = SPLIT_BF16, scale: float = 1.0) -> torch.Tensor:
    """Split planes of ``scale * x`` (scale: a power of two; gradients on the fp16-pair format, see the header)."""
    x = x if x.is_contiguous() or _is_nhwc(x) else x.contiguous()
    out = empty_split(x.numel(), x)
    if fmt == SPLIT_BF16 and scale == 1.0:
        _lib.call("vp_split_f32", _p(x), _pv(out), x.numel(), _stream())
    else:
        _lib.call("vp_split_fmt_f32", _p(x), _pv(out), x.numel(), fmt, scale, _stream())
    return out


def split_pad(x: torch.Tensor, cpad: int) -> torch.Tensor:
    """Split planes of an NHWC activation (logical (B, C, H, W), channels_last) with the channels zero-padded to ``cpad``."""
    B, C, H, W = x.shape
    x = x if _is_nhwc(x) else x.contiguous(memory_format=torch.channels_last)
    out = empty_split(B * H * W * cpad, x)
    _lib.call("vp_split_pad_f32", _p(x), _pv(out), B * H * W, C, cpad, _stream())
    return out


def unsplit(s: torch.Tensor, fmt: int = SPLIT_BF16) -> torch.Tensor:
    """hi + lo as fp32 (test helper; torch ops)."""
    dt = torch.float16 if fmt == SPLIT_F16 else torch.bfloat16
    return s[0].view(dt).float() + s[1].view(dt).float()


def pack_w5_split(w_ref: torch.Tensor, want_p0: bool, want_p1: bool, fmt: int = SPLIT_BF16):
    Cs, Cb = w_ref.shape[0], w_ref.shape[1]
    w_ref = w_ref.contiguous()
    n = Cs * Cb * 25
    p0 = empty_split(n, w_ref) if want_p0 else None
    p1 = empty_split(n, w_ref) if want_p1 else None
    if fmt == SPLIT_BF16:
        _lib.call("vp_pack_w5_split", _p(w_ref), _pv(p0), _pv(p1), Cs, Cb, _stream())
    else:
        job = (_lib.PackJob * 1)(_lib.PackJob(w_ref.data_ptr(), p0.data_ptr() if want_p0 else None, p1.data_ptr() if want_p1 else None,
                                               Cs, Cb, Cs, 2))
        _lib.call("vp_pack_w5_batch", job, 1, _stream())
    return p0, p1


def pack_w5_p1_split_padded(w_ref: torch.Tensor, cs_pad: int = 8) -> torch.Tensor:
    """Packed P1 split planes [Cbig][25][cs_pad] of a conv whose small side has 1 or 3 channels, zero-padded to ``cs_pad``
    (the narrow side of the final conv, models/networks.py:100-103): its input gradient then runs on the MFMA kernels."""
    Cs, Cb = w_ref.shape[0], w_ref.shape[1]
    w_ref = w_ref.contiguous()
    out = empty_split(Cb * 25 * cs_pad, w_ref)
    _lib.call("vp_pack_w5_p1_split_padded", _p(w_ref), _pv(out), Cs, Cb, cs_pad, _stream())
    return out


def conv5_gather_bf16x3(big_split, shape_big, w_p0_split, Cs: int, bias, stride: int, act: int = ACT_NONE):
    B, Cb, Hb, Wb = shape_big
    Hs, Ws = Hb // stride, Wb // stride
    out = torch.empty((B, Cs, Hs, Ws), dtype=torch.float32, device=big_split.device, memory_format=torch.channels_last)
    _lib.call("vp_conv5_gather_bf16x3", _pv(big_split), _pv(w_p0_split), _p(bias), _p(out), B, Hs, Ws, Cb, Cs, stride, act, _stream())
    return out


def conv5_scatter_bf16x3(small_split, shape_small, w_p1_split, Cb: int, stride: int):
    B, Cs, Hs, Ws = shape_small
    out = torch.empty((B, Cb, Hs * stride, Ws * stride), dtype=torch.float32, device=small_split.device,
                      memory_format=torch.channels_last)
    _lib.call("vp_conv5_scatter_bf16x3", _pv(small_split), _pv(w_p1_split), _p(out), B, Hs, Ws, Cs, Cb, stride, _stream())
    return out


def conv5_wgrad_bf16x3(big_split, shape_big, small_split, shape_small, stride: int, out: Optional[torch.Tensor] = None):
    B, Cb, Hb, Wb = shape_big
    _, Cs, Hs, Ws = shape_small
    nbytes = _lib.load().vp_conv5_wgrad_bf16x3_workspace_bytes(B, Hs, Ws, Cb, Cs, stride)
    ws = torch.empty(max(4, (nbytes + 3) // 4), dtype=torch.float32, device=big_split.device)
    dw = _out(out, (Cs, Cb, 5, 5), big_split.device)
    _lib.call("vp_conv5_wgrad_bf16x3", _pv(big_split), _pv(small_split), _p(dw), B, Hs, Ws, Cb, Cs, stride, _p(ws), ws.numel() * 4, _stream())
    return dw


# ---- fp16-pair operands: products = 3 (~1e-6 relative) or 2 (two MFMAs per fragment pair, declared tolerance ~2e-4 per layer) ----
def conv5_gather_f16(big_split, shape_big, w_p0_split, Cs: int, bias, stride: int, act: int = ACT_NONE, products: int = 2,
                     out_scale: float = 1.0):
    B, Cb, Hb, Wb = shape_big
    Hs, Ws = Hb // stride, Wb // stride
    out = torch.empty((B, Cs, Hs, Ws), dtype=torch.float32, device=big_split.device, memory_format=torch.channels_last)
    _lib.call("vp_conv5_gather_f16", _pv(big_split), _pv(w_p0_split), _p(bias), _p(out), B, Hs, Ws, Cb, Cs, stride, act, products, out_scale,
              _stream())
    return out


def conv5_scatter_f16(small_split, shape_small, w_p1_split, Cb: int, stride: int, products: int = 2, out_scale: float = 1.0):
    B, Cs, Hs, Ws = shape_small
    out = torch.empty((B, Cb, Hs * stride, Ws * stride), dtype=torch.float32, device=small_split.device,
                      memory_format=torch.channels_last)
    _lib.call("vp_conv5_scatter_f16", _pv(small_split), _pv(w_p1_split), _p(out), B, Hs, Ws, Cs, Cb, stride, products, out_scale, _stream())
    return out


def conv5_wgrad_f16x2(big_split, shape_big, small_split, shape_small, stride: int, out_scale: float = 1.0):
    B, Cb, Hb, Wb = shape_big
    _, Cs, Hs, Ws = shape_small
    nbytes = _lib.load().vp_conv5_wgrad_bf16x3_workspace_bytes(B, Hs, Ws, Cb, Cs, stride)
    ws = torch.empty(max(4, (nbytes + 3) // 4), dtype=torch.float32, device=big_split.device)
    dw = torch.empty((Cs, Cb, 5, 5), dtype=torch.float32, device=big_split.device)
    _lib.call("vp_conv5_wgrad_f16x2", _pv(big_split), _pv(small_split), _p(dw), B, Hs, Ws, Cb, Cs, stride, out_scale, _p(ws),
              ws.numel() * 4, _stream())
    return dw


# ---- k x k convolution families (models/blocks.py vocabulary) ------------------------------------------------
def pack_w(w_ref: torch.Tensor, want_p0: bool, want_p1: bool):
    """w_ref [Cs][Cb][k][k] -> p0 [Cs][k*k][Cb], p1 [Cb][k*k][Cs]."""
    Cs, Cb, ks = w_ref.shape[0], w_ref.shape[1], w_ref.shape[2]
    assert w_ref.shape[3] == ks
    w_ref = w_ref.contiguous()
    p0 = torch.empty((Cs, ks * ks, Cb), dtype=torch.float32, device=w_ref.device) if want_p0 else None
    p1 = torch.empty((Cb, ks * ks, Cs), dtype=torch.float32, device=w_ref.device) if want_p1 else None
    _lib.call("vp_pack_w_f32", _p(w_ref), _p(p0), _p(p1), Cs, Cb, ks, _stream())
    return p0, p1


def conv_out_size(n: int, ks: int, stride: int) -> int:
    return (n + 2 * ((ks - 1) // 2) - ks) // stride + 1


def conv_gather(big, w_p0, bias, ks: int, stride: int, act: int = ACT_NONE):
    assert _is_nhwc(big)
    B, Cb, Hb, Wb = big.shape
    Cs = w_p0.shape[0]
    Hs, Ws = conv_out_size(Hb, ks, stride), conv_out_size(Wb, ks, stride)
    out = empty_cl(B, Cs, Hs, Ws, big)
    _lib.call("vp_conv_gather_f32", _p(big), _p(w_p0), _p(bias), _p(out), B, Hs, Ws, Hb, Wb, Cb, Cs, ks, stride, act, _stream())
    return out


def conv_scatter(small, w_p1, ks: int, stride: int, Hb: int, Wb: int):
    assert _is_nhwc(small)
    B, Cs, Hs, Ws = small.shape
    Cb = w_p1.shape[0]
    out = empty_cl(B, Cb, Hb, Wb, small)
    _lib.call("vp_conv_scatter_f32", _p(small), _p(w_p1), _p(out), B, Hs, Ws, Hb, Wb, Cs, Cb, ks, stride, _stream())
    return out


def conv_wgrad(big, small, ks: int, stride: int, out: Optional[torch.Tensor] = None):
    assert _is_nhwc(big) and _is_nhwc(small)
    B, Cb, Hb, Wb = big.shape
    _, Cs, Hs, Ws = small.shape
    nbytes = _lib.load().vp_conv_wgrad_workspace_bytes(B, Hs, Ws, Hb, Wb, Cb, Cs, ks, stride)
    ws = _ws(nbytes, big)
    dw = _out(out, (Cs, Cb, ks, ks), big.device)
    _lib.call("vp_conv_wgrad_f32", _p(big), _p(small), _p(dw), B, Hs, Ws, Hb, Wb, Cb, Cs, ks, stride, _p(ws), ws.numel() * 4, _stream())
    return dw


# ---- k x k split-bf16 ("bf16x3") forms ---------------------------------------------------------------------------------
def pack_w_split(w_ref: torch.Tensor, want_p0: bool, want_p1: bool):
    Cs, Cb, ks = w_ref.shape[0], w_ref.shape[1], w_ref.shape[2]
    w_ref = w_ref.contiguous()
    n = Cs * Cb * ks * ks
    p0 = empty_split(n, w_ref) if want_p0 else None
    p1 = empty_split(n, w_ref) if want_p1 else None
    _lib.call("vp_pack_w_split", _p(w_ref), _pv(p0), _pv(p1), Cs, Cb, ks, _stream())
    return p0, p1


def conv_gather_bf16x3(big_split, shape_big, w_p0_split, Cs: int, bias, ks: int, stride: int, act: int = ACT_NONE):
    B, Cb, Hb, Wb = shape_big
    Hs, Ws = conv_out_size(Hb, ks, stride), conv_out_size(Wb, ks, stride)
    out = torch.empty((B, Cs, Hs, Ws), dtype=torch.float32, device=big_split.device, memory_format=torch.channels_last)
    _lib.call("vp_conv_gather_bf16x3", _pv(big_split), _pv(w_p0_split), _p(bias), _p(out), B, Hs, Ws, Hb, Wb, Cb, Cs, ks, stride, act,
              _stream())
    return out


def conv_scatter_bf16x3(small_split, shape_small, w_p1_split, Cb: int, ks: int, stride: int, Hb: int, Wb: int):
    B, Cs, Hs, Ws = shape_small
    out = torch.empty((B, Cb, Hb, Wb), dtype=torch.float32, device=small_split.device, memory_format=torch.channels_last)
    _lib.call("vp_conv_scatter_bf16x3", _pv(small_split), _pv(w_p1_split), _p(out), B, Hs, Ws, Hb, Wb, Cs, Cb, ks, stride, _stream())
    return out


def conv_wgrad_bf16x3(big_split, shape_big, small_split, shape_small, ks: int, stride: int, out: Optional[torch.Tensor] = None):
    B, Cb, Hb, Wb = shape_big
    _, Cs, Hs, Ws = shape_small
    nbytes = _lib.load().vp_conv_wgrad_bf16x3_workspace_bytes(B, Hs, Ws, Hb, Wb, Cb, Cs, ks, stride)
    ws = _ws(nbytes, big_split)
    dw = _out(out, (Cs, Cb, ks, ks), big_split.device)
    _lib.call("vp_conv_wgrad_bf16x3", _pv(big_split), _pv(small_split), _p(dw), B, Hs, Ws, Hb, Wb, Cb, Cs, ks, stride, _p(ws),
              ws.numel() * 4, _stream())
    return dw


def upsample2x_fwd(x):
    assert _is_nhwc(x)
    B, C, H, W = x.shape
    y = empty_cl(B, C, 2 * H, 2 * W, x)
    _lib.call("vp_upsample2x_bilinear_fwd_f32", _p(x), _p(y), B, H, W, C, _stream())
    return y


def upsample2x_bwd(dy):
    assert _is_nhwc(dy)
    B, C, Ho, Wo = dy.shape
    dx = empty_cl(B, C, Ho // 2, Wo // 2, dy)
    _lib.call("vp_upsample2x_bilinear_bwd_f32", _p(dy), _p(dx), B, Ho // 2, Wo // 2, C, _stream())
    return dx


def add_coords(x, normalize: bool):
    assert _is_nhwc(x)
    B, C, H, W = x.shape
    out = empty_cl(B, C + 2, H, W, x)
    _lib.call("vp_add_coords_f32", _p(x), _p(out), B, H, W, C, int(normalize), _stream())
    return out


def slice_channels(x, Cout: int):
    assert _is_nhwc(x)
    B, C, H, W = x.shape
    out = empty_cl(B, Cout, H, W, x)
    _lib.call("vp_slice_channels_f32", _p(x), _p(out), B * H * W, C, Cout, _stream())
    return out
