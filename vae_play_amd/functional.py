"""Autograd glue: each op of the VAE step as a torch.autograd.Function whose forward and
backward are HIP kernels (via ops.py -> C ABI).  This is what lets the reference's
``loss.backward()`` idiom (train_BE.py:63) work unchanged on the drop-in modules.

Reference ops replaced (relative to the reference checkout):
  conv5x5            nn.Conv2d(k5,p2,stride)                    models/networks.py:14,100
  conv_transpose5x5  nn.ConvTranspose2d(k5,s2,p2,op1)           models/networks.py:38
  batch_norm_act     nn.BatchNorm2d/1d + F.relu (+ blocks acts) models/networks.py:16,28-29,66-67; models/blocks.py:19-30
  linear             nn.Linear                                  models/networks.py:65,69-70,88
  reparameterize     VaeGan.reparameterize                      models/networks.py:228-231
  kl_divergence      VaeGan.loss (kl term)                      models/networks.py:270
  binary_cross_entropy  F.binary_cross_entropy                  train_BE_font.py:107
"""
from __future__ import annotations

import os
import weakref

from typing import Optional

import torch
from torch.autograd import Function

from . import _lib, ops
from .ops import ACT_CODES, ACT_NONE, ACT_SIGMOID


def _cl(t: torch.Tensor) -> torch.Tensor:
    return ops.channels_last(t)


_PRECISION = "f32"


def set_conv_precision(mode: str) -> None:
    """Arithmetic of the 5x5 convolutions on the autograd front end: "f32" (exact fp32 MFMA, default) or "bf16x3"
    (split-bf16: three bf16 MFMAs per product, ~5e-6 relative error, ~3x the matrix rate; used where both channel
    counts are multiples of 8).  The fused engine has its own ``precision=`` argument."""
    global _PRECISION
    if mode not in ("f32", "bf16x3"):
        raise ValueError("precision must be 'f32' or 'bf16x3'")
    _PRECISION = mode


def get_conv_precision() -> str:
    return _PRECISION


_SMALL3 = True               # VALU kernels for the few-channel 3x3 convolutions (5.8 -> 3.8 ms per BE-heads step, profiles/r02_notes.md section 8)
_SMALL3_ALL = True
_BWD_SPLIT = True            # BatchNorm backward emits the split planes of dx


def _split_of(x: torch.Tensor) -> torch.Tensor:
    """bf16 hi/lo planes of an NHWC activation: the copy its producer emitted in the same pass (a BatchNorm+activation
    output in bf16x3 mode carries one, valid while the tensor has not been written since), else a split pass."""
    c = getattr(x, "_vp_split", None)
    if c is not None and c[1] == x._version and c[0].numel() == 2 * x.numel():
        return c[0]
    return ops.split_f32(x)


def _grad_out(param) -> Optional[torch.Tensor]:
    """Where a backward pass may WRITE ``param``'s gradient instead of returning a fresh tensor: its slice of the optimiser's flat
    gradient arena, when the parameter lives in one (optim.FlatArena) and has no gradient yet (``module.zero_grad()`` of
    train.py:68, or ``optimizer.zero_grad(set_to_none=True)``).  autograd's AccumulateGrad then adopts that view as ``.grad``
    without a kernel, where it would otherwise add a fresh tensor onto the zeroed arena -- one small add per parameter tensor
    and step (114 launches, 6 % of the VAE-GAN step).  A parameter used twice in one graph gets the view once (its hook clears
    the marker after accumulation); further contributions take the ordinary path and are added by autograd."""
    if param is None or param.grad is not None or not _DIRECT_GRADS:
        return None
    arena = getattr(param, "_vp_arena", None)
    if arena is None or getattr(param, "_vp_pending", False) or torch.is_grad_enabled():
        return None
    param._vp_pending = True
    return arena.grad_view(param)


_DIRECT_GRADS = True         # parameter gradients written straight into the arena slices (8.20 -> 8.00 ms per VAE-GAN step)


_PACKED = {}     # (id(weight), layout) -> (weakref(weight), weight._version, optimiser epoch, data_ptr, device, packed tensor)


def _evict_packed(wid: int) -> None:
    for k in [k for k in _PACKED if k[0] == wid]:
        _PACKED.pop(k, None)


def _packed(fn, weight, want_p1: bool):
    """``fn(weight, want_p0, want_p1)`` (an ops.pack_* re-layout of a conv weight) through a cache: a layer that runs several times
    per step -- the VAE-GAN's decoder decodes z and z_p, its discriminator sees both -- packs each layout once per weight VALUE.
    Valid while neither torch (``_version``) nor a flat-arena optimiser (ops.PARAM_EPOCH: its kernels update the weights behind
    torch's back) has changed the parameter and its storage is where it was (``module.to()`` / ``.float()`` re-point ``.data``
    without bumping the version); only leaf parameters are cached, and an entry dies with its parameter (weakref.finalize).
    Writes through ``p.data`` that keep the storage (``p.data.mul_()``, ``dist.broadcast(p.data)``) bump neither counter: follow them
    with ``ops.PARAM_EPOCH[0] += 1``."""
    if not (isinstance(weight, torch.nn.Parameter) and _PACK_CACHE_ON):
        r = fn(weight, not want_p1, want_p1)
        return r[1] if want_p1 else r[0]
    key = (id(weight), fn.__name__, want_p1, _PRECISION)
    hit = _PACKED.get(key)
    if (hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == ops.PARAM_EPOCH[0]
            and hit[3] == weight.data_ptr() and hit[4] == weight.device):
        return hit[5]
    r = fn(weight, not want_p1, want_p1)
    t = r[1] if want_p1 else r[0]
    if not any(k[0] == id(weight) for k in _PACKED):      # first entry of this parameter: drop its entries when it dies
        weakref.finalize(weight, _evict_packed, id(weight))
    _PACKED[key] = (weakref.ref(weight), weight._version, ops.PARAM_EPOCH[0], weight.data_ptr(), weight.device, t)
    return t


_PACK_CACHE_ON = True


def first_conv_s1_edge(weight, stride: int) -> bool:
    """Does conv5x5(x, weight, bias, stride) run on the rows-in-K forward kernel (which takes a ReLU in its epilogue)?"""
    return (_PRECISION == "bf16x3" and stride == 1 and weight.shape[1] in (1, 3) and weight.shape[0] in (32, 64))


def _use16(weight) -> bool:
    return _PRECISION == "bf16x3" and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0


class _Conv5(Function):
    """small = act(bias + conv5x5(big)); weight (Csmall, Cbig, 5, 5) = nn.Conv2d layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride: int, act: int):
        x = _cl(x)
        ctx.x16 = _use16(weight) and act in (ACT_NONE, ACT_SIGMOID)
        ctx.cols = None
        Cs, Cb = weight.shape[0], weight.shape[1]
        if (_PRECISION == "bf16x3" and stride == 2 and Cb in (1, 3) and Cs % 8 == 0 and act == ACT_NONE and bias is None
                and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
            # first conv on a 1- / 3-channel image (models/networks.py:14 via :55): its im2col written once as split planes, then a
            # 1x1 layer on the split-bf16 kernels for the forward pass and the weight gradient (as the fused step does)
            B, _, H, W = x.shape
            KC = _lib.load().vp_im2col5s2_cols(Cb)
            xcol = ops.empty_split(B * (H // 2) * (W // 2) * KC, x)
            _lib.call("vp_im2col5s2_split_f32", ops._p(x), ops._pv(xcol), B, Cb, H, W, 0, ops._stream())
            w0s = ops.empty_split(Cs * KC, x)
            _lib.call("vp_pack_w_im2col5_split", ops._p(weight.contiguous()), ops._pv(w0s), Cs, Cb, ops._stream())
            y = ops.conv_gather_bf16x3(xcol, (B, KC, H // 2, W // 2), w0s, Cs, None, 1, 1, ACT_NONE)
            ctx.cols = (KC, B, H, W)
            ctx.stride, ctx.act, ctx.has_bias = stride, act, False
            ctx.save_for_backward(xcol, weight, None)
            return y
        if first_conv_s1_edge(weight, stride) and act in (ACT_NONE, ops.ACT_RELU):
            # stride-1 first conv on a 1- / 3-channel image (the VAE-GAN discriminator's, models/networks.py:160-163) with bias and
            # ReLU in the epilogue: a kernel row of taps per MFMA k-step (csrc/edge.hip, the rows-in-K kernel's forward form)
            B, _, H, W = x.shape
            y = ops.empty_cl(B, Cs, H, W, x)
            _lib.call("vp_conv5_smallin_fwd_bf16x3", ops._p(x), ops._p(weight.contiguous()), ops._p(bias), ops._p(y), B, H, W, Cb, Cs, act,
                      ops._stream())
            ctx.x16 = False
            ctx.stride, ctx.act, ctx.has_bias = stride, act, bias is not None
            ctx.bias_param = bias
            ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
            return y
        if ctx.x16:
            xs = _split_of(x)
            p0 = _packed(ops.pack_w5_split, weight, False)
            y = ops.conv5_gather_bf16x3(xs, x.shape, p0, weight.shape[0], bias, stride, act)
            ctx.xshape = tuple(x.shape)
            x = xs          # the split copy is what the weight gradient reads
        else:
            p0 = _packed(ops.pack_w5, weight, False)
            # the image side of the final conv (64 -> 1 | 3 channels, models/networks.py:100-103) on the matrix cores: the edge
            # kernels of the fused step (taps in the MFMA columns / rows, a kernel row per k-step) instead of the VALU kernels
            ctx.edge = (_PRECISION == "bf16x3" and stride == 1 and weight.shape[1] == 64 and weight.shape[0] in (1, 3)
                        and act in (ACT_NONE, ACT_SIGMOID))
            if ctx.edge:
                B, _, H, W = x.shape
                y = ops.empty_cl(B, weight.shape[0], H, W, x)
                _lib.call("vp_conv5_smallout_bf16x3", ops._p(x), ops._p(p0), ops._p(bias), ops._p(y), B, H, W, 64, weight.shape[0], act,
                          ops._stream())
            else:
                y = ops.conv5_gather(x, p0, bias, stride, act)
        ctx.stride, ctx.act, ctx.has_bias = stride, act, bias is not None
        ctx.bias_param = bias
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _cl(dy)
        if ctx.act != ACT_NONE:
            dy = ops.act_bwd_from_y(y, dy, ctx.act)
        dx = dw = db = None
        if ctx.cols is not None:
            KC, B, H, W = ctx.cols
            Cs, Cb = weight.shape[0], weight.shape[1]
            if ctx.needs_input_grad[1]:
                dys = _split_of(dy)
                dwc = ops.conv_wgrad_bf16x3(x, (B, KC, H // 2, W // 2), dys, tuple(dy.shape), 1, 1)      # [Cs][KC] in column order
                dw = _grad_out(weight)
                dw = dw if dw is not None else torch.empty_like(weight, memory_format=torch.contiguous_format)
                _lib.call("vp_unpack_dw_im2col5_f32", ops._p(dwc), ops._p(dw), Cs, Cb, ops._stream())
            if ctx.needs_input_grad[0]:           # (an image that requires a gradient: the padded split-bf16 scatter)
                wpad = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 8 - Cb))
                _, p1 = ops.pack_w5_split(wpad, False, True)
                dx = ops.conv5_scatter_bf16x3(ops.split_f32(dy), dy.shape, p1, 8, ctx.stride)[:, :Cb]
            return dx, dw, None, None, None
        if ctx.x16:
            dys = _split_of(dy)
            if ctx.needs_input_grad[0]:
                p1 = _packed(ops.pack_w5_split, weight, True)
                dx = ops.conv5_scatter_bf16x3(dys, dy.shape, p1, weight.shape[1], ctx.stride)
            if ctx.needs_input_grad[1]:
                dw = ops.conv5_wgrad_bf16x3(x, ctx.xshape, dys, tuple(dy.shape), ctx.stride, out=_grad_out(weight))
        else:
            if ctx.needs_input_grad[0]:
                Cs, Cb = weight.shape[0], weight.shape[1]
                if getattr(ctx, "edge", False):
                    B, _, H, W = dy.shape
                    dx = ops.empty_cl(B, Cb, H, W, dy)
                    _lib.call("vp_conv5_smallin_dgrad_bf16x3", ops._p(dy), ops._p(weight.contiguous()), ops._p(dx), B, H, W, Cs, Cb,
                              ops._stream())
                elif _PRECISION == "bf16x3" and Cs < 8 and Cb % 8 == 0 and ctx.stride == 1:
                    # the image side of the final conv (1 or 3 channels): pad it to 8 channels and run the input gradient on
                    # the split-bf16 halo kernel like the fused engine does (the exact-f32 scatter with K = 25 * Cs is a 94 %
                    # padded MFMA tile: 494 us against ~60 us at 32 images of 128 x 128)
                    B, _, H, W = dy.shape
                    dyp = torch.zeros((B, H, W, 8), dtype=torch.float32, device=dy.device)
                    dyp[..., :Cs] = dy.permute(0, 2, 3, 1)
                    dx = ops.conv5_scatter_bf16x3(ops.split_f32(dyp), (B, 8, H, W), ops.pack_w5_p1_split_padded(weight, 8), Cb, 1)
                elif (_PRECISION == "bf16x3" and Cb in (1, 3) and Cs in (32, 64) and ctx.stride == 1):
                    # input gradient of a stride-1 first conv (the VAE-GAN discriminator's 1 -> 32, models/networks.py:160-163:
                    # its input is the decoder's output): dx[p] = sum_{tap, co} dy[p - tap + 2][co] W[co][ci][tap] is the
                    # "many channels -> 1 | 3" correlation of the tap-in-N kernel with the taps flipped
                    B, _, H, W = dy.shape
                    wf = weight.flip(2, 3).permute(1, 2, 3, 0).reshape(Cb, 25, Cs).contiguous()      # [ci][tap'][co]
                    dx = ops.empty_cl(B, Cb, H, W, dy)
                    _lib.call("vp_conv5_smallout_bf16x3", ops._p(dy), ops._p(wf), None, ops._p(dx), B, H, W, Cs, Cb, ACT_NONE, ops._stream())
                elif _PRECISION == "bf16x3" and Cb < 8 and Cs % 8 == 0:
                    # the image side of a FIRST conv whose input gradient is needed (the VAE-GAN discriminator's, models/
                    # networks.py:160-163: its input is the decoder's output): zero-pad the weight's input channels to 8,
                    # scatter on the split-bf16 kernels, keep the real channels.  The exact-f32 scatter to one output channel
                    # pads the MFMA tile's 32 columns by 97 %: 522 us against 32 + 214 us at 48 images of 128 x 128
                    # (tools/microbench_narrow_dgrad.py)
                    wpad = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, 8 - Cb))
                    _, p1 = ops.pack_w5_split(wpad, False, True)
                    dx = ops.conv5_scatter_bf16x3(ops.split_f32(dy), dy.shape, p1, 8, ctx.stride)[:, :Cb]
                else:
                    p1 = _packed(ops.pack_w5, weight, True)
                    dx = ops.conv5_scatter(dy, p1, ctx.stride)
            if ctx.needs_input_grad[1]:
                B, Cs, H, W = dy.shape
                nb = (_lib.load().vp_conv5_smallout_wgrad_bf16x3_workspace_bytes(B, H, W, weight.shape[1], Cs)
                      if getattr(ctx, "edge", False) else 0)
                if nb:
                    ws = torch.empty(nb // 4, dtype=torch.float32, device=dy.device)
                    dw = _grad_out(weight)
                    dw = dw if dw is not None else torch.empty_like(weight, memory_format=torch.contiguous_format)
                    _lib.call("vp_conv5_smallout_wgrad_bf16x3", ops._p(x), ops._p(dy), ops._p(dw), B, H, W, weight.shape[1], Cs, ops._p(ws), nb,
                              ops._stream())
                else:
                    dw = ops.conv5_wgrad(x, dy, ctx.stride, out=_grad_out(weight))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            B, C, H, W = dy.shape
            db = ops.colsum(dy.permute(0, 2, 3, 1).reshape(B * H * W, C), out=_grad_out(ctx.bias_param))
        return dx, dw, db, None, None


class _ConvT5(Function):
    """big = convT5x5(small), stride 2, padding 2, output_padding 1; weight (Csmall, Cbig, 5, 5)
    = nn.ConvTranspose2d layout (in_channels first)."""

    @staticmethod
    def forward(ctx, x, weight, stride: int):
        x = _cl(x)
        ctx.x16 = _use16(weight)
        if ctx.x16:
            xs = _split_of(x)
            p1 = _packed(ops.pack_w5_split, weight, True)
            y = ops.conv5_scatter_bf16x3(xs, x.shape, p1, weight.shape[1], stride)
            ctx.xshape = tuple(x.shape)
            x = xs
        else:
            p1 = _packed(ops.pack_w5, weight, True)
            y = ops.conv5_scatter(x, p1, stride)
        ctx.stride = stride
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = _cl(dy)
        dx = dw = None
        if ctx.x16:
            dys = _split_of(dy)
            if ctx.needs_input_grad[0]:
                p0 = _packed(ops.pack_w5_split, weight, False)
                dx = ops.conv5_gather_bf16x3(dys, dy.shape, p0, weight.shape[0], None, ctx.stride, ACT_NONE)
            if ctx.needs_input_grad[1]:
                dw = ops.conv5_wgrad_bf16x3(dys, tuple(dy.shape), x, ctx.xshape, ctx.stride, out=_grad_out(weight))
            return dx, dw, None
        if ctx.needs_input_grad[0]:
            p0 = _packed(ops.pack_w5, weight, False)
            dx = ops.conv5_gather(dy, p0, None, ctx.stride, ACT_NONE)
        if ctx.needs_input_grad[1]:
            dw = ops.conv5_wgrad(dy, x, ctx.stride, out=_grad_out(weight))
        return dx, dw, None


class _BatchNormAct(Function):
    """y = act(BN(x)); x is (B,C,H,W) channels_last or (B,F).  Training mode uses batch statistics
    and updates the running buffers in place with torch's momentum semantics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training: bool, momentum: float, eps: float,
                act: int, slope: float):
        x = _cl(x) if x.dim() == 4 else x.contiguous()
        if training:
            mean, rstd = ops.bn_stats(x, eps, momentum, running_mean, running_var)
        else:
            mean = running_mean
            rstd = torch.rsqrt(running_var + eps)
        if _PRECISION == "bf16x3" and x.dim() == 4 and x.shape[1] % 8 == 0:
            # the consumer is almost always a convolution on the split-bf16 kernels: emit its operand planes from this pass
            y, ys = ops.bn_act_fwd(x, mean, rstd, gamma, beta, act, slope, want_split=True)
            y._vp_split = (ys, y._version)
        else:
            y = ops.bn_act_fwd(x, mean, rstd, gamma, beta, act, slope)
        ctx.act, ctx.slope, ctx.training = act, slope, training
        ctx.save_for_backward(x, mean, rstd, gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, beta = ctx.saved_tensors
        dy = _cl(dy) if dy.dim() == 4 else dy.contiguous()
        need_affine = gamma is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        og = _grad_out(gamma) if (need_affine and ctx.needs_input_grad[1]) else None
        ob = _grad_out(beta) if (need_affine and ctx.needs_input_grad[2]) else None
        if _PRECISION == "bf16x3" and _BWD_SPLIT and x.dim() == 4 and x.shape[1] % 8 == 0 and ctx.needs_input_grad[0]:
            # dx is almost always the output gradient of a convolution on the split-bf16 kernels: emit its operand planes from
            # this pass (the tensor object travels through the autograd engine with its attribute; _split_of checks the version)
            dx, dgamma, dbeta, dxs = ops.bn_act_bwd(x, dy, mean, rstd, gamma, beta, ctx.act, ctx.slope, ctx.training, need_affine,
                                                    out_dgamma=og, out_dbeta=ob, want_split=True)
            dx._vp_split = (dxs, dx._version)
        else:
            dx, dgamma, dbeta = ops.bn_act_bwd(x, dy, mean, rstd, gamma, beta, ctx.act, ctx.slope, ctx.training, need_affine,
                                               out_dgamma=og, out_dbeta=ob)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x = x.contiguous()
        y = ops.linear_fwd(x, weight, bias)
        ctx.has_bias = bias is not None
        ctx.bias_param = bias
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_dgrad(dy, weight)
        if ctx.needs_input_grad[1]:
            dw = ops.linear_wgrad(dy, x, out=_grad_out(weight))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dy, out=_grad_out(ctx.bias_param))
        return dx, dw, db


class _FlattenNCHW(Function):
    """(B,C,H,W) channels_last -> (B, C*H*W) in the reference's (C,H,W) flatten order
    (``ten.view(len(ten), -1)``, models/networks.py:74) via the HIP transpose."""

    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        ctx.shape = x.shape
        return ops.nhwc_to_nchw(x).reshape(x.shape[0], -1)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = ctx.shape
        return ops.nchw_to_nhwc(dy.contiguous().view(B, C, H, W))


class _UnflattenNCHW(Function):
    """(B, C*H*W) -> (B,C,H,W) channels_last (``ten.view(len(ten), -1, 8, 8)``, models/networks.py:110)."""

    @staticmethod
    def forward(ctx, x, C: int, H: int, W: int):
        B = x.shape[0]
        return ops.nchw_to_nhwc(x.contiguous().view(B, C, H, W))

    @staticmethod
    def backward(ctx, dy):
        dy = _cl(dy)
        return ops.nhwc_to_nchw(dy).reshape(dy.shape[0], -1), None, None, None


class _ConvK(Function):
    """nn.Conv2d(kernel k in {1,3,5}, stride 1|2, padding (k-1)//2) -- models/blocks.py:9-17."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride: int):
        x = _cl(x)
        ks = weight.shape[2]
        ctx.x16 = _use16(weight)
        ctx.xshape = tuple(x.shape)
        ctx.pad = None
        Co, Ci = weight.shape[0], weight.shape[1]
        # a handful of channels on both sides (the networks_BE heads): the weight gradient is a reduction over the pixels into a few
        # hundred numbers -- exact-fp32 VALU kernel (csrc/small3.hip) instead of a 64 x 64 MFMA tile that is 92 - 98 % padding
        ctx.small3 = (_SMALL3 and ks == 3 and stride == 1 and x.dtype == torch.float32
                      and ops.conv3_small_wgrad_applicable(x.shape[0], x.shape[2], x.shape[3], Ci, Co))
        x32 = x if ctx.small3 else None
        ctx.direct3 = ctx.small3 and _SMALL3_ALL
        if ctx.direct3:                 # forward and input gradient too: one output pixel per thread, no packing / padding / split
            y = ops.conv3_small_fwd(x, weight.contiguous(), bias)
            ctx.stride, ctx.ks, ctx.has_bias = stride, ks, bias is not None
            ctx.bias_param = bias
            ctx.save_for_backward(x, weight, x32)
            return y
        if _PRECISION == "bf16x3" and not ctx.x16:
            # a channel count that is not a multiple of 8 on either side (32 + 2 coordinate channels in, 1 or 2 channels out):
            # zero-pad it to the next multiple of 8 and stay on the split-bf16 kernels -- the exact-f32 tile kernels run these
            # shapes with 70-97 % padding of their own (networks_BE heads: 65 % of the step in one f32 weight-gradient kernel)
            Cop, Cip = (Co + 7) // 8 * 8, (Ci + 7) // 8 * 8
            wp = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, Cip - Ci, 0, Cop - Co)) if (Cop != Co or Cip != Ci) else weight
            bp = None if bias is None else (torch.nn.functional.pad(bias, (0, Cop - Co)) if Cop != Co else bias)
            xs = ops.split_pad(x, Cip) if Cip != Ci else _split_of(x)
            B, _, H, W = x.shape
            p0, _ = ops.pack_w_split(wp, True, False)
            y = ops.conv_gather_bf16x3(xs, (B, Cip, H, W), p0, Cop, bp, ks, stride, ACT_NONE)
            ctx.pad = (Co, Ci, Cop, Cip)
            ctx.stride, ctx.ks, ctx.has_bias = stride, ks, bias is not None
            ctx.bias_param = bias
            ctx.save_for_backward(xs, weight, x32)
            return y[:, :Co] if Cop != Co else y
        if ctx.x16:                     # split-bf16 kernels (set_conv_precision("bf16x3"), channel counts multiples of 8)
            xs = _split_of(x)
            p0 = _packed(ops.pack_w_split, weight, False)
            y = ops.conv_gather_bf16x3(xs, x.shape, p0, weight.shape[0], bias, ks, stride, ACT_NONE)
            x = xs
        else:
            p0 = _packed(ops.pack_w, weight, False)
            y = ops.conv_gather(x, p0, bias, ks, stride, ACT_NONE)
        ctx.stride, ctx.ks, ctx.has_bias = stride, ks, bias is not None
        ctx.bias_param = bias
        ctx.save_for_backward(x, weight, x32)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, x32 = ctx.saved_tensors
        dy = _cl(dy)
        dx = dw = db = None
        small_dw = None
        if ctx.small3 and ctx.needs_input_grad[1]:
            small_dw = ops.conv3_small_wgrad(x32, dy, out=_grad_out(weight))
        if ctx.direct3:
            if ctx.needs_input_grad[0]:
                dx = ops.conv3_small_dgrad(dy, weight.contiguous())
            if ctx.has_bias and ctx.needs_input_grad[2]:
                B, C, H, W = dy.shape
                db = ops.colsum(dy.permute(0, 2, 3, 1).reshape(B * H * W, C), out=_grad_out(ctx.bias_param))
            return dx, small_dw, db, None
        if ctx.pad is not None:
            Co, Ci, Cop, Cip = ctx.pad
            B, _, Ho, Wo = dy.shape
            dys = ops.split_pad(dy, Cop) if Cop != Co else _split_of(dy)
            if ctx.needs_input_grad[0]:
                wp = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, Cip - Ci, 0, Cop - Co)) if (Cop != Co or Cip != Ci) else weight
                _, p1 = ops.pack_w_split(wp, False, True)
                dx = ops.conv_scatter_bf16x3(dys, (B, Cop, Ho, Wo), p1, Cip, ctx.ks, ctx.stride, ctx.xshape[2], ctx.xshape[3])
                dx = dx[:, :Ci] if Cip != Ci else dx
            if small_dw is not None:
                dw = small_dw
            elif ctx.needs_input_grad[1]:
                dw = ops.conv_wgrad_bf16x3(x, (B, Cip, ctx.xshape[2], ctx.xshape[3]), dys, (B, Cop, Ho, Wo), ctx.ks, ctx.stride)
                dw = dw[:Co, :Ci].contiguous() if (Cop != Co or Cip != Ci) else dw
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = ops.colsum(dy.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Co))
            return dx, dw, db, None
        if ctx.x16:
            dys = _split_of(dy)
            if ctx.needs_input_grad[0]:
                p1 = _packed(ops.pack_w_split, weight, True)
                dx = ops.conv_scatter_bf16x3(dys, dy.shape, p1, weight.shape[1], ctx.ks, ctx.stride, ctx.xshape[2], ctx.xshape[3])
            if small_dw is not None:
                dw = small_dw
            elif ctx.needs_input_grad[1]:
                dw = ops.conv_wgrad_bf16x3(x, ctx.xshape, dys, tuple(dy.shape), ctx.ks, ctx.stride, out=_grad_out(weight))
        else:
            if ctx.needs_input_grad[0]:
                p1 = _packed(ops.pack_w, weight, True)
                dx = ops.conv_scatter(dy, p1, ctx.ks, ctx.stride, x.shape[2], x.shape[3])
            if small_dw is not None:
                dw = small_dw
            elif ctx.needs_input_grad[1]:
                dw = ops.conv_wgrad(x, dy, ctx.ks, ctx.stride, out=_grad_out(weight))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            B, C, H, W = dy.shape
            db = ops.colsum(dy.permute(0, 2, 3, 1).reshape(B * H * W, C), out=_grad_out(ctx.bias_param))
        return dx, dw, db, None


class _Act(Function):
    """Standalone activation (conv + bias + act blocks with bn=None, models/blocks.py:24-30)."""

    @staticmethod
    def forward(ctx, x, act: int, slope: float):
        y = ops.act_fwd(x if (x.dim() != 4 or ops._is_nhwc(x)) else _cl(x), act, slope)
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _cl(dy) if dy.dim() == 4 else dy.contiguous()
        return ops.act_bwd_from_y(y, dy, ctx.act, ctx.slope), None, None


class _InstanceNormAct(Function):
    """nn.InstanceNorm2d (affine=False, no running stats, eps 1e-5) + activation: per-(image, channel) statistics, one
    batched launch sequence (the BatchNorm kernels with blockIdx.z = image)."""

    @staticmethod
    def forward(ctx, x, eps: float, act: int, slope: float):
        x = _cl(x)
        if _PRECISION == "bf16x3" and _BWD_SPLIT and x.shape[1] % 8 == 0:
            # as _BatchNormAct: the consumer is almost always a convolution on the split-bf16 kernels
            y, mean, rstd, ys = ops.instnorm_act_fwd(x, eps, act, slope, want_split=True)
            y._vp_split = (ys, y._version)
        else:
            y, mean, rstd = ops.instnorm_act_fwd(x, eps, act, slope)
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        if _PRECISION == "bf16x3" and _BWD_SPLIT and x.shape[1] % 8 == 0:
            dx, dxs = ops.instnorm_act_bwd(x, _cl(dy), mean, rstd, ctx.act, ctx.slope, want_split=True)
            dx._vp_split = (dxs, dx._version)
            return dx, None, None, None
        return ops.instnorm_act_bwd(x, _cl(dy), mean, rstd, ctx.act, ctx.slope), None, None, None


class _Upsample2x(Function):
    """F.interpolate(scale_factor=2, mode='bilinear') -- models/blocks.py:145."""

    @staticmethod
    def forward(ctx, x):
        return ops.upsample2x_fwd(_cl(x))

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2x_bwd(_cl(dy))


class _AddCoords(Function):
    """AddCoords (models/blocks.py:97-112)."""

    @staticmethod
    def forward(ctx, x, normalize: bool):
        ctx.C = x.shape[1]
        return ops.add_coords(_cl(x), normalize)

    @staticmethod
    def backward(ctx, dy):
        return ops.slice_channels(_cl(dy), ctx.C), None


class _Reparam(Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        z, _ = ops.latent_fwd(mu, logvar, eps, want_kl=False)
        ctx.save_for_backward(mu, logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        mu, logvar, eps = ctx.saved_tensors
        dmu, dlv = ops.latent_bwd(mu, logvar, eps, dz, None, 0.0)
        return dmu, dlv, None


class _KL(Function):
    """kl[b] = -0.5 * sum_j(1 + logvar - mu^2 - exp(logvar))."""

    @staticmethod
    def forward(ctx, mu, logvar):
        zeros = torch.zeros_like(mu)
        _, kl = ops.latent_fwd(mu, logvar, zeros, want_kl=True)
        ctx.save_for_backward(mu, logvar, zeros)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        mu, logvar, zeros = ctx.saved_tensors
        dmu, dlv = ops.latent_bwd(mu, logvar, zeros, None, gkl, 0.0)
        return dmu, dlv


class _BCESum(Function):
    @staticmethod
    def forward(ctx, p, t):
        if p.dim() == 4:
            p, t = _cl(p), _cl(t)
        else:
            p, t = p.contiguous(), t.contiguous()
        out = ops.bce_sum(p, t)
        ctx.save_for_backward(p, t)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        return ops.bce_bwd(p, t, g.reshape(1).contiguous(), 1.0), None


class _GlobalAvgPool(Function):
    """nn.AdaptiveAvgPool2d((1, 1)) + flatten: (B, C, H, W) -> (B, C)."""

    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        ctx.shape = tuple(x.shape)
        return ops.global_avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.global_avgpool_bwd(dy.contiguous(), ctx.shape)


class _SelfAttention(Function):
    """out = gamma * (V^T att^T) + x with att = softmax(Q K^T) per image (models/blocks.py:77-96); q, k, v are the
    already-projected (B, c', H, W) maps.  Per image: three HIP GEMMs + a row softmax; N = H*W = 1 (the only use in
    the font model, models/networks_BE_font.py:29-45) short-circuits to out = gamma * v + x."""

    @staticmethod
    def forward(ctx, x, q, k, v, gamma):
        x, q, k, v = _cl(x), _cl(q), _cl(k), _cl(v)
        B, C, H, W = x.shape
        N, C8 = H * W, q.shape[1]
        ctx.dims = (B, C, H, W, C8)
        if N == 1:
            att, o = None, v
        else:
            qm, km, vm = (t.permute(0, 2, 3, 1).reshape(B, N, -1) for t in (q, k, v))     # views of the NHWC memory
            att = torch.empty((B, N, N), dtype=torch.float32, device=x.device)
            o_m = torch.empty((B, N, C), dtype=torch.float32, device=x.device)
            for b in range(B):
                e = ops.gemm(qm[b], C8, 1, km[b], C8, 1, N, N, C8, 0)                    # energy = Q K^T
                att[b] = ops.softmax_rows_fwd(e)
                ops.gemm(att[b], N, 1, vm[b], 1, C, N, C, N, 1, out=o_m[b])              # out = att V
            o = o_m.view(B, H, W, C).permute(0, 3, 1, 2)
        ctx.save_for_backward(q, k, v, att, o, gamma)
        return gamma * o + x

    @staticmethod
    def backward(ctx, g):
        q, k, v, att, o, gamma = ctx.saved_tensors
        B, C, H, W, C8 = ctx.dims
        N = H * W
        g = _cl(g)
        dgamma = (g * o).sum().reshape(1)
        do = gamma * g
        if N == 1:
            return g, torch.zeros_like(q), torch.zeros_like(k), do, dgamma
        qm, km, vm = (t.permute(0, 2, 3, 1).reshape(B, N, -1) for t in (q, k, v))
        dom = _cl(do).permute(0, 2, 3, 1).reshape(B, N, C)
        dq, dk, dv = (torch.empty((B, N, n), dtype=torch.float32, device=g.device) for n in (C8, C8, C))
        for b in range(B):
            ops.gemm(att[b], 1, N, dom[b], 1, C, N, C, N, 2, out=dv[b])                  # dV = att^T dO
            datt = ops.gemm(dom[b], C, 1, vm[b], C, 1, N, N, C, 0)                       # dAtt = dO V^T
            de = ops.softmax_rows_bwd(att[b], datt)
            ops.gemm(de, N, 1, km[b], 1, C8, N, C8, N, 1, out=dq[b])                     # dQ = dE K
            ops.gemm(de, 1, N, qm[b], 1, C8, N, C8, N, 2, out=dk[b])                     # dK = dE^T Q
        to4 = lambda t, c: t.view(B, H, W, c).permute(0, 3, 1, 2)
        return g, to4(dq, C8), to4(dk, C8), to4(dv, C), dgamma


class _L1Mean(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        return ops.l1_mean(a, b).reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da, db = ops.l1_mean_bwd(a, b, g.reshape(1).contiguous(), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return da, db


class _CrossEntropy(Function):
    """F.cross_entropy(logits, labels) with torch's defaults (mean), on vp_cross_entropy_{fwd,bwd}_f32."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, prob = ops.cross_entropy_fwd(logits.contiguous(), labels.contiguous())
        ctx.save_for_backward(prob, labels)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        prob, labels = ctx.saved_tensors
        return ops.cross_entropy_bwd(prob, labels.contiguous(), g.reshape(1).contiguous()), None


class _BELoss(Function):
    """bce_weight * BCEWithLogits(x, t) (mean) + dice(sigmoid(x), t) as one reduction + one elementwise backward."""

    @staticmethod
    def forward(ctx, logits, targets, bce_weight: float, smooth: float):
        logits, targets = logits.contiguous(), targets.contiguous()
        loss, sums = ops.be_loss_fwd(logits, targets, bce_weight, smooth)
        ctx.w, ctx.smooth = bce_weight, smooth
        ctx.save_for_backward(logits, targets, sums)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, targets, sums = ctx.saved_tensors
        return ops.be_loss_bwd(logits, targets, sums, g.reshape(1).contiguous(), ctx.w, ctx.smooth), None, None, None


class _DiceLoss(Function):
    """1 - mean_b (2 sum(p t) + s) / (sum p + sum t + s) on probabilities (no gradient to the targets)."""

    @staticmethod
    def forward(ctx, probs, targets, smooth: float):
        probs, targets = probs.contiguous(), targets.contiguous()
        loss, sums = ops.dice_loss_fwd(probs, targets, smooth)
        ctx.smooth = smooth
        ctx.save_for_backward(probs, targets, sums)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        probs, targets, sums = ctx.saved_tensors
        return ops.dice_loss_bwd(probs, targets, sums, g.reshape(1).contiguous(), ctx.smooth), None, None


class _HalfSqDiff(Function):
    """0.5*(a-b)^2 per element ("nle", models/networks.py:267) or summed over all but the first dim (":273")."""

    @staticmethod
    def forward(ctx, a, b, rowsum: bool):
        a, b = a.contiguous(), b.contiguous()
        ctx.rowsum = rowsum
        ctx.save_for_backward(a, b)
        return ops.half_sqdiff_rowsum(a, b) if rowsum else ops.half_sqdiff(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da, db = ops.half_sqdiff_bwd(a, b, g.contiguous(), ctx.rowsum, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return da, db, None


# ---- public functional API ----------------------------------------------------------------------
def conv5x5(x, weight, bias=None, stride: int = 2, act: Optional[str] = None):
    return _Conv5.apply(x, weight, bias, stride, ACT_CODES[act])


def conv_transpose5x5(x, weight, stride: int = 2):
    return _ConvT5.apply(x, weight, stride)


def batch_norm_act(x, gamma, beta, running_mean, running_var, training: bool, momentum: float = 0.1, eps: float = 1e-5,
                   act: Optional[str] = "relu", slope: float = 0.0):
    return _BatchNormAct.apply(x, gamma, beta, running_mean, running_var, training, momentum, eps, ACT_CODES[act], slope)


def linear(x, weight, bias=None):
    return _Linear.apply(x, weight, bias)


def conv2d(x, weight, bias=None, stride: int = 1):
    """k x k convolution, k in {1,3,5}, padding (k-1)//2 (models/blocks.py Conv2d)."""
    return _ConvK.apply(x, weight, bias, stride)


def activation(x, act: Optional[str], slope: float = 0.0):
    code = ACT_CODES[act]
    return x if code == ACT_NONE else _Act.apply(x, code, slope)


def instance_norm_act(x, eps: float = 1e-5, act: Optional[str] = None, slope: float = 0.0):
    return _InstanceNormAct.apply(x, eps, ACT_CODES[act], slope)


def upsample2x_bilinear(x):
    return _Upsample2x.apply(x)


def add_coords(x, normalize: bool = False):
    return _AddCoords.apply(x, normalize)


def flatten_nchw(x):
    return _FlattenNCHW.apply(x)


def unflatten_nchw(x, C: int, H: int, W: int):
    return _UnflattenNCHW.apply(x, C, H, W)


def reparameterize(mu, logvar, *, eps: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
    """z = eps * exp(0.5*logvar) + mu (models/networks.py:228-231).  The reference draws eps with
    ``normal_()``; here it may be injected (parity tests) or drawn with torch's device RNG."""
    if eps is None:
        eps = torch.randn(mu.shape, dtype=mu.dtype, device=mu.device, generator=generator)
    return _Reparam.apply(mu, logvar, eps)


def kl_divergence(mu, logvar):
    """Per-sample KL of models/networks.py:270, shape (B,)."""
    return _KL.apply(mu, logvar)


def binary_cross_entropy(p, t, reduction: str = "sum"):
    """F.binary_cross_entropy with torch's -100 log clamp; reduction 'sum' or 'mean'."""
    s = _BCESum.apply(p, t)
    if reduction == "sum":
        return s
    if reduction == "mean":
        return s / p.numel()
    raise ValueError("reduction must be 'sum' or 'mean'")


def global_avg_pool(x):
    """nn.AdaptiveAvgPool2d((1, 1)) followed by the flatten of models/networks_BE_font.py:66: (B, C, H, W) -> (B, C)."""
    return _GlobalAvgPool.apply(x)


def self_attention(x, q, k, v, gamma):
    return _SelfAttention.apply(x, q, k, v, gamma)


def cross_entropy(logits, labels):
    """``F.cross_entropy(logits, labels)`` for (B, classes) logits and int64 class indices (train_BE_GAN.py:135,159: the
    discriminator's type losses; train_BE_font.py:109): a HIP kernel, like every other loss of the steps."""
    return _CrossEntropy.apply(logits, labels)


def l1_loss(a, b):
    """F.l1_loss(a, b) (mean)."""
    return _L1Mean.apply(a, b)


def be_loss(logits, targets, bce_weight: float = 0.5, smooth: float = 1.0):
    """train_BE.py:58-59: ``bce_weight * F.binary_cross_entropy_with_logits(logits, targets)
    + compute_dice_loss(logits.sigmoid(), targets)`` (tools/ops.py:12-19) for (B, ...) logits; scalar."""
    return _BELoss.apply(logits, targets, bce_weight, smooth)


def dice_loss(probs, targets, smooth: float = 1.0):
    """tools/ops.py:12-19 / :178-185 on probabilities."""
    return _DiceLoss.apply(probs, targets, smooth)


_EDGE_KERNEL = None


def edge_loss(mask_probs, mask_targets):
    """tools/ops.py:187-215: dice loss between |Laplacian| edge maps of prediction and target (single-channel maps;
    3x3 kernel [[-1,-1,-1],[-1,8,-1],[-1,-1,-1]] / 8, zero padding).  The filter is a fixed-weight 3x3 convolution on the
    HIP conv kernel; |e| = relu(e) + relu(-e) (the activation kernels differentiate through their OUTPUT, so an
    even function has to be assembled from monotone pieces)."""
    global _EDGE_KERNEL
    if _EDGE_KERNEL is None or _EDGE_KERNEL.device != mask_probs.device:
        k = torch.full((3, 3), -1.0)
        k[1, 1] = 8.0
        _EDGE_KERNEL = (k / 8).reshape(1, 1, 3, 3).to(mask_probs.device)
    def absmap(x):
        e = conv2d(x, _EDGE_KERNEL, None, 1)
        return activation(e, "relu") + activation(-e, "relu")

    return dice_loss(absmap(mask_probs), absmap(mask_targets).detach())


def half_sq_diff(a, b):
    """0.5 * (a - b) ** 2, elementwise."""
    return _HalfSqDiff.apply(a, b, False)


def half_sq_diff_rowsum(a, b):
    """torch.sum(0.5 * (a - b) ** 2, 1) for 2-D a, b."""
    return _HalfSqDiff.apply(a, b, True)


def vae_loss(x, x_tilde, mu, logvar):
    """(sum-BCE + sum-KL) / B -- the composed loss of SURVEY.md 3.3.  Returns (loss, recon, kl)."""
    recon = binary_cross_entropy(x_tilde, x, "sum")
    kl = kl_divergence(mu, logvar).sum()
    return (recon + kl) / x.shape[0], recon, kl
