"""Error behaviour of the C ABI on the device (include/vaeplay_hip.h): bad arguments and short workspaces are refused
with a status code and a message, nothing is launched, and the library stays usable afterwards."""
from ctypes import c_void_p

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def P(t):
    return None if t is None else c_void_p(t.data_ptr())


def test_bad_arguments_and_short_workspaces_are_refused():
    from vae_play_amd import _lib
    lib = _lib.load()
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(2 * 8 * 8 * 16, device=DEV)
    y = torch.empty(2 * 4 * 4 * 8, device=DEV)
    w = torch.randn(8 * 25 * 16, device=DEV)
    # null pointer
    assert lib.vp_conv5_gather_f32(None, P(w), None, P(y), 2, 4, 4, 16, 8, 2, 0, st) == -1
    assert b"null" in lib.vp_last_error().lower() or b"bad" in lib.vp_last_error().lower()
    # unsupported stride
    assert lib.vp_conv5_gather_f32(P(x), P(w), None, P(y), 2, 4, 4, 16, 8, 3, 0, st) == -1
    assert b"stride" in lib.vp_last_error()
    # split-bf16 path wants channel counts in multiples of 8
    xs = torch.empty((2, 2 * 8 * 8 * 12), dtype=torch.int16, device=DEV)
    ws_ = torch.empty((2, 8 * 25 * 12), dtype=torch.int16, device=DEV)
    assert lib.vp_conv5_gather_bf16x3(P(xs), P(ws_), None, P(y), 2, 4, 4, 12, 8, 2, 0, st) == -1
    assert b"multiple of 8" in lib.vp_last_error()
    # workspace too small -> VP_ERR_WORKSPACE, with the size query telling the truth
    need = lib.vp_conv5_wgrad_workspace_bytes(2, 4, 4, 16, 8, 2)
    assert need > 0
    dw = torch.empty(8 * 16 * 25, device=DEV)
    small = torch.empty(max(1, need // 4 - 64), device=DEV)
    assert lib.vp_conv5_wgrad_f32(P(x), P(y), P(dw), 2, 4, 4, 16, 8, 2, P(small), small.numel() * 4, st) == -3
    assert b"workspace" in lib.vp_last_error()
    assert lib.vp_bn_stats_f32(P(x), 128, 16, 1e-5, 0.9, P(torch.empty(16, device=DEV)), P(torch.empty(16, device=DEV)), None, None,
                               P(small), 4, st) == -3
    # k x k split-bf16 entry points: kernel sizes 1 / 3 / 5 only; padded planes need a multiple of 8 that covers C
    assert lib.vp_conv_gather_bf16x3(P(xs), P(ws_), None, P(y), 2, 4, 4, 8, 8, 16, 8, 2, 2, 0, st) == -1
    assert b"kernel size" in lib.vp_last_error()
    assert lib.vp_pack_w_split(P(w), P(ws_), None, 8, 16, 7, st) == -1 and b"kernel size" in lib.vp_last_error()
    pad = torch.empty((2, 64 * 8), dtype=torch.int16, device=DEV)
    assert lib.vp_split_pad_f32(P(x), P(pad), 64, 3, 12, st) == -1 and b"multiple of 8" in lib.vp_last_error()
    assert lib.vp_split_pad_f32(P(x), P(pad), 64, 9, 8, st) == -1
    assert lib.vp_split_pad_f32(P(x), P(pad), 64, 3, 8, st) == 0
    torch.cuda.synchronize()
    hi = pad[0].view(64, 8)
    assert int(hi[:, 3:].abs().max()) == 0, "padding channels must be zero"
    # skinny dense kernels: a split workspace that is too small is refused like everywhere else
    a_ = torch.randn(4, 4096, device=DEV); b_ = torch.randn(256, 4096, device=DEV); c_ = torch.empty(4, 256, device=DEV)
    need_g = lib.vp_gemm_workspace_bytes(4, 256, 4096)
    assert need_g > 0
    assert lib.vp_gemm_f32(P(a_), 4096, 1, P(b_), 4096, 1, P(c_), 256, None, 4, 256, 4096, 0, P(small), 16, st) == -3
    wsg = torch.empty(need_g // 4, device=DEV)
    assert lib.vp_gemm_f32(P(a_), 4096, 1, P(b_), 4096, 1, P(c_), 256, None, 4, 256, 4096, 0, P(wsg), need_g, st) == 0
    torch.cuda.synchronize()
    assert torch.allclose(c_.cpu(), a_.cpu() @ b_.cpu().t(), rtol=1e-4, atol=1e-3)
    # more than 32 layouts in one pack batch
    jobs = (_lib.PackJob * 20)(*[_lib.PackJob(w.data_ptr(), w.data_ptr(), w.data_ptr(), 8, 16, 0, 0) for _ in range(20)])
    assert lib.vp_pack_w5_batch(jobs, 20, st) == -1 and b"32" in lib.vp_last_error()
    # the library is still usable: the same call with a proper workspace succeeds and matches torch
    ws = torch.empty(need // 4 + 4, device=DEV)
    xb = torch.randn(2, 16, 8, 8)
    dyb = torch.randn(2, 8, 4, 4)
    xd = xb.to(DEV).contiguous(memory_format=torch.channels_last)
    dyd = dyb.to(DEV).contiguous(memory_format=torch.channels_last)
    assert lib.vp_conv5_wgrad_f32(P(xd), P(dyd), P(dw), 2, 4, 4, 16, 8, 2, P(ws), ws.numel() * 4, st) == 0
    wt = torch.zeros(8, 16, 5, 5, requires_grad=True)
    torch.nn.functional.conv2d(xb, wt, None, stride=2, padding=2).backward(dyb)
    torch.cuda.synchronize()
    assert torch.allclose(dw.view(8, 16, 5, 5).cpu(), wt.grad, rtol=1e-4, atol=1e-5)
    assert lib.vp_abi_version() >= 1


def test_python_wrappers_raise_on_failure():
    from vae_play_amd import _lib, ops
    with pytest.raises(_lib.VaePlayHipError):
        ops._p(torch.zeros(4))                                  # CPU tensor: no CPU path exists
    with pytest.raises(_lib.VaePlayHipError) as e:
        _lib.call("vp_sum_f32", None, 16, None, None, 0, None)
    assert "vp_sum_f32" in str(e.value)
