"""Flat-arena optimisers with the reference's call surface.

``optim.zero_grad(); loss.backward(); optim.step()`` (train_BE.py:62-64) with
``Adam(net.parameters(), lr=...)`` (train_BE.py:131, torch defaults) or
``RMSprop(params, lr=...)`` (train.py:136-140, torch defaults).

All parameters are moved into ONE contiguous fp32 arena (each tensor 256-B aligned) and their
``.grad`` into a second arena of the same layout, so that
  * the update is a single fused HIP kernel over the arena (HBM-bound: 28 B/param Adam,
    20 B/param RMSprop), and
  * data parallelism is a single RCCL all-reduce of the gradient arena (parallel.py).
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from . import _lib, ops

_ALIGN = 64  # floats (256 B)


def _clear_pending(p) -> None:
    p._vp_pending = False


class FlatArena:
    """Owns flat parameter / gradient buffers and re-points the nn.Parameters at views of them."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        allp: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not allp:
            raise ValueError("optimizer got an empty parameter list")
        # parameters that already live in another optimiser's arena (train_BE_font.py:280-281 builds one Adam over
        # net.parameters() and a second one over net.style_encoder.parameters()) stay where they are: this optimiser
        # keeps private state for them and updates them one tensor at a time
        self.foreign: List[torch.nn.Parameter] = [p for p in allp if getattr(p, "_vp_arena", None) is not None]
        self.params = [p for p in allp if getattr(p, "_vp_arena", None) is None]
        dev = allp[0].device
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be fp32 on one device")
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.flat_param = torch.zeros(max(off, 1), dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(max(off, 1), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
                p._vp_arena = self
                p._vp_off = o
                p._vp_pending = False
                # functional._grad_out hands this slice to ONE backward function per pass; autograd's accumulation re-arms it
                p.register_post_accumulate_grad_hook(_clear_pending)

    def grad_view(self, p) -> torch.Tensor:
        """``p``'s slice of the flat gradient buffer, shaped like ``p``"""
        return self.flat_grad[p._vp_off:p._vp_off + p.numel()].view_as(p)

    def zero_grad(self, set_to_none: bool = False) -> None:
        """set_to_none: drop the ``.grad`` views instead of zero-filling the arena (like ``module.zero_grad()``): the next backward
        pass then writes each parameter's gradient straight into its arena slice (functional._grad_out) and ``gather_grads`` zeroes
        the slices of parameters that received none."""
        if set_to_none:
            for p in self.params:
                p.grad = None
                p._vp_pending = False
            for p in self.foreign:
                p.grad = None
            return
        self.flat_grad.zero_()
        for p in self.foreign:
            if p.grad is not None:
                p.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # keep .grad pointing into the arena
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def adopt_views(self) -> None:
        """Re-point every ``.grad`` that is None at its arena slice WITHOUT touching the slice: the fused plans (engine.py,
        engine_gan.py) write gradients straight into the arena, so after ``optimizer.zero_grad(set_to_none=True)`` or the
        reference's ``module.zero_grad()`` (train.py:68) the slices hold this step's gradients although ``.grad`` is None --
        ``gather_grads()`` would otherwise take "None" for "received no gradient" and zero-fill them before the update."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
                p._vp_pending = False

    def gather_grads(self) -> None:
        """Bring gradients that live outside the arena back into it.  ``module.zero_grad()`` (train.py:68) sets
        ``.grad`` to None, after which autograd allocates fresh tensors: copy those in and re-point ``.grad``.
        A parameter that received no gradient contributes zeros (torch would skip it; its state then differs)."""
        base = self.flat_grad.data_ptr()
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_grad[o:o + p.numel()].view_as(p)
                if p.grad is None:
                    view.zero_()
                elif p.grad.data_ptr() != base + 4 * o:
                    view.copy_(p.grad)
                else:
                    continue
                p.grad = view


class _FlatOptimizer:
    def __init__(self, params, lr: float):
        self.arena = FlatArena(params)
        self.lr = lr
        self.grad_scale = 1.0   # set to 1/world_size by parallel.DataParallelStep after a sum all-reduce
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.arena.zero_grad(set_to_none)

    # ---- checkpointing: plain tensors / numbers only (vae_play_amd/checkpoint.py) -------------------------
    _STATE = ()

    def _hyper(self) -> dict:
        return {}

    def _foreign_state(self) -> list:
        """Per-tensor state of the parameters that live in another optimiser's arena, in parameter order."""
        return []

    def _load_foreign_state(self, items: list) -> None:
        if items:
            raise ValueError("optimizer state carries foreign-parameter entries this optimizer does not have")

    def state_dict(self) -> dict:
        """Arena moments, hyper-parameters and the per-tensor state of foreign parameters (those owned by another optimiser's
        arena: the font GAN's style-encoder Adam, train_BE_font.py:280-281, has nothing but such parameters) -- the reference
        pickles the whole optimiser (train.py:157); here everything is plain tensors and numbers."""
        out = {"kind": type(self).__name__, "lr": float(self.lr), "step_count": int(self.step_count), "numel": int(self.arena.numel),
               "hyper": self._hyper(), "foreign": self._foreign_state()}
        for k in self._STATE:
            out[k] = getattr(self, k).detach().cpu()
        return out

    def load_state_dict(self, sd: dict) -> None:
        if sd.get("kind") != type(self).__name__ or int(sd.get("numel", -1)) != self.arena.numel:
            raise ValueError("optimizer state does not match this optimizer (kind / parameter arena size)")
        foreign = sd.get("foreign", [])
        if len(foreign) != len(self.arena.foreign):
            raise ValueError(f"optimizer state has {len(foreign)} foreign-parameter entries, this optimizer has {len(self.arena.foreign)}")
        for item, p in zip(foreign, self.arena.foreign):
            for t in item["tensors"]:
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError("foreign-parameter state does not match the parameter's shape")
        self.lr, self.step_count = float(sd["lr"]), int(sd["step_count"])
        for k, v in sd.get("hyper", {}).items():
            setattr(self, k, tuple(v) if isinstance(v, (list, tuple)) else v)
        for k in self._STATE:
            getattr(self, k).copy_(sd[k])
        self._load_foreign_state(foreign)

    @property
    def flat_grad(self) -> torch.Tensor:
        return self.arena.flat_grad

    @property
    def flat_param(self) -> torch.Tensor:
        return self.arena.flat_param


class Adam(_FlatOptimizer):
    """torch.optim.Adam semantics (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, lr)
        self.betas, self.eps = betas, eps
        self.exp_avg = torch.zeros_like(self.arena.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.arena.flat_param)
        self._STATE = ("exp_avg", "exp_avg_sq")
        # parameters owned by another arena: per-tensor state and step count (torch skips a tensor without a gradient)
        self._fstate = [(p, torch.zeros_like(p.data), torch.zeros_like(p.data), [0]) for p in self.arena.foreign]

    def _hyper(self) -> dict:
        return {"betas": [float(self.betas[0]), float(self.betas[1])], "eps": float(self.eps)}

    def _foreign_state(self) -> list:
        return [{"tensors": [m.detach().cpu(), v.detach().cpu()], "count": int(cnt[0])} for _, m, v, cnt in self._fstate]

    def _load_foreign_state(self, items: list) -> None:
        for (_, m, v, cnt), item in zip(self._fstate, items):
            m.copy_(item["tensors"][0]); v.copy_(item["tensors"][1]); cnt[0] = int(item["count"])

    # ---- sliced update (engine.FusedVAEStep.step): begin_step() once, then step_range() per arena slice ---------
    def begin_step(self) -> None:
        self.step_count += 1

    @torch.no_grad()
    def step_range(self, lo: int, hi: int) -> None:
        """Update arena elements [lo, hi) (offsets are multiples of 64 floats) with the current step count; the
        gradients must already be in the arena (the fused engine writes them there)."""
        if hi > lo:
            ops.PARAM_EPOCH[0] += 1
            a = self.arena
            ops.adam_step(a.flat_param[lo:hi], a.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr,
                          self.betas[0], self.betas[1], self.eps, self.step_count, self.grad_scale)

    def set_outer_grad(self, param, A: torch.Tensor, Bm: torch.Tensor) -> None:
        """Declare that ``param`` (a [R][Cn] weight of this arena) has the gradient A^T Bm (A [K][R], Bm [K][Cn], static
        buffers): ``step(outer=True)`` then contracts it inside the update (ops.adam_outer_step) and does not read
        ``param.grad``."""
        a = self.arena
        off = a.offsets[[id(q) for q in a.params].index(id(param))]
        R, Cn = param.shape
        assert A.shape[1] == R and Bm.shape[1] == Cn and A.shape[0] == Bm.shape[0]
        self._outer = (off, (param.numel() + 63) // 64 * 64, R, Cn, A, Bm)

    @torch.no_grad()
    def step_outer_early(self, with_tail: bool = False) -> None:
        """The factored parameter's update of the step that ``step(outer=True)`` will complete, launched NOW on the current stream
        (the fused engine: on its side stream, as soon as both factors are final and the weight has been read for the last time in
        this step, underneath the rest of backward).  ``with_tail``: also the arena slice behind that parameter (the caller
        guarantees that its gradients are final and its parameters no longer read).  ``step(outer=True)`` skips what ran here."""
        off, span, R, Cn, A, Bm = self._outer
        n = R * Cn
        a = self.arena
        ops.PARAM_EPOCH[0] += 1
        ops.adam_outer_step(a.flat_param[off:off + n].view(R, Cn), self.exp_avg[off:off + n].view(R, Cn),
                            self.exp_avg_sq[off:off + n].view(R, Cn), A, Bm, self.lr, self.betas[0], self.betas[1], self.eps,
                            self.step_count + 1, self.grad_scale)
        if with_tail:
            lo, hi = off + span, a.flat_param.numel()
            if hi > lo:
                ops.adam_step(a.flat_param[lo:hi], a.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.lr,
                              self.betas[0], self.betas[1], self.eps, self.step_count + 1, self.grad_scale)
        self._outer_early_done = 2 if with_tail else 1

    @torch.no_grad()
    def step(self, outer: bool = False) -> None:
        self.step_count += 1
        ops.PARAM_EPOCH[0] += 1
        a = self.arena
        early = getattr(self, "_outer_early_done", 0)
        self._outer_early_done = 0
        if early and not (outer and getattr(self, "_outer", None) is not None):
            raise _lib.VaePlayHipError("step_outer_early() must be followed by step(outer=True)")
        if a.numel:
            a.gather_grads()
            if outer and getattr(self, "_outer", None) is not None:
                off, span, R, Cn, A, Bm = self._outer
                n = R * Cn
                self.step_range(0, off)
                if not early:
                    ops.adam_outer_step(a.flat_param[off:off + n].view(R, Cn), self.exp_avg[off:off + n].view(R, Cn),
                                        self.exp_avg_sq[off:off + n].view(R, Cn), A, Bm, self.lr, self.betas[0], self.betas[1], self.eps,
                                        self.step_count, self.grad_scale)
                if early < 2:
                    self.step_range(off + span, a.flat_param.numel())
            else:
                ops.adam_step(a.flat_param, a.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1],
                              self.eps, self.step_count, self.grad_scale)
        for p, m, v, cnt in self._fstate:
            if p.grad is None:
                continue
            cnt[0] += 1
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            ops.adam_step(p.data, g, m, v, self.lr, self.betas[0], self.betas[1], self.eps, cnt[0], self.grad_scale)


class RMSprop(_FlatOptimizer):
    """torch.optim.RMSprop semantics (alpha 0.99, eps 1e-8, no momentum, not centered)."""

    def __init__(self, params, lr: float = 1e-2, alpha: float = 0.99, eps: float = 1e-8):
        super().__init__(params, lr)
        self.alpha, self.eps = alpha, eps
        self.square_avg = torch.zeros_like(self.arena.flat_param)
        self._STATE = ("square_avg",)
        self._fstate = [(p, torch.zeros_like(p.data)) for p in self.arena.foreign]

    def _hyper(self) -> dict:
        return {"alpha": float(self.alpha), "eps": float(self.eps)}

    def _foreign_state(self) -> list:
        return [{"tensors": [sq.detach().cpu()], "count": 0} for _, sq in self._fstate]

    def _load_foreign_state(self, items: list) -> None:
        for (_, sq), item in zip(self._fstate, items):
            sq.copy_(item["tensors"][0])

    def begin_step(self) -> None:
        self.step_count += 1

    @torch.no_grad()
    def step_range(self, lo: int, hi: int) -> None:
        if hi > lo:
            ops.PARAM_EPOCH[0] += 1
            a = self.arena
            ops.rmsprop_step(a.flat_param[lo:hi], a.flat_grad[lo:hi], self.square_avg[lo:hi], self.lr, self.alpha, self.eps,
                             self.grad_scale)

    @torch.no_grad()
    def step(self) -> None:
        self.step_count += 1
        ops.PARAM_EPOCH[0] += 1
        a = self.arena
        if a.numel:
            a.gather_grads()
            ops.rmsprop_step(a.flat_param, a.flat_grad, self.square_avg, self.lr, self.alpha, self.eps, self.grad_scale)
        for p, sq in self._fstate:
            if p.grad is not None:
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.rmsprop_step(p.data, g, sq, self.lr, self.alpha, self.eps, self.grad_scale)
