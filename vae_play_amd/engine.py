"""Fused VAE training step: the whole forward + loss + backward of SURVEY.md 3.3 as a fixed,
pre-planned sequence of HIP kernel launches (no autograd, no per-step allocation).

The plan is built once for a (model, batch size) pair:
  * every activation / gradient buffer is preallocated NHWC in HBM and reused across steps;
  * every parameter gradient is written exactly once, directly into the optimiser's flat gradient
    arena (no zero_grad pass, no accumulate pass) -- the arena is then all-reduced once (RCCL) and
    consumed by the fused optimiser kernel;
  * the launch list is replayable and hipGraph-capturable (``capture()``); the eager form stays the default because it runs
    the weight gradients on a side stream underneath the main chain, which a replayed graph serialises;
  * three arithmetic modes (``precision=``): exact fp32, split-bf16 (three MFMAs per product) and fp16 pairs with two MFMAs per
    product on the backward layers.

It drives the same C-ABI entry points as the autograd modules and must produce identical
gradients (tests/test_gpu_engine.py).  Reference path replaced: the loop body train_BE.py:54-64
with the model/loss of models/networks.py (Encoder :72-78, reparameterize :228-231, Decoder
:107-112, KL :270) and F.binary_cross_entropy.
"""
from __future__ import annotations

import math
import os
from ctypes import c_void_p
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib, ops, parallel
from .networks import VAE

_ACT_RELU, _ACT_NONE, _ACT_SIGMOID = ops.ACT_RELU, ops.ACT_NONE, ops.ACT_SIGMOID


class _Plan:
    """A list of (c_function, argument list) with the stream slot patched at run time.
    ``flops`` is the algorithmic FLOP count of a call (0 for bandwidth-bound glue); ``timers`` lets
    bench.py bracket selected calls with HIP events on the launch stream.

    A call added with ``side=k`` runs on the side stream (after everything enqueued on the main stream so far) and
    records side event k when it is done; ``wait_side(k)`` makes the main stream wait for that event.  Used to run
    the weight-gradient GEMMs underneath the HBM-bound BatchNorm backward kernels of the next layer."""

    def __init__(self):
        self.calls: List[list] = []

    def add(self, name: str, *args, flops: float = 0.0, tag: str = "", side: Optional[int] = None, side_args: Optional[dict] = None):
        """``side_args`` = {argument index: (value on the main stream, value on the side stream)}: arguments that depend on where
        the call ends up at run time (the weight gradients' CU budget: the whole chip alone, part of it beside the main stream)."""
        fn = getattr(_lib.load(), name)
        a = list(args) + [None]  # last argument of every entry point is the stream
        self.calls.append([name, fn, a, len(a) - 1, flops, tag, side, side_args])

    def add_first(self, name: str, *args, side: Optional[int] = None):
        fn = getattr(_lib.load(), name)
        a = list(args) + [None]
        self.calls.insert(0, [name, fn, a, len(a) - 1, 0.0, "", side, None])

    def wait_side(self, k: int):
        self.calls.append(["__wait_side__", None, [k], 0, 0.0, "", None, None])

    def hook(self, key: str):
        """A named point of the plan: ``run(..., hooks={key: fn})`` calls ``fn(side)`` there (between two launches)."""
        self.calls.append(["__hook__", None, [key], 0, 0.0, "", None, None])

    def run(self, stream_ptr: int, timers: Optional[dict] = None, start: int = 0, stop: Optional[int] = None, side=None, hooks=None):
        """``side`` = a ``_SideCtx`` (side stream, its events, the fork event) or None (everything on the main stream)."""
        s = c_void_p(stream_ptr)
        for ci, (name, fn, a, slot, flops, tag, sev, sargs) in enumerate(self.calls[start:stop], start):
            if name == "__hook__":
                if hooks is not None and a[0] in hooks:
                    hooks[a[0]](side)
                continue
            if fn is None:                                   # main stream waits for a side event
                if side is not None:
                    side.flush_if_pending(a[0])              # (an event that was never recorded would not be waited for)
                    torch.cuda.current_stream().wait_event(side[1][a[0]])
                continue
            on_side = side is not None and sev is not None
            if sargs is not None:
                for idx, (v_main, v_side) in sargs.items():
                    a[idx] = v_side if on_side else v_main
            if on_side:
                a[slot] = c_void_p(side[0].cuda_stream)
                side.defer(name, fn, a, sev)                 # launched by the next flush (one fork for several launches)
                continue
            a[slot] = s
            timed = timers is not None and name in timers["names"]
            if timed:
                pool = timers.get("pool")
                if pool is not None:      # events are created once per (instrumented-step slot, call) and re-recorded
                    key = (id(self), ci, timers.get("slot", 0))
                    if key not in pool:
                        pool[key] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    e0, e1 = pool[key]
                else:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = fn(*a)
            if timed:
                e1.record()
                timers["events"].append((name, tag, flops, e0, e1))
            if rc != 0:
                _lib.check(rc, name)


class _SideCtx:
    """Side stream of a fused plan: ``ctx[0]`` = the stream, ``ctx[1]`` = one event per side launch, ``ctx[2]`` = the fork event.

    A fork (event record on the main stream + wait on the side stream) costs the MAIN stream ~6 us: the kernel behind the
    record starts that much later (rocprofv3 timeline of the concurrent step: fifteen such gaps, ~90 us per step).  Side launches
    can therefore be DEFERRED and handed over in batches of ``batch`` (VP_SIDE_BATCH) behind one fork -- measured SLOWER
    (3.756 ms at 1, 3.850 at 2, 3.825 at 3, 3.811 at 4, each with enough gradient planes in the rotation; profiles/r02_notes.md
    section 7): a weight gradient that starts one layer late no longer runs underneath the next layer's HBM-bound BatchNorm
    backward but underneath its MFMA-bound input gradient.  The default is 1 = fork per launch.  ``flush()`` forks and launches
    everything pending; it must run before the main stream joins or waits for the side stream, and ``flush_if_pending(k)``
    before a wait for side event k (the plans' buffer rotation: main must not rewrite a gradient plane that a deferred weight
    gradient has yet to read)."""

    def __init__(self, n_events: int):
        self.stream = torch.cuda.Stream()
        self.events = [torch.cuda.Event() for _ in range(n_events)]
        self.fork = torch.cuda.Event()
        self.batch = 1          # side launches per fork (batching them was measured slower, see above)
        self.pending: List[tuple] = []

    def __getitem__(self, i):
        return (self.stream, self.events, self.fork)[i]

    def defer(self, name, fn, args, sev):
        self.pending.append((name, fn, list(args), sev))     # (argument list copied: input slots are re-bound per step)
        if len(self.pending) >= self.batch:
            self.flush()

    def flush_if_pending(self, sev):
        if any(p[3] == sev for p in self.pending):
            self.flush()

    def flush(self):
        if not self.pending:
            return
        self.fork.record()                                   # the side stream starts after the main stream's work so far
        self.stream.wait_event(self.fork)
        pend, self.pending = self.pending, []
        for name, fn, a, sev in pend:
            rc = fn(*a)
            self.events[sev].record(self.stream)
            if rc != 0:
                _lib.check(rc, name)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _noop():
    return None


class FusedVAEStep:
    """forward + loss + backward (+ all-reduce + optimiser) for a ``networks.VAE``.

    ``optimizer`` must be a flat-arena optimiser from ``vae_play_amd.optim`` built over
    ``vae.parameters()`` (its gradient arena receives the gradients).
    """

    def __init__(self, vae: VAE, optimizer, batch_size: int, img_size: int, channels: int, group=None,
                 precision: str = "bf16x3", grad_scale16: float = 4096.0):
        """precision: "f32"    -- every contraction on v_mfma_f32_32x32x2_f32 (exact fp32, ~1e-6 parity);
                      "bf16x3" -- 5x5 convolutions whose channel counts are multiples of 8 run on the
                                  split-bf16 kernel (3 bf16 MFMAs per product, fp32 accumulate, ~1e-5 parity);
                                  edge layers and dense layers stay on the f32 kernels;
                      "f16x2"  -- the same plan with those convolutions on fp16-pair operands: forward layers with 3 fp16 MFMAs
                                  per product (outputs keep the ~1e-5 parity), BACKWARD layers (input and weight gradients) with
                                  2 (DECLARED tolerance ~2e-4 relative per layer on gradients; the final convolution's narrow
                                  side stays on its bf16x3 kernels).  ``grad_scale16`` (a power of two) is the factor
                                  gradient operands are multiplied by before they are written as fp16 pairs -- the consuming
                                  launch divides it out of its accumulators, so every fp32 buffer and the gradient arena hold
                                  true values; gradient elements beyond 65504 / grad_scale16 saturate."""
        if precision not in ("f32", "bf16x3", "f16x2"):
            raise ValueError("precision must be 'f32', 'bf16x3' or 'f16x2'")
        if precision == "f16x2" and not (grad_scale16 > 0 and math.log2(grad_scale16).is_integer()):
            raise ValueError("grad_scale16 must be a positive power of two")
        self.precision = precision
        self.grad_scale16 = float(grad_scale16)
        self.vae, self.opt, self.B, self.S, self.C = vae, optimizer, batch_size, img_size, channels
        self.Z, self.L = vae.z_size, vae.iter_level
        self.group = group
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
        self.opt.grad_scale = 1.0 / self.world
        dev = next(vae.parameters()).device
        if dev.type != "cuda":
            raise _lib.VaePlayHipError("FusedVAEStep needs the model on the HIP device")
        self.dev = dev
        self._bufs: Dict[str, torch.Tensor] = {}
        self._graph = None
        self.reload_switches()
        self._build()
        # BatchNorm ``num_batches_tracked`` is advanced lazily (sync_counters): make every state_dict() / checkpoint of the model
        # see the true counters
        import weakref
        me = weakref.ref(self)

        def _sync(module, prefix, keep_vars):
            o = me()
            if o is not None:
                o.sync_counters()
        self._sd_hook = vae.register_state_dict_pre_hook(_sync)

    def reload_switches(self) -> None:
        """The A/B switches that shape a STEP (as opposed to the plan, which reads its own in ``_build``) are resolved here, once,
        when the engine is built -- not on every ``step()``.  The in-process A/B tools (tools/ab_env.py) call this again after
        flipping an environment variable."""
        env = os.environ.get
        self._side_wgrad = env("VP_SIDE_WGRAD", "1") != "0"          # weight gradients on the side stream
        self._adam_outer = env("VP_ADAM_OUTER", "1") != "0"          # one rank: fc.0's update contracted from its factors
        self._adam_outer_early = int(env("VP_ADAM_OUTER_EARLY", "2"))  # 0 end of step | 1 fc.0 early | 2 fc.0 + the slice behind it
        self._dp_factored = env("VP_DP_FACTORED", "1") != "0"        # several ranks: fc.0's gradient exchanged as its factors
        self._dp_enc_tail = env("VP_DP_ENC_TAIL", "1") != "0"        # several ranks: the deep encoder convs in their own bucket

    # ---- buffers ----------------------------------------------------------------------------
    def _buf(self, name: str, *shape) -> torch.Tensor:
        t = torch.empty(shape, dtype=torch.float32, device=self.dev)
        self._bufs[name] = t
        return t

    def _ws(self, name: str, nbytes: int) -> torch.Tensor:
        return self._buf(name, max(4, (nbytes + 3) // 4))

    def _sbuf(self, name: str, n: int) -> torch.Tensor:
        """split tensor: (2, n) int16 = bf16 hi plane + bf16 lo plane"""
        t = torch.empty((2, n), dtype=torch.int16, device=self.dev)
        self._bufs[name] = t
        return t

    # ---- plan construction ------------------------------------------------------------------
    def _build(self):
        lib = _lib.load()
        B, S, C, Z, L = self.B, self.S, self.C, self.Z, self.L
        enc, dec = self.vae.encoder, self.vae.decoder
        fwd, bwd = _Plan(), _Plan()
        self._bn_momentum_eps = {}
        P = _ptr
        pack_jobs = []   # every conv weight is re-packed by ONE launch at the head of the forward plan
        # Weight gradients go to a side stream (bf16x3 plans): they only feed the optimiser, so they can run underneath
        # the next layer's HBM-bound BatchNorm backward.  The split output gradient they read is ping-ponged between
        # two buffers; before a buffer is rewritten the main stream waits for the weight gradient that read it.
        x2 = self.precision == "f16x2"                 # fp16-pair planes + the *_f16x2 launches (same plan structure)
        # sticky device flag: a producer of fp16 gradient planes clamped a value (|g| * grad_scale16 > 65504); read in sync_counters()
        self._f16_sat = torch.zeros(1, dtype=torch.int32, device=self.dev)
        x3 = self.precision in ("bf16x3", "f16x2")
        # (exact-f32 plans keep everything on the main stream: with their weight gradients on the side stream -- fp32 output gradients
        # rotating over two buffers, 128 / 160 / 192 CUs -- the step measured 8.39 / 8.06 / 8.03 ms against 7.89 ms in line; with only
        # the bandwidth-bound glue there -- slab reductions, final conv's VALU weight gradient, weight re-pack, early Adam slices, one
        # category at a time -- 7.72 - 7.79 ms against 7.66 in line: beside an fp32-MFMA kernel ANY second kernel costs more than it
        # hides, profiles/r03_notes.md sections 2 and 8)
        side_on = x3
        FMT = 1 if x2 else 0                           # VP_SPLIT_F16 | VP_SPLIT_BF16
        GS = self.grad_scale16 if x2 else 1.0          # scale of gradient planes; 1/GS in the launches that consume them
        n_side = [0]

        # fp16 plans: forward layers contract with three products (outputs keep the bf16x3 tolerance), backward layers with two
        # (the decoder forward on two products measured 3.51 -> 3.375 ms but 40x the ReLU-mask flips, every forward layer on two
        # products puts mu outside the 1e-3 bar: profiles/r02_notes.md section 4)
        FWD_PRODUCTS = DEC_FWD_PRODUCTS = 3

        def add_gather(plan, a_s, w_s, bias, out, geom, act, alpha=1.0, products=2, **kw):
            if x2:
                plan.add("vp_conv5_gather_f16", P(a_s), P(w_s), P(bias), P(out), *geom, act, products, alpha, **kw)
            else:
                plan.add("vp_conv5_gather_bf16x3", P(a_s), P(w_s), P(bias), P(out), *geom, act, **kw)

        def add_scatter(plan, a_s, w_s, out, geom, alpha=1.0, products=2, **kw):
            if x2:
                plan.add("vp_conv5_scatter_f16", P(a_s), P(w_s), P(out), *geom, products, alpha, **kw)
            else:
                plan.add("vp_conv5_scatter_bf16x3", P(a_s), P(w_s), P(out), *geom, **kw)

        # CU budget of a weight gradient: the whole chip on the main stream, WGRAD_SIDE_CUS beside the main stream's kernels
        side_cus = int(os.environ.get("VP_WGRAD_SIDE_CUS", "160"))
        main_cus = int(os.environ.get("VP_WGRAD_MAIN_CUS", "0"))      # (tests: the same budget on both streams = the same arithmetic)

        def add_wgrad(plan, big_s, small_s, dw, geom, ws, alpha=1.0, **kw):
            if x2:
                plan.add("vp_conv5_wgrad_f16x2_cus", P(big_s), P(small_s), P(dw), *geom, alpha, 0, P(ws), ws.numel() * 4,
                         side_args={4 + len(geom): (main_cus, side_cus)}, **kw)
            else:
                plan.add("vp_conv5_wgrad_bf16x3_cus", P(big_s), P(small_s), P(dw), *geom, 0, P(ws), ws.numel() * 4,
                         side_args={3 + len(geom): (main_cus, side_cus)}, **kw)

        def side_slot():
            if not side_on:
                return None
            n_side[0] += 1
            return n_side[0] - 1

        # the batched weight re-pack (65 us, ~180 MB of traffic) also runs on the side stream, underneath the first
        # encoder block, whose own (fp32, 3-channel) pack stays on the main stream
        k_pack = side_slot()
        first_pack_jobs = []

        def pack(weight, p0, p1, Cs, Cb, split, Cs_pad=0, first=False, bf16=False):
            """split: planes in the plan's format (bf16=True forces bf16 pairs: the final conv's kernels read those)"""
            (first_pack_jobs if (first and k_pack is not None) else pack_jobs).append(_lib.PackJob(weight.data_ptr(), p0.data_ptr() if p0 is not None else None,
                                          p1.data_ptr() if p1 is not None else None, Cs, Cb, Cs_pad, (2 if (x2 and not bf16) else 1) if split else 0))

        def grad_of(p: torch.nn.Parameter) -> torch.Tensor:
            arena = getattr(p, "_vp_arena", None)
            if arena is None:
                raise _lib.VaePlayHipError("parameter has no arena gradient; build the optimiser first")
            return arena.grad_view(p)        # (not p.grad: optimizer.zero_grad(set_to_none=True) drops that attribute, not the slice)

        def use16(cin, cout):
            return x3 and cin % 8 == 0 and cout % 8 == 0

        fuse_stats = os.environ.get("VP_FUSE_BN_STATS", "1") != "0"
        small_bn = True            # single-launch BatchNorm for <= 64 rows (-33 us per step, profiles/r02_notes.md section 2)

        def bn_block(tag, x_buf, R, Cn, bn_mod, y_buf, y_split=None, conv=None):
            """stats + fused normalise/ReLU (fp32 and/or split output); returns the saved (mean, rstd).
            Momentum and eps are the module's (models/networks.py:16,40,66,89 use momentum=0.9, eps=1e-5).
            ``conv`` = (family, entry point, leading arguments, geometry, flops, tag) of the split-bf16 convolution that produces
            ``x_buf``: when that launch shape can emit the statistics from its epilogue the convolution and the statistics
            become ONE call (vp_conv5_*_stats_bf16x3) and the activation is not read again for them."""
            mean, rstd = self._buf(f"{tag}.mean", Cn), self._buf(f"{tag}.rstd", Cn)
            ws = self._ws(f"{tag}.bnws", lib.vp_bn_workspace_bytes(R, Cn))
            mom, eps_bn = float(bn_mod.momentum), float(bn_mod.eps)
            self._bn_momentum_eps[id(bn_mod)] = (mom, eps_bn)
            fused = False
            if conv is not None:
                family, name, lead, geom, fl, ctag = conv
                qgeom = geom if family == 0 else (geom[0], geom[1], geom[2], geom[4], geom[3], geom[5])   # query takes (Cbig, Csmall)
                query = (lib.vp_conv5_stats_f16_workspace_bytes if x2 else lib.vp_conv5_stats_workspace_bytes) if x3 else lib.vp_conv5_stats_f32_workspace_bytes
                nst = query(family, *qgeom) if fuse_stats else 0
                if nst:
                    st = self._ws(f"{tag}.statws", nst)
                    fwd.add(name.replace("_f32", "_stats_f32") if not x3 else name.replace("_bf16x3", "_stats_f16" if x2 else "_stats_bf16x3"), *lead, *geom, *(((FWD_PRODUCTS if family == 0 else DEC_FWD_PRODUCTS),) if x2 else ()),
                            eps_bn, mom, P(mean), P(rstd), P(bn_mod.running_mean), P(bn_mod.running_var), P(st), st.numel() * 4,
                            flops=fl, tag=ctag)
                    fused = True
                else:
                    extra = (None,) if family == 0 else ()
                    tail = ((_ACT_NONE,) if family == 0 else ()) + (((FWD_PRODUCTS if family == 0 else DEC_FWD_PRODUCTS), 1.0) if x2 else ())
                    fwd.add(name.replace("_bf16x3", "_f16") if x2 else name, lead[0], lead[1], *extra, lead[2], *geom, *tail, flops=fl, tag=ctag)
            if not fused and conv is None and y_split is None and R <= 64 and Cn % 4 == 0 and small_bn:
                # the dense layers' BatchNorm1d (R = batch rows): statistics + finalisation + normalise/ReLU in ONE launch
                fwd.add("vp_bn_small_fwd_f32", P(x_buf), R, Cn, eps_bn, mom, P(bn_mod.weight), P(bn_mod.bias), P(mean), P(rstd),
                        P(bn_mod.running_mean), P(bn_mod.running_var), P(y_buf), _ACT_RELU, 0.0)
                return mean, rstd, ws
            if not fused:
                fwd.add("vp_bn_stats_f32", P(x_buf), R, Cn, eps_bn, mom, P(mean), P(rstd), P(bn_mod.running_mean),
                        P(bn_mod.running_var), P(ws), ws.numel() * 4)
            if x2 and y_split is not None:
                fwd.add("vp_bn_act_fwd_split_fmt_f32", P(x_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias), P(y_buf),
                        P(y_split), R, Cn, _ACT_RELU, 0.0, FMT)
            else:
                fwd.add("vp_bn_act_fwd_split_f32", P(x_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias), P(y_buf),
                        P(y_split), R, Cn, _ACT_RELU, 0.0)
            return mean, rstd, ws

        def bn_block_bwd(x_buf, dy_buf, dx_buf, R, Cn, bn_mod, mean, rstd, ws, dx_split=None):
            if dx_split is None and R <= 64 and Cn % 4 == 0 and small_bn:
                bwd.add("vp_bn_small_bwd_f32", P(x_buf), P(dy_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias), P(dx_buf),
                        P(grad_of(bn_mod.weight)), P(grad_of(bn_mod.bias)), R, Cn, _ACT_RELU, 0.0, 1)
                return
            if x2 and dx_split is not None:      # gradient planes: fp16 pairs of GS * dx
                bwd.add("vp_bn_act_bwd_split_fmt_sat_f32", P(x_buf), P(dy_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias),
                        P(dx_buf), P(dx_split), P(grad_of(bn_mod.weight)), P(grad_of(bn_mod.bias)), R, Cn, _ACT_RELU, 0.0, 1, FMT, GS,
                        c_void_p(self._f16_sat.data_ptr()), P(ws), ws.numel() * 4)
                return
            bwd.add("vp_bn_act_bwd_split_f32", P(x_buf), P(dy_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias),
                    P(dx_buf), P(dx_split), P(grad_of(bn_mod.weight)), P(grad_of(bn_mod.bias)), R, Cn, _ACT_RELU, 0.0, 1,
                    P(ws), ws.numel() * 4)

        # ---------------- forward ----------------
        self.x_nchw = self._buf("x_nchw", B, C, S, S)
        self.eps = self._buf("eps", B, Z)
        x_nhwc = self._buf("x_nhwc", B * S * S * C)
        if C > 1:
            fwd.add("vp_nchw_to_nhwc_f32", P(self.x_nchw), P(x_nhwc), B, C, S, S)
        else:
            x_nhwc = self.x_nchw  # identical memory order for one channel

        enc_ch = [C] + [blk.conv.weight.shape[0] for blk in enc.conv]
        sp = [S // (2 ** i) for i in range(L + 1)]
        enc16 = [use16(enc_ch[i], enc_ch[i + 1]) for i in range(L)]
        enc0_cols = x3 and C in (1, 3) and enc_ch[1] % 8 == 0
        # exact f32: the same im2col as plain fp32, both 1x1 layers on the fp32-MFMA kernels (the 3-channel implicit GEMM and the VALU
        # weight gradient took 48 + 67 us at the benchmark shard, the two 1x1 layers 36 + 40 + 20 us of im2col / re-order: 7.63 -> 7.60 ms)
        enc0_cols32 = (not x3) and C in (1, 3) and enc_ch[1] % 16 == 0
        enc_in = [x_nhwc]        # fp32 inputs (None when only the split copy exists)
        enc_in_s = [None]        # split inputs
        enc_rec = []
        for i, blk in enumerate(enc.conv):
            if i == 1 and k_pack is not None:
                fwd.wait_side(k_pack)
            Cin, Cout, Hs = enc_ch[i], enc_ch[i + 1], sp[i + 1]
            n_out = B * Hs * Hs * Cout
            c = self._buf(f"enc{i}.c", n_out)
            fl = 50.0 * B * Hs * Hs * Cin * Cout
            conv = None
            if i == 0 and enc0_cols:
                # first conv (1 or 3 image channels): im2col written once as split planes, then a 1x1 layer on the MFMA kernels
                KC = lib.vp_im2col5s2_cols(Cin)
                xcol = self._sbuf("enc0.xcol", B * Hs * Hs * KC)
                w0s = self._sbuf("enc0.w0s", Cout * KC)
                self._enc0 = (xcol, KC)
                if x2:
                    fwd.add("vp_im2col5s2_split_fmt_f32", P(self.x_nchw), P(xcol), B, Cin, S, S, 1, FMT)
                    fwd.add("vp_pack_w_im2col5_split_fmt", P(blk.conv.weight), P(w0s), Cout, Cin, FMT)
                    fwd.add("vp_conv_gather_f16", P(xcol), P(w0s), None, P(c), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, _ACT_NONE, FWD_PRODUCTS, 1.0,
                            flops=fl, tag="enc0.fwd")
                else:
                    fwd.add("vp_im2col5s2_split_f32", P(self.x_nchw), P(xcol), B, Cin, S, S, 1)
                    fwd.add("vp_pack_w_im2col5_split", P(blk.conv.weight), P(w0s), Cout, Cin)
                    fwd.add("vp_conv_gather_bf16x3", P(xcol), P(w0s), None, P(c), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, _ACT_NONE,
                            flops=fl, tag="enc0.fwd")
                p1 = None
            elif i == 0 and enc0_cols32:
                KC = lib.vp_im2col5s2_cols(Cin)
                xcol = self._buf("enc0.xcol32", B * Hs * Hs * KC)
                w0 = self._buf("enc0.w0", Cout * KC)
                self._enc0 = (xcol, KC)
                fwd.add("vp_im2col5s2_f32", P(self.x_nchw), P(xcol), B, Cin, S, S, 1)
                fwd.add("vp_pack_w_im2col5_f32", P(blk.conv.weight), P(w0), Cout, Cin)
                fwd.add("vp_conv_gather_f32", P(xcol), P(w0), None, P(c), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, _ACT_NONE, flops=fl, tag="enc0.fwd")
                p1 = None
            elif enc16[i]:
                p0 = self._sbuf(f"enc{i}.p0s", Cout * 25 * Cin)
                p1 = self._sbuf(f"enc{i}.p1s", Cin * 25 * Cout)
                pack(blk.conv.weight, p0, p1, Cout, Cin, True, first=(i == 0))
                conv = (0, "vp_conv5_gather_bf16x3", (P(enc_in_s[-1]), P(p0), P(c)), (B, Hs, Hs, Cin, Cout, 2), fl, f"enc{i}.fwd")
            else:
                p0 = self._buf(f"enc{i}.p0", Cout * 25 * Cin)
                p1 = self._buf(f"enc{i}.p1", Cin * 25 * Cout) if i > 0 else None
                pack(blk.conv.weight, p0, p1, Cout, Cin, False, first=(i == 0))
                conv = (0, "vp_conv5_gather_f32", (P(enc_in[-1]), P(p0), P(c)), (B, Hs, Hs, Cin, Cout, 2), fl, f"enc{i}.fwd")
            # the activation feeds the next conv (+ its wgrad) or, for the last block, the flatten
            nxt16 = i + 1 < L and enc16[i + 1]
            a = None if nxt16 else self._buf(f"enc{i}.a", n_out)
            a_s = self._sbuf(f"enc{i}.as", n_out) if nxt16 else None
            mean, rstd, ws = bn_block(f"enc{i}", c, B * Hs * Hs, Cout, blk.bn, a, a_s, conv=conv)
            enc_rec.append((blk, Cin, Cout, Hs, p1, c, mean, rstd, ws))
            enc_in.append(a)
            enc_in_s.append(a_s)
        if L == 1 and k_pack is not None:
            fwd.wait_side(k_pack)
        size = enc_ch[-1]
        F0 = 64 * size
        flat = self._buf("enc.flat", B * F0)
        fwd.add("vp_nhwc_to_nchw_f32", P(enc_in[-1]), P(flat), B, size, 8, 8)
        fc_lin, fc_bn = enc.fc[0], enc.fc[1]
        h = self._buf("enc.h", B * 1024)
        hb = self._buf("enc.hb", B * 1024)
        ws_fc = self._ws("enc.fc.ws", lib.vp_gemm_workspace_bytes(B, 1024, F0))
        fwd.add("vp_gemm_f32", P(flat), F0, 1, P(fc_lin.weight), F0, 1, P(h), 1024, None, B, 1024, F0, 0, P(ws_fc), ws_fc.numel() * 4)
        h_mean, h_rstd, h_ws = bn_block("enc.fc", h, B, 1024, fc_bn, hb)
        self.mu, self.logvar = self._buf("mu", B, Z), self._buf("logvar", B, Z)
        ws_mu = self._ws("enc.mu.ws", lib.vp_gemm_workspace_bytes(B, Z, 1024))
        for lin, out in ((enc.l_mu, self.mu), (enc.l_var, self.logvar)):
            fwd.add("vp_gemm_f32", P(hb), 1024, 1, P(lin.weight), 1024, 1, P(out), Z, P(lin.bias), B, Z, 1024, 0, P(ws_mu), ws_mu.numel() * 4)
        self.z, self.kl = self._buf("z", B, Z), self._buf("kl", B)
        fwd.add("vp_latent_fwd_f32", P(self.mu), P(self.logvar), P(self.eps), P(self.z), P(self.kl), B, Z)

        dfc_lin, dfc_bn = dec.fc[0], dec.fc[1]
        dsize = dec._c0
        F1 = 64 * dsize
        d = self._buf("dec.d", B * F1)
        db = self._buf("dec.db", B * F1)
        ws_dfc = self._ws("dec.fc.ws", lib.vp_gemm_workspace_bytes(B, F1, Z))
        fwd.add("vp_gemm_f32", P(self.z), Z, 1, P(dfc_lin.weight), Z, 1, P(d), F1, None, B, F1, Z, 0, P(ws_dfc), ws_dfc.numel() * 4)
        d_mean, d_rstd, d_ws = bn_block("dec.fc", d, B, F1, dfc_bn, db)
        dec_ch = [dsize] + [blk.conv.weight.shape[1] for blk in list(dec.conv)[:L]]
        dec16 = [use16(dec_ch[i], dec_ch[i + 1]) for i in range(L)]
        dn = None if dec16[0] else self._buf("dec.in", B * F1)
        dn_s = self._sbuf("dec.in_s", B * F1) if dec16[0] else None
        if x2:
            fwd.add("vp_nchw_to_nhwc_split_fmt_f32", P(db), P(dn), P(dn_s), B, dsize, 8, 8, FMT)
        else:
            fwd.add("vp_nchw_to_nhwc_split_f32", P(db), P(dn), P(dn_s), B, dsize, 8, 8)

        dec_in = [dn]
        dec_in_s = [dn_s]
        dec_rec = []
        for i in range(L):
            blk = dec.conv[i]
            Cin, Cout, Hs = dec_ch[i], dec_ch[i + 1], 8 * (2 ** i)
            n_out = B * 4 * Hs * Hs * Cout
            tbuf = self._buf(f"dec{i}.t", n_out)
            fl = 50.0 * B * Hs * Hs * Cin * Cout
            conv = None
            if dec16[i]:
                p1 = self._sbuf(f"dec{i}.p1s", Cout * 25 * Cin)   # T family: [Cbig=Cout][25][Csmall=Cin]
                p0 = self._sbuf(f"dec{i}.p0s", Cin * 25 * Cout)   # F family (dgrad): [Csmall=Cin][25][Cbig=Cout]
                pack(blk.conv.weight, p0, p1, Cin, Cout, True)
                conv = (1, "vp_conv5_scatter_bf16x3", (P(dec_in_s[-1]), P(p1), P(tbuf)), (B, Hs, Hs, Cin, Cout, 2), fl, f"dec{i}.fwd")
            else:
                p1 = self._buf(f"dec{i}.p1", Cout * 25 * Cin)
                p0 = self._buf(f"dec{i}.p0", Cin * 25 * Cout)
                pack(blk.conv.weight, p0, p1, Cin, Cout, False)
                conv = (1, "vp_conv5_scatter_f32", (P(dec_in[-1]), P(p1), P(tbuf)), (B, Hs, Hs, Cin, Cout, 2), fl, f"dec{i}.fwd")
            nxt16 = i + 1 < L and dec16[i + 1]
            # the last block feeds the final conv (fp32: its 3-channel side runs on the VALU kernels of narrow.hip;
            # measured: padding 3 -> 32 output columns for the MFMA halo kernel is LDS-read bound and 25 % slower)
            fin_halo = False
            u = None if nxt16 else self._buf(f"dec{i}.u", n_out)
            u_s = self._sbuf(f"dec{i}.us", n_out) if (nxt16 or fin_halo) else None
            mean, rstd, ws = bn_block(f"dec{i}", tbuf, B * 4 * Hs * Hs, Cout, blk.bn, u, u_s, conv=conv)
            dec_rec.append((blk, Cin, Cout, Hs, p0, tbuf, mean, rstd, ws))
            dec_in.append(u)
            dec_in_s.append(u_s)
        fin = dec.conv[L][0]
        Cf = dec_ch[-1]
        fp0 = self._buf("fin.p0", C * 25 * Cf)
        fp1 = self._buf("fin.p1", Cf * 25 * C)
        pack(fin.weight, fp0, fp1, C, Cf, False)
        xt_nhwc = self._buf("xt_nhwc", B * S * S * C)
        if dec_in_s[-1] is not None:
            fp0s = self._sbuf("fin.p0s", C * 25 * Cf)
            pack(fin.weight, fp0s, None, C, Cf, True)
            add_gather(fwd, dec_in_s[-1], fp0s, fin.bias, xt_nhwc, (B, S, S, Cf, C, 1), _ACT_SIGMOID, products=FWD_PRODUCTS,
                       flops=50.0 * B * S * S * Cf * C, tag="fin.fwd")
        elif x3 and Cf == 64 and C in (1, 3):
            # split-bf16 on the matrix cores, taps folded into the MFMA columns (the exact-f32 plan keeps the VALU kernel)
            fwd.add("vp_conv5_smallout_bf16x3", P(dec_in[-1]), P(fp0), P(fin.bias), P(xt_nhwc), B, S, S, Cf, C, _ACT_SIGMOID,
                    flops=50.0 * B * S * S * Cf * C, tag="fin.fwd")
        else:
            fwd.add("vp_conv5_gather_f32", P(dec_in[-1]), P(fp0), P(fin.bias), P(xt_nhwc), B, S, S, Cf, C, 1, _ACT_SIGMOID,
                    flops=50.0 * B * S * S * Cf * C, tag="fin.fwd")
        self.recon = self._buf("recon", 1)
        self.kl_sum = self._buf("kl_sum", 1)
        n_pix = B * S * S * C
        ws_red = self._ws("red.ws", lib.vp_reduce_workspace_bytes(n_pix))
        self._loss_num = self._buf("loss_num", 1)
        fwd.add("vp_vae_loss_f32", P(xt_nhwc), P(x_nhwc), n_pix, P(self.kl), B, P(self.recon), P(self.kl_sum), P(self._loss_num),
                1.0 / B, P(ws_red), ws_red.numel() * 4)
        self.xt_nhwc = xt_nhwc
        self.x_tilde = xt_nhwc.view(B, S, S, C).permute(0, 3, 1, 2)  # logical NCHW, channels_last memory

        # ---------------- backward ----------------
        inv_b = 1.0 / B
        dlogit = self._buf("g.dlogit", n_pix)
        # final conv's input gradient: rows-in-K kernel of edge.hip (C = 1 | 3 image channels, 64 decoder channels), else the bf16x3
        # halo kernel with dlogit padded to 8 channels
        fin_rowk = x3 and Cf == 64 and C in (1, 3)
        fin16 = x3 and Cf % 8 == 0 and C < 8 and not fin_rowk
        if fin16:
            dlogit_s = self._sbuf("g.dlogit_s", B * S * S * 8)
            fp1s = self._sbuf("fin.p1s", Cf * 25 * 8)
            pack(fin.weight, None, fp1s, C, Cf, True, 8, bf16=True)       # bf16 pairs: read by the halo kernel in every mode
            bwd.add("vp_bce_sigmoid_bwd_pad_split_f32", P(xt_nhwc), P(x_nhwc), inv_b, P(dlogit), P(dlogit_s), B * S * S, C, 8)
        else:
            bwd.add("vp_bce_sigmoid_bwd_f32", P(xt_nhwc), P(x_nhwc), inv_b, P(dlogit), n_pix)
        ws_cs = self._ws("g.colsum.ws", lib.vp_colsum_workspace_bytes(B * S * S, C))
        bwd.add("vp_colsum_f32", P(dlogit), P(grad_of(fin.bias)), B * S * S, C, P(ws_cs), ws_cs.numel() * 4, side=side_slot())
        ws_wg = self._ws("g.wgrad.ws", self._max_wgrad_ws(enc_rec, dec_rec, Cf))
        n_tapm = lib.vp_conv5_smallout_wgrad_bf16x3_workspace_bytes(B, S, S, Cf, C)      # (the exact-f32 form takes the same slabs)
        if n_tapm:      # on the matrix cores, taps folded into the MFMA rows (csrc/edge.hip); its own slab workspace
            ws_fw = self._ws("g.finwgrad.ws", n_tapm)
            bwd.add("vp_conv5_smallout_wgrad_bf16x3" if x3 else "vp_conv5_smallout_wgrad_f32", P(dec_in[-1]), P(dlogit), P(grad_of(fin.weight)), B, S, S, Cf, C, P(ws_fw),
                    ws_fw.numel() * 4, flops=50.0 * B * S * S * Cf * C, tag="fin.wgrad", side=side_slot())
        else:
            bwd.add("vp_conv5_wgrad_f32", P(dec_in[-1]), P(dlogit), P(grad_of(fin.weight)), B, S, S, Cf, C, 1, P(ws_wg), ws_wg.numel() * 4,
                    flops=50.0 * B * S * S * Cf * C, tag="fin.wgrad", side=side_slot())   # reads dlogit / dec_in[-1]: both live on
        # two ping-pong gradient buffers sized for the largest activation
        big = max([B * F0, B * F1, n_pix] + [B * 4 * r[3] * r[3] * r[2] for r in dec_rec] + [B * r[3] * r[3] * r[2] for r in enc_rec]
                  + [B * S * S * Cf])
        gA, gB = self._buf("g.A", big), self._buf("g.B", big)
        if fin_rowk:
            bwd.add("vp_conv5_smallin_dgrad_bf16x3", P(dlogit), P(fin.weight), P(gA), B, S, S, C, Cf, flops=50.0 * B * S * S * Cf * C, tag="fin.dgrad")
        elif fin16:
            bwd.add("vp_conv5_scatter_bf16x3", P(dlogit_s), P(fp1s), P(gA), B, S, S, 8, Cf, 1, flops=50.0 * B * S * S * Cf * C, tag="fin.dgrad")
        elif Cf == 64 and C in (1, 3):    # exact fp32 on the same rows-in-K tiling (csrc/edge.hip dgrad_rowk_f32_kernel)
            bwd.add("vp_conv5_smallin_dgrad_f32", P(dlogit), P(fin.weight), P(gA), B, S, S, C, Cf, flops=50.0 * B * S * S * Cf * C, tag="fin.dgrad")
        else:
            bwd.add("vp_conv5_scatter_f32", P(dlogit), P(fp1), P(gA), B, S, S, C, Cf, 1, flops=50.0 * B * S * S * Cf * C, tag="fin.dgrad")
        cur, other = gA, gB
        # split gradient (output of BN backward) for the 16-bit kernels, two buffers used alternately
        # (one buffer per layer instead of the pair -- the main stream then never waits for a side-stream weight gradient before
        # rewriting a buffer -- was measured neutral, 3.690 vs 3.691 ms: profiles/r02_notes.md section 7)
        n_gs = 2
        gS2 = [self._sbuf(f"g.S{j}", big) for j in range(n_gs)] if x3 else [None] * n_gs
        self._grad_planes = gS2 if x2 else []
        gs_last = [None] * n_gs     # side event of the weight gradient that last read each buffer
        gs_turn = [0]

        def next_gs(plan):
            k = gs_turn[0] % n_gs
            gs_turn[0] += 1
            if gs_last[k] is not None:
                plan.wait_side(gs_last[k])
            return k
        for i in range(L - 1, -1, -1):
            blk, Cin, Cout, Hs, p0, tbuf, mean, rstd, ws = dec_rec[i]
            R = B * 4 * Hs * Hs
            fl = 50.0 * B * Hs * Hs * Cin * Cout
            if dec16[i]:
                k = next_gs(bwd)
                gS = gS2[k]
                bn_block_bwd(tbuf, cur, None, R, Cout, blk.bn, mean, rstd, ws, gS)  # gS = d t_i (split)
                gs_last[k] = side_slot()
                add_wgrad(bwd, gS, dec_in_s[i], grad_of(blk.conv.weight), (B, Hs, Hs, Cout, Cin, 2), ws_wg, 1.0 / GS,
                          flops=fl, tag=f"dec{i}.wgrad", side=gs_last[k])
                add_gather(bwd, gS, p0, None, cur, (B, Hs, Hs, Cout, Cin, 2), _ACT_NONE, 1.0 / GS,
                           flops=fl, tag=f"dec{i}.dgrad")                                  # cur = d input_i
            else:
                bn_block_bwd(tbuf, cur, other, R, Cout, blk.bn, mean, rstd, ws)          # other = d t_i
                bwd.add("vp_conv5_wgrad_f32", P(other), P(dec_in[i]), P(grad_of(blk.conv.weight)), B, Hs, Hs, Cout, Cin, 2,
                        P(ws_wg), ws_wg.numel() * 4, flops=fl, tag=f"dec{i}.wgrad")
                bwd.add("vp_conv5_gather_f32", P(other), P(p0), None, P(cur), B, Hs, Hs, Cout, Cin, 2, _ACT_NONE,
                        flops=fl, tag=f"dec{i}.dgrad")                                     # cur = d input_i
        bwd.add("vp_nhwc_to_nchw_f32", P(cur), P(other), B, dsize, 8, 8)                # other = d db  (B, F1)
        bn_block_bwd(d, other, cur, B, F1, dfc_bn, d_mean, d_rstd, d_ws)                # cur = d d
        ws_g1 = self._ws("g.gemm1.ws", max(lib.vp_gemm_workspace_bytes(F1, Z, B), lib.vp_gemm_workspace_bytes(B, Z, F1),
                                            lib.vp_gemm_workspace_bytes(1024, F0, B), lib.vp_gemm_workspace_bytes(B, F0, 1024),
                                            lib.vp_gemm_workspace_bytes(Z, 1024, B), lib.vp_gemm_workspace_bytes(B, 1024, Z)))
        wsn = ws_g1.numel() * 4
        bwd.add("vp_gemm_f32", P(cur), 1, F1, P(self.z), 1, Z, P(grad_of(dfc_lin.weight)), Z, None, F1, Z, B, 2, P(ws_g1), wsn)
        dz = self._buf("g.dz", B, Z)
        bwd.add("vp_gemm_f32", P(cur), F1, 1, P(dfc_lin.weight), 1, Z, P(dz), Z, None, B, Z, F1, 1, P(ws_g1), wsn)
        # every decoder gradient is final and no decoder parameter is read any more: the data-parallel step may start
        # reducing that slice of the arena and the optimiser may update it
        self._bwd_dec = bwd
        bwd = _Plan()
        dmu, dlv = self._buf("g.dmu", B, Z), self._buf("g.dlv", B, Z)
        bwd.add("vp_latent_bwd_f32", P(self.mu), P(self.logvar), P(self.eps), P(dz), None, inv_b, P(dmu), P(dlv), B, Z)
        ws_cs2 = self._ws("g.colsum2.ws", lib.vp_colsum_workspace_bytes(B, Z))
        dhb_a, dhb_b = self._buf("g.dhb_a", B * 1024), self._buf("g.dhb_b", B * 1024)
        for lin, dsrc, dst in ((enc.l_mu, dmu, dhb_a), (enc.l_var, dlv, dhb_b)):
            bwd.add("vp_gemm_f32", P(dsrc), 1, Z, P(hb), 1, 1024, P(grad_of(lin.weight)), 1024, None, Z, 1024, B, 2, P(ws_g1), wsn)
            bwd.add("vp_colsum_f32", P(dsrc), P(grad_of(lin.bias)), B, Z, P(ws_cs2), ws_cs2.numel() * 4)
            bwd.add("vp_gemm_f32", P(dsrc), Z, 1, P(lin.weight), 1, 1024, P(dst), 1024, None, B, 1024, Z, 1, P(ws_g1), wsn)
        bwd.add("vp_add_f32", P(dhb_a), P(dhb_b), P(dhb_a), B * 1024)     # d hb = dgrad(mu head) + dgrad(logvar head)
        bwd_b = _Plan()
        self._bwd_a = bwd
        bwd = bwd_b
        dh = self._buf("g.dh", B * 1024)

        def bn_block_bwd2(x_buf, dy_buf, dx_buf, R, Cn, bn_mod, mean, rstd, ws, dx_split=None):
            if dx_split is None and R <= 64 and Cn % 4 == 0 and small_bn:
                bwd.add("vp_bn_small_bwd_f32", P(x_buf), P(dy_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias), P(dx_buf),
                        P(grad_of(bn_mod.weight)), P(grad_of(bn_mod.bias)), R, Cn, _ACT_RELU, 0.0, 1)
                return
            if x2 and dx_split is not None:      # gradient planes: fp16 pairs of GS * dx
                bwd.add("vp_bn_act_bwd_split_fmt_sat_f32", P(x_buf), P(dy_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias),
                        P(dx_buf), P(dx_split), P(grad_of(bn_mod.weight)), P(grad_of(bn_mod.bias)), R, Cn, _ACT_RELU, 0.0, 1, FMT, GS,
                        c_void_p(self._f16_sat.data_ptr()), P(ws), ws.numel() * 4)
                return
            bwd.add("vp_bn_act_bwd_split_f32", P(x_buf), P(dy_buf), P(mean), P(rstd), P(bn_mod.weight), P(bn_mod.bias),
                    P(dx_buf), P(dx_split), P(grad_of(bn_mod.weight)), P(grad_of(bn_mod.bias)), R, Cn, _ACT_RELU, 0.0, 1,
                    P(ws), ws.numel() * 4)

        bn_block_bwd2(h, dhb_a, dh, B, 1024, fc_bn, h_mean, h_rstd, h_ws)
        self._bwd_b_fc_wgrad = len(bwd.calls)        # this call is replaced by the factored exchange in multi-rank steps
        bwd.add("vp_gemm_f32", P(dh), 1, 1024, P(flat), 1, F0, P(grad_of(fc_lin.weight)), F0, None, 1024, F0, B, 2, P(ws_g1), wsn)
        self._fc_factors = (dh, flat, fc_lin.weight, F0)
        bwd.add("vp_gemm_f32", P(dh), 1024, 1, P(fc_lin.weight), 1, F0, P(gA), F0, None, B, F0, 1024, 1, P(ws_g1), wsn)
        # the encoder's dense gradients (fc.0 = 134 MB at config 3, fc.1, l_mu, l_var) are final and its dense
        # parameters are not read any more: second bucket
        self._bwd_b_dense_done = len(bwd.calls)
        bwd.add("vp_nchw_to_nhwc_f32", P(gA), P(gB), B, size, 8, 8)
        cur, other = gB, gA
        for i in range(L - 1, -1, -1):
            blk, Cin, Cout, Hs, p1, c, mean, rstd, ws = enc_rec[i]
            R = B * Hs * Hs
            fl = 50.0 * B * Hs * Hs * Cin * Cout
            if i == 0 and enc0_cols:
                xcol, KC = self._enc0
                k = next_gs(bwd)
                gS = gS2[k]
                bn_block_bwd2(c, cur, None, R, Cout, blk.bn, mean, rstd, ws, gS)  # gS = d c_0 (split)
                dwc = self._buf("enc0.dwc", Cout * KC)
                ws0 = self._ws("enc0.wgws", lib.vp_conv_wgrad_bf16x3_workspace_bytes(B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1))
                # the LAST weight gradient of the step stays on the main stream: on the side stream the join that follows it (and
                # precedes the optimiser) finds both queues idle for ~18 us -- the latency of a dependency between two hardware
                # queues (profiles/r02_notes.md) -- while here the side stream has long finished when the main stream joins it
                last_main = True
                gs_last[k] = None if last_main else side_slot()
                if x2:
                    bwd.add("vp_conv_wgrad_f16x2", P(xcol), P(gS), P(dwc), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, 1.0 / GS, P(ws0), ws0.numel() * 4,
                            flops=fl, tag="enc0.wgrad", side=gs_last[k])
                else:
                    bwd.add("vp_conv_wgrad_bf16x3", P(xcol), P(gS), P(dwc), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, P(ws0), ws0.numel() * 4,
                            flops=fl, tag="enc0.wgrad", side=gs_last[k])
                bwd.add("vp_unpack_dw_im2col5_f32", P(dwc), P(grad_of(blk.conv.weight)), Cout, Cin, side=(None if last_main else side_slot()))
            elif enc16[i]:
                k = next_gs(bwd)
                gS = gS2[k]
                bn_block_bwd2(c, cur, None, R, Cout, blk.bn, mean, rstd, ws, gS)  # gS = d c_i (split)
                gs_last[k] = side_slot()
                add_wgrad(bwd, enc_in_s[i], gS, grad_of(blk.conv.weight), (B, Hs, Hs, Cin, Cout, 2), ws_wg, 1.0 / GS,
                          flops=fl, tag=f"enc{i}.wgrad", side=gs_last[k])
                if i > 0:
                    add_scatter(bwd, gS, p1, cur, (B, Hs, Hs, Cout, Cin, 2), 1.0 / GS,
                                flops=fl, tag=f"enc{i}.dgrad")                             # cur = d a_{i-1}
                if i == max(L - 2, 1):
                    # the gradients of encoder.conv[i:] (16.4 of the 17 MB of conv parameters at config 3) are issued: third bucket
                    self._bwd_b_enc_tail, self._enc_tail_first = len(bwd.calls), i
            elif i == 0 and enc0_cols32:
                xcol, KC = self._enc0
                bn_block_bwd2(c, cur, other, R, Cout, blk.bn, mean, rstd, ws)            # other = d c_0
                dwc = self._buf("enc0.dwc", Cout * KC)
                ws0 = self._ws("enc0.wgws", lib.vp_conv_wgrad_workspace_bytes(B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1))
                bwd.add("vp_conv_wgrad_f32", P(xcol), P(other), P(dwc), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, P(ws0), ws0.numel() * 4,
                        flops=fl, tag="enc0.wgrad")
                bwd.add("vp_unpack_dw_im2col5_f32", P(dwc), P(grad_of(blk.conv.weight)), Cout, Cin)
            else:
                bn_block_bwd2(c, cur, other, R, Cout, blk.bn, mean, rstd, ws)            # other = d c_i
                # side stream only for the last layer of the walk (i == 0): nothing rewrites `other` after it
                bwd.add("vp_conv5_wgrad_f32", P(enc_in[i]), P(other), P(grad_of(blk.conv.weight)), B, Hs, Hs, Cin, Cout, 2,
                        P(ws_wg), ws_wg.numel() * 4, flops=fl, tag=f"enc{i}.wgrad", side=(side_slot() if i == 0 else None))
                if i > 0:
                    bwd.add("vp_conv5_scatter_f32", P(other), P(p1), P(cur), B, Hs, Hs, Cout, Cin, 2,
                            flops=fl, tag=f"enc{i}.dgrad")                                 # cur = d a_{i-1}
        self._pack_jobs = (_lib.PackJob * len(pack_jobs))(*pack_jobs)   # host array read by every call: keep it alive
        fwd.add_first("vp_pack_w5_batch", self._pack_jobs, len(pack_jobs), side=k_pack)
        if first_pack_jobs:
            self._pack_jobs0 = (_lib.PackJob * len(first_pack_jobs))(*first_pack_jobs)
            fwd.add_first("vp_pack_w5_batch", self._pack_jobs0, len(first_pack_jobs))
        self._fwd, self._bwd_b = fwd, bwd
        self._n_side_events = n_side[0]
        self._bn_mods = [m for m in self.vae.modules() if hasattr(m, "num_batches_tracked")]

    def _max_wgrad_ws(self, enc_rec, dec_rec, Cf) -> int:
        lib = _lib.load()
        B, S, C = self.B, self.S, self.C
        n = lib.vp_conv5_wgrad_workspace_bytes(B, S, S, Cf, C, 1)
        for blk, Cin, Cout, Hs, *_ in enc_rec:        # (the split-bf16 query may ask for more: tap pairs split the pixels deeper)
            n = max(n, lib.vp_conv5_wgrad_workspace_bytes(B, Hs, Hs, Cin, Cout, 2), lib.vp_conv5_wgrad_bf16x3_workspace_bytes(B, Hs, Hs, Cin, Cout, 2))
        for blk, Cin, Cout, Hs, *_ in dec_rec:
            n = max(n, lib.vp_conv5_wgrad_workspace_bytes(B, Hs, Hs, Cout, Cin, 2), lib.vp_conv5_wgrad_bf16x3_workspace_bytes(B, Hs, Hs, Cout, Cin, 2))
        return n

    # ---- execution ---------------------------------------------------------------------------
    def _launch_all(self, timers: Optional[dict] = None, on_decoder_grads=None, on_dense_grads=None, on_fc_wgrad=None,
                    after_forward=None, on_encoder_tail=None):
        s = torch.cuda.current_stream().cuda_stream
        # instrumented steps run the serial schedule: an event pair around a kernel that shares the GPU with another
        # stream's kernels would time the mixture, not the kernel
        side = self._side_ctx() if timers is None else None
        self._fwd.run(s, timers, side=side)
        if after_forward is not None:
            after_forward()
        self._bwd_dec.run(s, timers, side=side)
        if on_decoder_grads is not None:
            if side is not None:                         # the decoder's weight gradients are produced on the side stream
                side.flush()
                torch.cuda.current_stream().wait_stream(side[0])
            on_decoder_grads()
        self._bwd_a.run(s, timers)
        if on_fc_wgrad is not None:
            self._bwd_b.run(s, timers, 0, self._bwd_b_fc_wgrad)
            on_fc_wgrad()                                     # computes the fc.0 weight gradient from gathered factors
            self._bwd_b.run(s, timers, self._bwd_b_fc_wgrad + 1, self._bwd_b_dense_done)
        else:
            self._bwd_b.run(s, timers, 0, self._bwd_b_dense_done)
        if on_dense_grads is not None:
            on_dense_grads()
        tail = getattr(self, "_bwd_b_enc_tail", None)
        if on_encoder_tail is not None and tail is not None:
            self._bwd_b.run(s, timers, self._bwd_b_dense_done, tail, side=side)
            on_encoder_tail(side)
            self._bwd_b.run(s, timers, tail, side=side)
        else:
            self._bwd_b.run(s, timers, self._bwd_b_dense_done, side=side)
        if side is not None:
            side.flush()
            torch.cuda.current_stream().wait_stream(side[0])

    def _side_ctx(self):
        """(side stream, its events, fork event) when weight gradients run concurrently (bf16x3 plans, VP_SIDE_WGRAD != 0)."""
        if not self._n_side_events or not self._side_wgrad:
            return None
        if not hasattr(self, "_side"):
            self._side = _SideCtx(self._n_side_events)
        return self._side

    def forward_backward(self, x: torch.Tensor, eps: torch.Tensor, timers: Optional[dict] = None, on_decoder_grads=None,
                         on_dense_grads=None, on_fc_wgrad=None, after_forward=None, on_encoder_tail=None):
        """Gradients of (BCE_sum + KL_sum)/B land in the optimiser's flat gradient arena.
        Returns (loss, recon, kl) as device scalars (no host sync).  They are the plan's static output buffers (like the
        outputs of a captured graph): the next step overwrites them, so read or copy them before stepping again.  ``timers`` =
        {"names": set of entry points, "events": []} brackets those launches with HIP events
        (eager mode only)."""
        self._bind_inputs(x, eps)
        if self._graph is not None and timers is None and on_decoder_grads is None and on_dense_grads is None \
                and on_fc_wgrad is None and after_forward is None and on_encoder_tail is None:
            self._graph.replay()
        else:
            self._launch_all(timers, on_decoder_grads, on_dense_grads, on_fc_wgrad, after_forward, on_encoder_tail)
        # the plan wrote every gradient into the arena: a ``.grad`` left None by zero_grad(set_to_none=True) must not read as
        # "no gradient" in the optimiser's gather_grads()
        self.opt.arena.adopt_views()
        # BatchNorm num_batches_tracked is advanced lazily in sync_counters()
        self._steps_since_sync = getattr(self, "_steps_since_sync", 0) + 1
        return self._loss_num, self.recon, self.kl_sum     # (loss per image, recon sum, KL sum): device scalars of the plan

    def _bind_inputs(self, x: torch.Tensor, eps: torch.Tensor) -> None:
        """Point the launches that consume the batch at the caller's tensors (fp32, contiguous, on this device, right shape:
        the three-channel transpose reads ``x``, the two latent kernels read ``eps``); anything else -- and the hipGraph
        replay, whose node arguments are frozen -- is copied into the plan's static buffers."""
        if not hasattr(self, "_in_slots"):
            self._in_slots = {"x": [], "eps": []}
            statics = {"x": self.x_nchw.data_ptr(), "eps": self.eps.data_ptr()}
            if self.C != 1:                  # one channel: x_nchw doubles as the NHWC activation of the whole plan
                for plan in (self._fwd, self._bwd_dec, self._bwd_a, self._bwd_b):
                    for call in plan.calls:
                        for i, a in enumerate(call[2]):
                            if isinstance(a, c_void_p):
                                for k, v in statics.items():
                                    if a.value == v:
                                        self._in_slots[k].append((call[2], i))
        for key, t, static in (("x", x, self.x_nchw), ("eps", eps, self.eps)):
            if tuple(t.shape) != tuple(static.shape):
                # (``static.copy_(t)`` would broadcast a (1, C, H, W) batch silently; a DataLoader's short last batch -- no
                # drop_last -- would raise a bare torch broadcast error)
                raise _lib.VaePlayHipError(
                    f"FusedVAEStep: {key} has shape {tuple(t.shape)}, this plan was built for {tuple(static.shape)} "
                    f"(batch {self.B}; build a second FusedVAEStep over the same optimiser for another batch size)")
            slots = self._in_slots[key]
            direct = (bool(slots) and self._graph is None and t.device == static.device and t.dtype == torch.float32
                      and t.is_contiguous() and t.shape == static.shape)
            if not direct:
                static.copy_(t, non_blocking=True)
                t = static
            for args, i in slots:
                args[i] = c_void_p(t.data_ptr())

    def f16_saturated(self) -> int:
        """precision="f16x2" diagnostics (a host sync; not on the step's path): how many elements of the fp16 gradient planes sit at
        +-65504, i.e. |g| * grad_scale16 overflowed fp16's range in the last step and was clamped.  The two ping-pong buffers hold
        the last two layers' planes (earlier layers were overwritten): a non-zero count means grad_scale16 is too large for this
        model / loss scale -- lower it (powers of two; 4096 leaves |g| < 16)."""
        n = 0
        for t in getattr(self, "_grad_planes", []):
            hi = t[0].view(torch.float16)
            n += int((hi.abs() >= 65504.0).sum().item())
        return n

    def sync_counters(self):
        """Advance BatchNorm ``num_batches_tracked`` buffers (bookkeeping only; kept off the hot path).  precision="f16x2": also reads
        the sticky saturation flag of the fp16 gradient planes (a host sync, like everything here) and raises when a step since the
        last call clamped a gradient at +-65504 / grad_scale16 -- the weights of those steps are off; lower ``grad_scale16``."""
        n = getattr(self, "_steps_since_sync", 0)
        if n:
            for m in self._bn_mods:
                m.num_batches_tracked.add_(n)
            self._steps_since_sync = 0
        if self.precision == "f16x2" and int(self._f16_sat.item()):
            self._f16_sat.zero_()
            raise _lib.VaePlayHipError(
                f"precision='f16x2': gradient planes saturated fp16's range since the last sync_counters() (|g| * grad_scale16 > 65504 "
                f"with grad_scale16 = {self.grad_scale16:g}); build the step with a smaller power of two")

    def _decoder_slice_start(self) -> int:
        """Arena offset of the first decoder parameter (encoder parameters precede it in ``vae.parameters()``)."""
        dec_ids = {id(p) for p in self.vae.decoder.parameters()}
        offs = [o for p, o in zip(self.opt.arena.params, self.opt.arena.offsets) if id(p) in dec_ids]
        first = min(offs)
        assert all(o >= first for o in offs) and all(
            (id(p) in dec_ids) == (o >= first) for p, o in zip(self.opt.arena.params, self.opt.arena.offsets)), \
            "decoder parameters must form the tail of the flat arena"
        return first

    def _encoder_dense_start(self) -> int:
        """Arena offset of ``encoder.fc.0.weight``: the encoder's conv/BN parameters precede it, its dense
        parameters (fc.0, fc.1, l_mu, l_var) run from there to the decoder slice."""
        a = self.opt.arena
        off = {id(p): o for p, o in zip(a.params, a.offsets)}
        first = off[id(self.vae.encoder.fc[0].weight)]
        conv_ids = {id(p) for p in self.vae.encoder.conv.parameters()}
        assert all((id(p) in conv_ids) == (o < first) for p, o in zip(a.params, a.offsets) if o < self._decoder_slice_start()), \
            "encoder conv parameters must form the head of the flat arena"
        return first

    def step(self, x: torch.Tensor, eps: torch.Tensor, timers: Optional[dict] = None, overlap: bool = True, comm_trace=None):
        """One full training step: fwd + loss + bwd, SUM all-reduce of the flat gradient arena, fused update.

        With several ranks the arena is reduced as four buckets of the same flat buffer, each handed to RCCL as
        soon as its gradients are final so that the all-reduce runs on the communicator's stream underneath the rest
        of backward: the decoder slice after the decoder's backward, the encoder's dense slice (fc.0 is 134 MB of the
        213 MB at config 3) after its weight gradient, the deep encoder blocks' conv slice (16.4 MB) while the shallow blocks
        still run backward, and the rest of the encoder's conv slice (< 1 MB) after backward; the
        optimiser kernel waits for all of them.  ``overlap=False`` issues one all-reduce of the whole arena.
        ``comm_trace`` (parallel.CommTrace, instrumented steps of bench.py): every collective is issued through it, which
        records its bytes, its device time and the time the main stream waits for collectives.
        (Updating each slice right after its all-reduce, on a second side stream underneath the rest of backward, was
        measured and is NOT done: the HBM-bound optimiser kernel slows the concurrent kernels by more than it hides,
        4.52 vs 4.46 ms/step on one GPU; ``optim.*.step_range`` remains available.)"""
        tr = comm_trace

        def reduce(name, t):         # SUM all-reduce of one bucket, issued now
            if tr is not None:
                return tr.issue(name, "all_reduce", t.numel() * 4, lambda: parallel.allreduce_flat_grads(t, self.group, async_op=True))
            return parallel.allreduce_flat_grads(t, self.group, async_op=True)

        def gather(name, out_all, local):
            if tr is not None:
                return tr.issue(name, "all_gather", out_all.numel() * 4,
                                lambda: parallel.allgather_rows(out_all, local, self.group, async_op=True))
            return parallel.allgather_rows(out_all, local, self.group, async_op=True)

        def wait_all(works):
            if tr is not None:
                tr.wait()
                return
            for w in works:
                if w is not None:
                    w.wait()

        if parallel.dp_active(self.group) and overlap:
            g = self.opt.flat_grad
            cut = self._decoder_slice_start()
            dense = self._encoder_dense_start()
            works = []
            factored = self._dp_factored
            # fourth cut: encoder.conv[i:] for the deepest blocks -- almost all of the encoder's conv parameters -- is reduced
            # while the remaining shallow blocks still run backward, so that only their < 1 MB is left for the exposed
            # all-reduce at the end of the step.  Its weight gradients are produced on the side stream and its BatchNorm
            # gradients on the main stream: the collective is issued from the side stream after it has joined the main one.
            tail_lo = dense
            if getattr(self, "_bwd_b_enc_tail", None) is not None and self._dp_enc_tail:
                off = {id(p): o for p, o in zip(self.opt.arena.params, self.opt.arena.offsets)}
                tail_lo = off[id(self.vae.encoder.conv[self._enc_tail_first].conv.weight)]

            def enc_tail(side):
                if tail_lo >= dense:
                    return
                if side is not None:
                    side.flush()
                    side[0].wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side[0]):
                        works.append(reduce("encoder.conv[deep]", g[tail_lo:dense]))
                else:
                    works.append(reduce("encoder.conv[deep]", g[tail_lo:dense]))
            if factored:
                # fc.0's weight gradient (134 MB of the 213 MB at config 3) is dW = dh^T flat, a sum of B outer products
                # per rank: exchange the two factors (W x 4.3 MB all-gather) and contract over all W*B rows locally
                # instead of all-reducing the weight-sized result.  Identical to the sum of the ranks' gradients.
                dh, flat, fcw, F0 = self._fc_factors
                W, B = self.world, self.B
                if not hasattr(self, "_fc_all"):
                    self._fc_all = (torch.empty((W * B, 1024), device=self.dev), torch.empty((W * B, F0), device=self.dev))
                dh_all, flat_all = self._fc_all
                gathers = []
                fc_lo = self.opt.arena.offsets[[id(p) for p in self.opt.arena.params].index(id(fcw))]
                fc_hi = fc_lo + (fcw.numel() + 63) // 64 * 64

                def after_fwd():
                    gathers.append(gather("fc.0 factor: flat", flat_all, flat.view(B, F0)))

                def fc_wgrad():
                    gathers.append(gather("fc.0 factor: dh", dh_all, dh.view(B, 1024)))
                    wait_all(gathers)
                    ops.gemm(dh_all, 1, 1024, flat_all, 1, F0, 1024, F0, W * B, 2, out=self.opt.arena.grad_view(fcw).view(1024, F0))

                def dense_bucket():
                    assert fc_lo == dense, "fc.0.weight must open the encoder's dense slice"
                    works.append(reduce("encoder dense (without fc.0)", g[fc_hi:cut]))

                out = self.forward_backward(
                    x, eps, timers, after_forward=after_fwd, on_fc_wgrad=fc_wgrad, on_dense_grads=dense_bucket, on_encoder_tail=enc_tail,
                    on_decoder_grads=lambda: works.append(reduce("decoder", g[cut:])))
            else:
                out = self.forward_backward(
                    x, eps, timers, on_encoder_tail=enc_tail,
                    on_decoder_grads=lambda: works.append(reduce("decoder", g[cut:])),
                    on_dense_grads=lambda: works.append(reduce("encoder dense", g[dense:cut])))
            works.append(reduce("encoder.conv[shallow]" if tail_lo < dense else "encoder.conv", g[:tail_lo]))
            wait_all(works)
        elif self.world == 1 and self._graph is None and self._outer_adam():      # (a captured graph replays the materialising plan)
            # one rank: encoder.fc.0's weight gradient (134 MB at config 3) is contracted from its two factors inside the Adam
            # kernel instead of being written by a GEMM and read back by the update; fc.0.weight.grad is NOT written by step()
            # (forward_backward() alone materialises every gradient).
            # ... and that update runs on the side stream as soon as its factors are final and the dense input gradient has read the
            # weight for the last time in this step: 134 us of HBM-bound work underneath the encoder's MFMA-bound convolution backward
            # instead of behind it (VP_ADAM_OUTER_EARLY=0: at the end of the step; 1: fc.0 only; 2, the default: fc.0 and the arena
            # slice behind it.  3.640 / 3.609 / 3.572 ms in one process, tools/ab_env.py)
            early = None
            if timers is None and self._adam_outer_early:
                def early():
                    side = self._side_ctx()
                    if side is None:
                        return
                    side.flush()
                    side.fork.record()
                    side.stream.wait_event(side.fork)
                    with torch.cuda.stream(side.stream):
                        # (mode 2: also the arena slice behind fc.0 -- the rest of the encoder's dense layers and the whole decoder,
                        # whose weight gradients precede this launch on the side stream and whose other gradients the fork covers)
                        self.opt.step_outer_early(with_tail=(self._adam_outer_early == 2 and getattr(self, "_early_tail_ok", False)))
            out = self.forward_backward(x, eps, timers, on_fc_wgrad=_noop, on_dense_grads=early)
            self.opt.step(outer=True)
            return out
        else:
            out = self.forward_backward(x, eps, timers)
            if parallel.dp_active(self.group):
                wait_all([reduce("whole gradient arena", self.opt.flat_grad)])
        if tr is not None:
            tr.end_step()
        self.opt.step()
        return out

    def _outer_adam(self) -> bool:
        if not self._adam_outer or not hasattr(self.opt, "set_outer_grad"):
            return False
        # the factors are THIS engine's static buffers: bind them on every call in which the optimiser holds another engine's
        # (a second FusedVAEStep over the same optimiser -- another batch size, precision or a rebuilt plan -- would otherwise
        # have its fc.0 update contracted from the first engine's stale buffers)
        dh, flat, fcw, F0 = self._fc_factors
        cur = getattr(self.opt, "_outer", None)
        if cur is None or cur[4].data_ptr() != dh.data_ptr() or cur[5].data_ptr() != flat.data_ptr() or cur[4].shape[0] != self.B:
            self.opt.set_outer_grad(fcw, dh.view(self.B, 1024), flat.view(self.B, F0))
        if not hasattr(self, "_early_tail_ok"):
            # the early update of the arena slice BEHIND fc.0 (step()) relies on the arena order of vae.parameters(): encoder conv
            # blocks, fc.0, the encoder's other dense layers, the decoder
            try:
                self._early_tail_ok = self._encoder_dense_start() == self.opt._outer[0]
            except AssertionError:
                self._early_tail_ok = False
        return True

    def capture(self, warmup: int = 2):
        """Capture forward+backward into a hipGraph (torch.cuda.CUDAGraph) and replay it from then on."""
        self._bind_inputs(self.x_nchw, self.eps)      # the graph's nodes must read the static buffers
        # warm-up and capture execute the step: keep the BatchNorm running buffers unchanged by them
        saved = [(m, m.running_mean.clone(), m.running_var.clone()) for m in self._bn_mods]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._launch_all()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        ops.CAPTURE_OK[0] = True          # (ops._stream() refuses captures it does not know: the autograd front end's)
        try:
            with torch.cuda.graph(g):
                self._launch_all()
        finally:
            ops.CAPTURE_OK[0] = False
        self._graph = g
        for m, rm, rv in saved:
            m.running_mean.copy_(rm)
            m.running_var.copy_(rv)
        return g
