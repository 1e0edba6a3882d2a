"""Fused VAE-GAN training step (BASELINE config 4): the loop body of train.py:43-78 -- VaeGan.forward (models/networks.py:233-248),
VaeGan.loss (:264-281), the five losses of train.py:61-66, their five ``backward(retain_graph=True)`` calls and the four RMSprop
updates -- as ONE pre-planned sequence of HIP kernel launches on the ``_Plan`` machinery of engine.py: no autograd graph, no
per-step allocation, no ATen arithmetic on the path.

What the plan does differently from the autograd front end (functional.py), with identical arithmetic per kernel:
  * ONE backward traversal: train.py:69-73 accumulates the five losses' gradients into the same ``.grad`` tensors, i.e. every
    parameter receives the gradient of  recon + sum(kl) + (1 + lambda) sum(mse) + (1 - (1 - lambda)) loss_discriminator + l1;
    the two coefficients are formed in fp32 exactly as autograd's accumulation forms them;
  * ONE pass over the discriminator's conv stack for its "REC" and "GAN" calls (models/networks.py:244-245); the second
    running-statistics update of the blocks both calls run is replayed from the buffers (as Discriminator.forward_rec_and_gan);
  * convolution weights packed once per step for both decoder passes (z and z_p) by one batched launch on the side stream;
  * BatchNorm statistics from the convolution's epilogue, split-bf16 operand planes written by the BatchNorm kernels;
  * weight gradients on a side stream underneath the HBM-bound BatchNorm / elementwise kernels of the main chain;
  * every gradient written once, straight into the four optimisers' flat arenas (the decoder's second pass into a shadow arena
    that ONE add folds in); the arenas are then all-reduced (several ranks) and consumed by the fused RMSprop kernels.

Arithmetic: the 5x5 convolutions with channel counts that are multiples of 8 on the split-bf16 kernels ("bf16x3"), the image-side
convolutions on the edge MFMA kernels, dense layers on the exact-fp32 kernels -- the same kernels ``set_conv_precision("bf16x3")``
selects for the drop-in modules, so the step equals the autograd path (tests/test_gpu_engine_gan.py).
"""
from __future__ import annotations

import os
from ctypes import c_void_p
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, ops, parallel
from .engine import _Plan, _SideCtx, _ptr
from .networks import VaeGan

_ACT_RELU, _ACT_NONE, _ACT_SIGMOID = ops.ACT_RELU, ops.ACT_NONE, ops.ACT_SIGMOID


class FusedVAEGANStep:
    """forward + losses + backward (+ all-reduce + the four RMSprop updates) for a ``networks.VaeGan``.

    ``optimizers`` = flat-arena optimisers (vae_play_amd.optim) over net.encoder / net.decoder / net.discriminator /
    net.param_encoder parameters, in any order (train.py:212-219 builds exactly these four); every parameter must live in one
    of their arenas, the decoder's in a single one."""

    def __init__(self, net: VaeGan, optimizers, batch_size: int, img_size: int, lambda_mse: float = 1e-6, group=None,
                 _plan_only: bool = False):
        self.net, self.opts, self.B, self.S = net, list(optimizers), int(batch_size), int(img_size)
        self.Z, self.L = net.z_size, net.iter_level
        self.lam = float(lambda_mse)
        self.group = group
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
        for o in self.opts:
            o.grad_scale = 1.0 / self.world
        dev = next(net.parameters()).device
        if dev.type != "cuda" and not _plan_only:      # (_plan_only: build the launch list over host buffers, for structural tests)
            raise _lib.VaePlayHipError("FusedVAEGANStep needs the model on the HIP device")
        if self.S != 8 * 2 ** self.L:
            raise ValueError("img_size must be 8 * 2**iter_level")
        if net.discriminator.recon_levl != len(net.discriminator.conv) - 1:
            raise NotImplementedError("the fused plan taps the discriminator's last block (VaeGan's own construction)")
        for p in net.parameters():
            if getattr(p, "_vp_arena", None) is None:
                raise _lib.VaePlayHipError("every VaeGan parameter must live in a flat-arena optimiser")
        self.dev = dev
        self._bufs: Dict[str, torch.Tensor] = {}
        self._steps_since_sync = 0
        self._build()
        import weakref
        me = weakref.ref(self)

        def _sync(module, prefix, keep_vars):
            o = me()
            if o is not None:
                o.sync_counters()
        self._sd_hook = net.register_state_dict_pre_hook(_sync)

    # ---- buffers ----------------------------------------------------------------------------
    def _buf(self, name: str, *shape) -> torch.Tensor:
        t = torch.empty(shape, dtype=torch.float32, device=self.dev)
        assert name not in self._bufs, name
        self._bufs[name] = t
        return t

    def _ws(self, name: str, nbytes: int) -> torch.Tensor:
        return self._buf(name, max(4, (int(nbytes) + 3) // 4))

    def _sbuf(self, name: str, n: int) -> torch.Tensor:
        t = torch.empty((2, n), dtype=torch.int16, device=self.dev)
        assert name not in self._bufs, name
        self._bufs[name] = t
        return t

    # ---- plan construction ------------------------------------------------------------------
    def _build(self):
        lib = _lib.load()
        B, S, Z, L = self.B, self.S, self.Z, self.L
        net = self.net
        enc, dec, disc, pe = net.encoder, net.decoder, net.discriminator, net.param_encoder
        P = _ptr
        fwd, fwd_disc, bwd = _Plan(), _Plan(), _Plan()
        n3 = 3 * B
        n_pix = B * S * S
        n_side = [0]
        fuse_stats = os.environ.get("VP_FUSE_BN_STATS", "1") != "0"
        side_wgrad = os.environ.get("VP_SIDE_WGRAD", "1") != "0"
        self._bn_counts: List[tuple] = []          # (BatchNormAct module, forward passes per step)

        def side_slot():
            if not side_wgrad:
                return None
            n_side[0] += 1
            return n_side[0] - 1

        def at(t: torch.Tensor, off_elems: int):
            return c_void_p(t.data_ptr() + 4 * off_elems)

        def grad_of(p):
            return p._vp_arena.grad_view(p)

        dec_arena = next(dec.parameters())._vp_arena
        if any(p._vp_arena is not dec_arena for p in dec.parameters()):
            raise _lib.VaePlayHipError("the decoder's parameters must live in one arena")
        self._dec_arena = dec_arena
        self._dec_shadow = torch.zeros_like(dec_arena.flat_grad)
        # optimisers whose arena holds exactly one sub-network's parameters (train.py:212-219 builds one optimiser per sub-network):
        # candidates for an early update; an arena shared between sub-networks is updated at the end of the step
        self._arena_opts = {"disc": [], "dec": []}
        for kind, mod in (("disc", net.discriminator), ("dec", dec)):
            ids = {id(p) for p in mod.parameters()}
            for o in self.opts:
                if o.arena.numel and {id(p) for p in o.arena.params} == ids and not getattr(o.arena, "foreign", None):
                    self._arena_opts[kind].append(o)
        self._early_done = set()
        # the encoder's arena: conv blocks first, then fc.0 / fc.1 / l_mu / l_var (Encoder's parameter order); the dense slice -- 90 % of
        # it -- can be updated as soon as the dense backward is through, the conv slice only at the end of the step
        self._enc_dense = []
        enc_ids = {id(p) for p in enc.parameters()}
        conv_ids = {id(p) for p in enc.conv.parameters()}
        for o in self.opts:
            a = o.arena
            if a.numel and {id(p) for p in a.params} == enc_ids and not getattr(a, "foreign", None):
                lo = min(off for p, off in zip(a.params, a.offsets) if id(p) not in conv_ids)
                if all((id(p) in conv_ids) == (off < lo) for p, off in zip(a.params, a.offsets)):
                    self._enc_dense.append((o, lo))
        self._early_partial = {}

        def grad2_of(p):         # second decoder pass: same offsets in the shadow arena
            return self._dec_shadow[p._vp_off:p._vp_off + p.numel()].view_as(p)

        # dense layers share one workspace (all on the main stream); its size is known once every call is planned
        gemm_calls, gemm_need = [], [0]

        def gemm(plan, A, sam, sak, Bm, sbn, sbk, C, ldc, bias, M, N, K, mode):
            gemm_need[0] = max(gemm_need[0], lib.vp_gemm_workspace_bytes(M, N, K))
            plan.add("vp_gemm_f32", A, sam, sak, Bm, sbn, sbk, C, ldc, bias, M, N, K, mode, None, 0)
            gemm_calls.append(plan.calls[-1][2])

        def lin_fwd(plan, x, W, bias, y, M, N, K):            # y[M,N] = x[M,K] W[N,K]^T + bias
            gemm(plan, P(x), K, 1, P(W), K, 1, P(y), N, P(bias), M, N, K, 0)

        def lin_dgrad(plan, dy, W, dx, M, N, K):              # dx[M,K] = dy[M,N] W[N,K]
            gemm(plan, P(dy), N, 1, P(W), 1, K, P(dx), K, None, M, K, N, 1)

        def lin_wgrad(plan, dy, x, dW, M, N, K):              # dW[N,K] = dy[M,N]^T x[M,K]
            gemm(plan, P(dy), 1, N, P(x), 1, K, P(dW), K, None, N, K, M, 2)

        def colsum(plan, tag, x, out, R, C, side=None):
            ws = self._ws(f"{tag}.csws", lib.vp_colsum_workspace_bytes(R, C))
            plan.add("vp_colsum_f32", P(x), P(out), R, C, P(ws), ws.numel() * 4, side=side)

        pack_jobs = []

        def pack(weight, p0, p1, Cs, Cb, split):
            pack_jobs.append(_lib.PackJob(weight.data_ptr(), p0.data_ptr() if p0 is not None else None,
                                          p1.data_ptr() if p1 is not None else None, Cs, Cb, 0, 1 if split else 0))

        k_pack = side_slot()

        # ---- BatchNorm helpers -------------------------------------------------------------------------------------------
        def conv_bn_fwd(plan, tag, family, lead, geom, fl, R, Cn, bn, y, y_s, count):
            """split-bf16 convolution (family 0 = gather / nn.Conv2d, 1 = scatter / nn.ConvTranspose2d) + batch statistics (from
            the convolution's epilogue where the launch shape allows) + normalise / ReLU writing fp32 and / or split planes"""
            name = "vp_conv5_gather_bf16x3" if family == 0 else "vp_conv5_scatter_bf16x3"
            mean, rstd = self._buf(f"{tag}.mean", Cn), self._buf(f"{tag}.rstd", Cn)
            ws = self._ws(f"{tag}.bnws", lib.vp_bn_workspace_bytes(R, Cn))
            mom, eps = float(bn.momentum), float(bn.eps)
            self._bn_counts.append((bn, count))
            qgeom = geom if family == 0 else (geom[0], geom[1], geom[2], geom[4], geom[3], geom[5])
            nst = lib.vp_conv5_stats_workspace_bytes(family, *qgeom) if fuse_stats else 0
            if nst:
                st = self._ws(f"{tag}.statws", nst)
                plan.add(name.replace("_bf16x3", "_stats_bf16x3"), *lead, *geom, eps, mom, P(mean), P(rstd), P(bn.running_mean),
                         P(bn.running_var), P(st), st.numel() * 4, flops=fl, tag=f"{tag}.fwd")
            else:
                if family == 0:
                    plan.add(name, lead[0], lead[1], None, lead[2], *geom, _ACT_NONE, flops=fl, tag=f"{tag}.fwd")
                else:
                    plan.add(name, lead[0], lead[1], lead[2], *geom, flops=fl, tag=f"{tag}.fwd")
                plan.add("vp_bn_stats_f32", lead[2], R, Cn, eps, mom, P(mean), P(rstd), P(bn.running_mean), P(bn.running_var),
                         P(ws), ws.numel() * 4)
            plan.add("vp_bn_act_fwd_split_f32", lead[2], P(mean), P(rstd), P(bn.weight), P(bn.bias), P(y), P(y_s), R, Cn, _ACT_RELU, 0.0)
            return mean, rstd, ws

        def bn_plain_fwd(plan, tag, x, R, Cn, bn, y, y_s, count):
            """statistics + normalise / ReLU of an fp32 [R][Cn] buffer (dense layers: one launch when R <= 64)"""
            mean, rstd = self._buf(f"{tag}.mean", Cn), self._buf(f"{tag}.rstd", Cn)
            mom, eps = float(bn.momentum), float(bn.eps)
            self._bn_counts.append((bn, count))
            if y_s is None and R <= 64 and Cn % 4 == 0:
                plan.add("vp_bn_small_fwd_f32", P(x), R, Cn, eps, mom, P(bn.weight), P(bn.bias), P(mean), P(rstd), P(bn.running_mean),
                         P(bn.running_var), P(y), _ACT_RELU, 0.0)
                return mean, rstd, None
            ws = self._ws(f"{tag}.bnws", lib.vp_bn_workspace_bytes(R, Cn))
            plan.add("vp_bn_stats_f32", P(x), R, Cn, eps, mom, P(mean), P(rstd), P(bn.running_mean), P(bn.running_var), P(ws), ws.numel() * 4)
            plan.add("vp_bn_act_fwd_split_f32", P(x), P(mean), P(rstd), P(bn.weight), P(bn.bias), P(y), P(y_s), R, Cn, _ACT_RELU, 0.0)
            return mean, rstd, ws

        def bn_bwd(x, dy, dx, dx_s, R, Cn, bn, mean, rstd, ws, gfn):
            if ws is None:
                bwd.add("vp_bn_small_bwd_f32", P(x), P(dy), P(mean), P(rstd), P(bn.weight), P(bn.bias), P(dx), P(gfn(bn.weight)),
                        P(gfn(bn.bias)), R, Cn, _ACT_RELU, 0.0, 1)
            else:
                bwd.add("vp_bn_act_bwd_split_f32", P(x), P(dy), P(mean), P(rstd), P(bn.weight), P(bn.bias), P(dx), P(dx_s),
                        P(gfn(bn.weight)), P(gfn(bn.bias)), R, Cn, _ACT_RELU, 0.0, 1, P(ws), ws.numel() * 4)

        # =================================================== forward ===================================================
        self.xcat = self._buf("xcat", n3, 1, S, S)        # (original | reconstructed | sampled): the discriminator's batch
        self.eps, self.z_p = self._buf("eps", B, Z), self._buf("z_p", B, Z)
        self.targets = self._buf("targets", B, 3)
        xcat = self.xcat

        # ---- encoder (models/networks.py:49-78) ----
        enc_ch = [1] + [blk.conv.weight.shape[0] for blk in enc.conv]
        enc_rec = []
        in_s = None
        for i, blk in enumerate(enc.conv):
            if i == 1 and k_pack is not None:
                fwd.wait_side(k_pack)
            Cin, Cout, Hs = enc_ch[i], enc_ch[i + 1], S >> (i + 1)
            n_out = B * Hs * Hs * Cout
            c = self._buf(f"enc{i}.c", n_out)
            fl = 50.0 * B * Hs * Hs * Cin * Cout
            last = i == L - 1
            a = self._buf(f"enc{i}.a", n_out) if last else None
            a_s = None if last else self._sbuf(f"enc{i}.as", n_out)
            if i == 0:
                if Cout % 8:
                    raise NotImplementedError("encoder width must be a multiple of 8")
                KC = lib.vp_im2col5s2_cols(1)
                xcol = self._sbuf("enc0.xcol", B * Hs * Hs * KC)
                w0s = self._sbuf("enc0.w0s", Cout * KC)
                fwd.add("vp_im2col5s2_split_f32", P(xcat), P(xcol), B, 1, S, S, 1)
                fwd.add("vp_pack_w_im2col5_split", P(blk.conv.weight), P(w0s), Cout, 1)
                fwd.add("vp_conv_gather_bf16x3", P(xcol), P(w0s), None, P(c), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, _ACT_NONE,
                        flops=fl, tag="enc0.fwd")
                mean, rstd, ws = bn_plain_fwd(fwd, "enc0", c, B * Hs * Hs, Cout, blk.bn, a, a_s, 1)
                enc_rec.append((blk, Cin, Cout, Hs, None, c, mean, rstd, ws, (xcol, KC)))
            else:
                p0 = self._sbuf(f"enc{i}.p0s", Cout * 25 * Cin)
                p1 = self._sbuf(f"enc{i}.p1s", Cin * 25 * Cout)
                pack(blk.conv.weight, p0, p1, Cout, Cin, True)
                mean, rstd, ws = conv_bn_fwd(fwd, f"enc{i}", 0, (P(in_s), P(p0), P(c)), (B, Hs, Hs, Cin, Cout, 2), fl, B * Hs * Hs, Cout,
                                             blk.bn, a, a_s, 1)
                enc_rec.append((blk, Cin, Cout, Hs, p1, c, mean, rstd, ws, in_s))
            enc_last, in_s = a, a_s
        if L == 1 and k_pack is not None:
            fwd.wait_side(k_pack)
        size = enc_ch[-1]
        F0 = 64 * size
        flat = self._buf("enc.flat", B * F0)
        fwd.add("vp_nhwc_to_nchw_f32", P(enc_last), P(flat), B, size, 8, 8)
        fc_lin, fc_bn = enc.fc[0], enc.fc[1]
        H1 = fc_lin.weight.shape[0]
        h, hb = self._buf("enc.h", B * H1), self._buf("enc.hb", B * H1)
        lin_fwd(fwd, flat, fc_lin.weight, None, h, B, H1, F0)
        h_mean, h_rstd, h_ws = bn_plain_fwd(fwd, "enc.fc", h, B, H1, fc_bn, hb, None, 1)
        self.mu, self.logvar = self._buf("mu", B, Z), self._buf("logvar", B, Z)
        for lin, out in ((enc.l_mu, self.mu), (enc.l_var, self.logvar)):
            lin_fwd(fwd, hb, lin.weight, lin.bias, out, B, Z, H1)
        self.z, self.kl = self._buf("z", B, Z), self._buf("kl", B)
        fwd.add("vp_latent_fwd_f32", P(self.mu), P(self.logvar), P(self.eps), P(self.z), P(self.kl), B, Z)

        # ---- decoder (models/networks.py:81-112), run on z and on z_p with the same packed weights ----
        dfc_lin, dfc_bn = dec.fc[0], dec.fc[1]
        dsize = dec._c0
        F1 = 64 * dsize
        dec_ch = [dsize] + [blk.conv.weight.shape[1] for blk in list(dec.conv)[:L]]
        if any(ch % 8 for ch in dec_ch):
            raise NotImplementedError("decoder widths must be multiples of 8")
        dec_pk = []
        for i in range(L):
            blk = dec.conv[i]
            Cin, Cout = dec_ch[i], dec_ch[i + 1]
            p1 = self._sbuf(f"dec{i}.p1s", Cout * 25 * Cin)      # T family: [Cbig = Cout][25][Csmall = Cin]
            p0 = self._sbuf(f"dec{i}.p0s", Cin * 25 * Cout)      # F family (input gradient): [Csmall = Cin][25][Cbig = Cout]
            pack(blk.conv.weight, p0, p1, Cin, Cout, True)
            dec_pk.append((p0, p1))
        fin = dec.conv[L][0]
        Cf = dec_ch[-1]
        fin_edge = Cf == 64
        fp0 = self._buf("fin.p0", 25 * Cf)
        fp1 = None if fin_edge else self._buf("fin.p1", Cf * 25)
        pack(fin.weight, fp0, fp1, 1, Cf, False)

        def dec_fwd(tag, zbuf, out_ptr):
            d, db = self._buf(f"{tag}.d", B * F1), self._buf(f"{tag}.db", B * F1)
            lin_fwd(fwd, zbuf, dfc_lin.weight, None, d, B, F1, Z)
            fc_rec = bn_plain_fwd(fwd, f"{tag}.fc", d, B, F1, dfc_bn, db, None, 1)
            dn_s = self._sbuf(f"{tag}.in_s", B * F1)
            fwd.add("vp_nchw_to_nhwc_split_f32", P(db), None, P(dn_s), B, dsize, 8, 8)
            cur_s, recs, u = dn_s, [], None
            for i in range(L):
                blk = dec.conv[i]
                Cin, Cout, Hs = dec_ch[i], dec_ch[i + 1], 8 << i
                n_out = B * 4 * Hs * Hs * Cout
                tbuf = self._buf(f"{tag}{i}.t", n_out)
                fl = 50.0 * B * Hs * Hs * Cin * Cout
                last = i == L - 1
                u = self._buf(f"{tag}{i}.u", n_out) if last else None
                u_s = None if last else self._sbuf(f"{tag}{i}.us", n_out)
                mean, rstd, ws = conv_bn_fwd(fwd, f"{tag}{i}", 1, (P(cur_s), P(dec_pk[i][1]), P(tbuf)), (B, Hs, Hs, Cin, Cout, 2), fl,
                                             B * 4 * Hs * Hs, Cout, blk.bn, u, u_s, 1)
                recs.append((blk, Cin, Cout, Hs, dec_pk[i][0], tbuf, mean, rstd, ws, cur_s))
                cur_s = u_s
            flf = 50.0 * B * S * S * Cf
            if fin_edge:
                fwd.add("vp_conv5_smallout_bf16x3", P(u), P(fp0), P(fin.bias), out_ptr, B, S, S, Cf, 1, _ACT_SIGMOID, flops=flf, tag=f"{tag}.fin.fwd")
            else:
                fwd.add("vp_conv5_gather_f32", P(u), P(fp0), P(fin.bias), out_ptr, B, S, S, Cf, 1, 1, _ACT_SIGMOID, flops=flf, tag=f"{tag}.fin.fwd")
            return {"tag": tag, "z": zbuf, "d": d, "fc": fc_rec, "blocks": recs, "u": u, "out": out_ptr}

        dec1 = dec_fwd("dec", self.z, at(xcat, n_pix))            # x_tilde = xcat[B:2B]

        # ---- param_encoder (DirectDecoder, models/networks.py:118-148): bias-ed Linear layers without activations ----
        pe_rec = []

        def pe_lin(tag, lin, x):
            N, K = lin.weight.shape
            y = self._buf(f"pe.{tag}", B * N)
            lin_fwd(fwd, x, lin.weight, lin.bias, y, B, N, K)
            pe_rec.append((tag, lin, x, y, N, K))
            return y
        t_ = self.z
        for j, lin in enumerate(pe.head):
            t_ = pe_lin(f"head{j}", lin, t_)
        pe_head = t_
        self.p_r = pe_lin("r1", pe.r_fc[1], pe_lin("r0", pe.r_fc[0], pe_head))
        self.p_xy = pe_lin("xy1", pe.xy_fc[1], pe_lin("xy0", pe.xy_fc[0], pe_head))
        n_r, n_xy = pe.r_fc[1].weight.shape[0], pe.xy_fc[1].weight.shape[0]
        if n_r + n_xy != self.targets.shape[1]:
            raise ValueError("targets must have as many columns as DirectDecoder returns")
        self.l1 = self._buf("l1", 1)
        d_r, d_xy = self._buf("g.d_r", B * n_r), self._buf("g.d_xy", B * n_xy)
        fwd.add("vp_smooth_l1_cat_f32", P(self.targets), P(self.p_r), P(self.p_xy), B, n_r, n_xy, 1.0 / B, P(self.l1), P(d_r), P(d_xy))

        dec2 = dec_fwd("decp", self.z_p, at(xcat, 2 * n_pix))     # x_p = xcat[2B:3B]

        # ---- discriminator on (x | x_tilde | x_p) (models/networks.py:151-198), one pass for "REC" and "GAN" ----
        conv0 = disc.conv[0][0]
        C0 = conv0.weight.shape[0]
        if C0 not in (32, 64) or conv0.weight.shape[1] != 1:
            raise NotImplementedError("discriminator stem must be 1 -> 32 | 64 channels")
        y0 = self._buf("disc0.y", n3 * S * S * C0)
        y0s = self._sbuf("disc0.ys", n3 * S * S * C0)
        fwd_disc.add("vp_conv5_smallin_fwd_bf16x3", P(xcat), P(conv0.weight), P(conv0.bias), P(y0), n3, S, S, 1, C0, _ACT_RELU,
                     flops=50.0 * n3 * S * S * C0, tag="disc0.fwd")
        fwd_disc.add("vp_split_f32", P(y0), P(y0s), n3 * S * S * C0)
        disc_ch = [C0] + [blk.conv.weight.shape[0] for blk in list(disc.conv)[1:]]
        disc_rec = []
        in_s = y0s
        self._disc_replay = []
        for i in range(1, L + 1):
            blk = disc.conv[i]
            Cin, Cout, Hs = disc_ch[i - 1], disc_ch[i], S >> i
            n_out = n3 * Hs * Hs * Cout
            c = self._buf(f"disc{i}.c", n_out)
            fl = 50.0 * n3 * Hs * Hs * Cin * Cout
            last = i == L
            a = self._buf(f"disc{i}.a", n_out) if last else None
            a_s = None if last else self._sbuf(f"disc{i}.as", n_out)
            p0 = self._sbuf(f"disc{i}.p0s", Cout * 25 * Cin)
            p1 = self._sbuf(f"disc{i}.p1s", Cin * 25 * Cout)
            pack(blk.conv.weight, p0, p1, Cout, Cin, True)
            mean, rstd, ws = conv_bn_fwd(fwd_disc, f"disc{i}", 0, (P(in_s), P(p0), P(c)), (n3, Hs, Hs, Cin, Cout, 2), fl, n3 * Hs * Hs, Cout,
                                         blk.bn, a, a_s, 2)
            self._disc_replay.append(blk.bn)
            disc_rec.append((blk, Cin, Cout, Hs, p1, c, mean, rstd, ws, in_s))
            disc_last, in_s = a, a_s
        Cd = disc_ch[-1]
        nf = 64 * Cd                                               # features per image at the tap
        tap = disc_rec[-1][5]                                      # pre-BatchNorm output of the last block: "disc_layer"
        dflat = self._buf("disc.flat", n3 * nf)
        fwd_disc.add("vp_nhwc_to_nchw_f32", P(disc_last), P(dflat), n3, Cd, 8, 8)
        dl0, dl_bn, dl3 = disc.fc[0], disc.fc[1], disc.fc[3]
        Hd = dl0.weight.shape[0]
        dh, dhb = self._buf("disc.h", n3 * Hd), self._buf("disc.hb", n3 * Hd)
        lin_fwd(fwd_disc, dflat, dl0.weight, None, dh, n3, Hd, nf)
        dh_mean, dh_rstd, dh_ws = bn_plain_fwd(fwd_disc, "disc.fc", dh, n3, Hd, dl_bn, dhb, None, 1)
        logit = self._buf("disc.logit", n3)
        lin_fwd(fwd_disc, dhb, dl3.weight, dl3.bias, logit, n3, 1, Hd)
        # coefficients of the summed losses, formed in fp32 as autograd's accumulation forms them (train.py:63-66)
        one = np.float32(1.0)
        self.c_disc = float(one + (-np.float32(1.0 - self.lam)))          # loss_discriminator: 1 from itself, -(1 - lambda) from loss_decoder
        self.c_mse = float(one + np.float32(self.lam))                    # sum(mse): 1 from loss_encoder, lambda from loss_decoder
        self.disc_class = self._buf("disc_class", n3, 1)
        self.bce_sums = self._buf("bce_sums", 3)
        dlogit = self._buf("g.dlogit", n3)
        fwd_disc.add("vp_gan_head_f32", P(logit), B, self.c_disc, P(self.disc_class), P(self.bce_sums), P(dlogit))
        self.mse = self._buf("mse", B)
        fwd_disc.add("vp_half_sqdiff_rowsum_f32", P(tap), at(tap, B * nf), P(self.mse), B, nf)
        self.nle_rows = self._buf("nle_rows", B)                   # sum over pixels of "nle" (models/networks.py:267) per image
        fwd_disc.add("vp_half_sqdiff_rowsum_f32", P(xcat), at(xcat, n_pix), P(self.nle_rows), B, S * S)
        self.disc_layer_nhwc = tap.view(n3, 8, 8, Cd)

        # =================================================== backward ===================================================
        big_d = max([n3 * S * S * C0] + [n3 * r[3] * r[3] * r[2] for r in disc_rec] + [n3 * nf])
        big_v = max([B * F0, B * F1, n_pix * Cf] + [B * 4 * r[3] * r[3] * r[2] for r in dec1["blocks"]] + [B * r[3] * r[3] * r[2] for r in enc_rec])
        gA, gB = self._buf("g.A", big_d), self._buf("g.B", big_d)          # discriminator phase
        hA, hB = self._buf("g.hA", big_v), self._buf("g.hB", big_v)        # decoder / encoder phases
        big_s = max([n3 * r[3] * r[3] * r[2] for r in disc_rec] + [B * 4 * r[3] * r[3] * r[2] for r in dec1["blocks"]]
                    + [B * r[3] * r[3] * r[2] for r in enc_rec])
        n_gs = max(2, int(os.environ.get("VP_GS_BUFS", "2")))     # rotation depth of the gradient planes (engine.py)
        gS2 = [self._sbuf(f"g.S{j}", big_s) for j in range(n_gs)]
        gs_last = [None] * n_gs
        gs_turn = [0]

        def next_gs():
            k = gs_turn[0] % n_gs
            gs_turn[0] += 1
            if gs_last[k] is not None:
                bwd.wait_side(gs_last[k])
            return k

        nws = 0
        for r in disc_rec:
            nws = max(nws, lib.vp_conv5_wgrad_workspace_bytes(n3, r[3], r[3], r[1], r[2], 2), lib.vp_conv5_wgrad_bf16x3_workspace_bytes(n3, r[3], r[3], r[1], r[2], 2))
        for r in enc_rec[1:]:
            nws = max(nws, lib.vp_conv5_wgrad_workspace_bytes(B, r[3], r[3], r[1], r[2], 2), lib.vp_conv5_wgrad_bf16x3_workspace_bytes(B, r[3], r[3], r[1], r[2], 2))
        for r in dec1["blocks"]:
            nws = max(nws, lib.vp_conv5_wgrad_workspace_bytes(B, r[3], r[3], r[2], r[1], 2), lib.vp_conv5_wgrad_bf16x3_workspace_bytes(B, r[3], r[3], r[2], r[1], 2))
        ws_wg = self._ws("g.wgrad.ws", nws)                        # side-stream weight gradients run one after another
        # CU budget of a weight gradient beside the main stream's kernels (csrc/wgrad5.h; 128 / 192 / 256: 5.78 / 5.91 / 6.04 ms)
        side_cus = int(os.environ.get("VP_WGRAD_SIDE_CUS", "128"))

        def gather_block_bwd(rec, Bn, cur, gfn, need_dx, tag, pre_split=None):
            """conv5x5 s2 + BatchNorm + ReLU block, dy in ``cur`` (fp32 NHWC) -> dx in ``cur``"""
            blk, Cin, Cout, Hs, p1, c, mean, rstd, ws, in_s = rec
            R = Bn * Hs * Hs
            fl = 50.0 * Bn * Hs * Hs * Cin * Cout
            k = next_gs()
            gS = gS2[k]
            if pre_split is None:
                bn_bwd(c, cur, None, gS, R, Cout, blk.bn, mean, rstd, ws, gfn)
            else:
                pre_split(gS)
            gs_last[k] = side_slot()
            bwd.add("vp_conv5_wgrad_bf16x3_cus", P(in_s), P(gS), P(gfn(blk.conv.weight)), Bn, Hs, Hs, Cin, Cout, 2, 0, P(ws_wg), ws_wg.numel() * 4,
                    flops=fl, tag=f"{tag}.wgrad", side=gs_last[k], side_args={9: (0, side_cus)})
            if need_dx:
                bwd.add("vp_conv5_scatter_bf16x3", P(gS), P(p1), P(cur), Bn, Hs, Hs, Cout, Cin, 2, flops=fl, tag=f"{tag}.dgrad")

        # ---- discriminator head ----
        lin_wgrad(bwd, dlogit, dhb, grad_of(dl3.weight), n3, 1, Hd)
        colsum(bwd, "disc.b3", dlogit, grad_of(dl3.bias), n3, 1)
        g_hb, g_h = self._buf("g.disc_hb", n3 * Hd), self._buf("g.disc_h", n3 * Hd)
        lin_dgrad(bwd, dlogit, dl3.weight, g_hb, n3, 1, Hd)
        bn_bwd(dh, g_hb, g_h, None, n3, Hd, dl_bn, dh_mean, dh_rstd, dh_ws, grad_of)
        lin_wgrad(bwd, g_h, dflat, grad_of(dl0.weight), n3, Hd, nf)
        lin_dgrad(bwd, g_h, dl0.weight, gA, n3, Hd, nf)
        bwd.add("vp_nchw_to_nhwc_f32", P(gA), P(gB), n3, Cd, 8, 8)         # gB = d (last block's activation)
        # ---- last block: its pre-BatchNorm output also feeds the feature loss (1 + lambda) * sum(mse) ----
        g_mse = torch.full((B,), self.c_mse, dtype=torch.float32, device=self.dev)
        self._bufs["g.c_mse"] = g_mse
        m_ab = self._buf("g.mse_ab", 2 * B * nf)
        blk, Cin, Cout, Hs, p1, c, mean, rstd, ws, in_s_l = disc_rec[-1]

        def tap_split(gS):
            bn_bwd(c, gB, gA, None, n3 * Hs * Hs, Cout, blk.bn, mean, rstd, ws, grad_of)              # gA = BatchNorm path
            bwd.add("vp_half_sqdiff_bwd_f32", P(tap), at(tap, B * nf), P(g_mse), P(m_ab), at(m_ab, B * nf), B, nf, 1)
            bwd.add("vp_add_f32", P(gA), P(m_ab), P(gA), 2 * B * nf)                              # + feature loss (original, reconstructed)
            bwd.add("vp_split_f32", P(gA), P(gS), n3 * nf)
        gather_block_bwd(disc_rec[-1], n3, gB, grad_of, True, f"disc{L}", pre_split=tap_split)
        for i in range(L - 1, 0, -1):
            gather_block_bwd(disc_rec[i - 1], n3, gB, grad_of, True, f"disc{i}")
        # ---- stem: conv 1 -> C0 + bias + ReLU (ReLU in the convolution's epilogue) ----
        n0 = n3 * S * S * C0
        bwd.add("vp_act_bwd_from_y_f32", P(y0), P(gB), P(gA), n0, _ACT_RELU, 0.0)                  # gA = d (conv output)
        k_stem = side_slot()
        colsum(bwd, "disc.b0", gA, grad_of(conv0.bias), n3 * S * S, C0, side=side_slot())
        ws_w0 = self._ws("disc0.wgws", lib.vp_conv5_wgrad_workspace_bytes(n3, S, S, 1, C0, 1))
        bwd.add("vp_conv5_wgrad_f32", P(xcat), P(gA), P(grad_of(conv0.weight)), n3, S, S, 1, C0, 1, P(ws_w0), ws_w0.numel() * 4,
                flops=50.0 * n3 * S * S * C0, tag="disc0.wgrad", side=k_stem)
        self._wf = self._buf("disc0.wf", 25 * C0)                   # taps flipped, [ci = 1][tap][co]: the input gradient is a correlation
        self._conv0_w = conv0.weight
        dxc = self._buf("g.dxcat", 2 * n_pix)                       # d x_tilde | d x_p from the discriminator
        bwd.add("vp_conv5_smallout_bf16x3", at(gA, B * S * S * C0), P(self._wf), None, P(dxc), 2 * B, S, S, C0, 1, _ACT_NONE,
                flops=50.0 * 2 * B * S * S * C0, tag="disc0.dgrad")
        bwd.hook("disc_done")       # every discriminator gradient is launched and none of its parameters is read again in this step

        # ---- decoder ----
        def dec_bwd(rec, dout, gfn, dz):
            """``dout`` = pointer to the gradient w.r.t. the sigmoid output (fp32, B*S*S); writes d z into ``dz`` when given"""
            tag = rec["tag"]
            dlg = self._buf(f"g.{tag}.dlogit", n_pix)
            bwd.add("vp_act_bwd_from_y_f32", rec["out"], dout, P(dlg), n_pix, _ACT_SIGMOID, 0.0)
            colsum(bwd, f"{tag}.finb", dlg, gfn(fin.bias), n_pix, 1, side=side_slot())
            flf = 50.0 * B * S * S * Cf
            if fin_edge:
                nb = lib.vp_conv5_smallout_wgrad_bf16x3_workspace_bytes(B, S, S, Cf, 1)      # 0: shape outside the taps-in-M kernel
                if nb:
                    wsf = self._ws(f"{tag}.finwg.ws", nb)
                    bwd.add("vp_conv5_smallout_wgrad_bf16x3", P(rec["u"]), P(dlg), P(gfn(fin.weight)), B, S, S, Cf, 1, P(wsf), wsf.numel() * 4,
                            flops=flf, tag=f"{tag}.fin.wgrad", side=side_slot())
                else:
                    wsf = self._ws(f"{tag}.finwg.ws", lib.vp_conv5_wgrad_workspace_bytes(B, S, S, Cf, 1, 1))
                    bwd.add("vp_conv5_wgrad_f32", P(rec["u"]), P(dlg), P(gfn(fin.weight)), B, S, S, Cf, 1, 1, P(wsf), wsf.numel() * 4,
                            flops=flf, tag=f"{tag}.fin.wgrad", side=side_slot())
                bwd.add("vp_conv5_smallin_dgrad_bf16x3", P(dlg), P(fin.weight), P(hA), B, S, S, 1, Cf, flops=flf, tag=f"{tag}.fin.dgrad")
            else:
                wsf = self._ws(f"{tag}.finwg.ws", lib.vp_conv5_wgrad_workspace_bytes(B, S, S, Cf, 1, 1))
                bwd.add("vp_conv5_wgrad_f32", P(rec["u"]), P(dlg), P(gfn(fin.weight)), B, S, S, Cf, 1, 1, P(wsf), wsf.numel() * 4,
                        flops=flf, tag=f"{tag}.fin.wgrad", side=side_slot())
                bwd.add("vp_conv5_scatter_f32", P(dlg), P(fp1), P(hA), B, S, S, 1, Cf, 1, flops=flf, tag=f"{tag}.fin.dgrad")
            for i in range(L - 1, -1, -1):
                blk, Cin, Cout, Hs, p0, tbuf, mean, rstd, ws, in_s = rec["blocks"][i]
                fl = 50.0 * B * Hs * Hs * Cin * Cout
                k = next_gs()
                gS = gS2[k]
                bn_bwd(tbuf, hA, None, gS, B * 4 * Hs * Hs, Cout, blk.bn, mean, rstd, ws, gfn)
                gs_last[k] = side_slot()
                bwd.add("vp_conv5_wgrad_bf16x3_cus", P(gS), P(in_s), P(gfn(blk.conv.weight)), B, Hs, Hs, Cout, Cin, 2, 0, P(ws_wg), ws_wg.numel() * 4,
                        flops=fl, tag=f"{tag}{i}.wgrad", side=gs_last[k], side_args={9: (0, side_cus)})
                bwd.add("vp_conv5_gather_bf16x3", P(gS), P(p0), None, P(hA), B, Hs, Hs, Cout, Cin, 2, _ACT_NONE, flops=fl, tag=f"{tag}{i}.dgrad")
            bwd.add("vp_nhwc_to_nchw_f32", P(hA), P(hB), B, dsize, 8, 8)                             # hB = d db (B, F1)
            mean, rstd, ws = rec["fc"]
            bn_bwd(rec["d"], hB, hA, None, B, F1, dfc_bn, mean, rstd, ws, gfn)                       # hA = d d
            lin_wgrad(bwd, hA, rec["z"], gfn(dfc_lin.weight), B, F1, Z)
            if dz is not None:
                lin_dgrad(bwd, hA, dfc_lin.weight, dz, B, F1, Z)

        # second pass first (x_p: only the discriminator's gradient reaches it), into the shadow arena
        dec_bwd(dec2, at(dxc, n_pix), grad2_of, None)

        # ---- param_encoder ----
        pe_by = {r[0]: r for r in pe_rec}

        def pe_bwd(tag, dy):
            _, lin, x, y, N, K = pe_by[tag]
            lin_wgrad(bwd, dy, x, grad_of(lin.weight), B, N, K)
            colsum(bwd, f"pe.{tag}", dy, grad_of(lin.bias), B, N)
            dx = self._buf(f"g.pe.{tag}.dx", B * K)
            lin_dgrad(bwd, dy, lin.weight, dx, B, N, K)
            return dx
        d_head_a = pe_bwd("r0", pe_bwd("r1", d_r))
        d_head_b = pe_bwd("xy0", pe_bwd("xy1", d_xy))
        bwd.add("vp_add_f32", P(d_head_a), P(d_head_b), P(d_head_a), d_head_a.numel())
        dz_pe = d_head_a
        for j in range(len(pe.head) - 1, -1, -1):
            dz_pe = pe_bwd(f"head{j}", dz_pe)
        # ---- first decoder pass: d x_tilde = discriminator part + reconstruction loss mean((x - x_tilde)^2) (train.py:61) ----
        g_rec = torch.full((1,), 2.0 / n_pix, dtype=torch.float32, device=self.dev)
        self._bufs["g.c_rec"] = g_rec
        dxt = self._buf("g.dxt", n_pix)
        bwd.add("vp_half_sqdiff_bwd_f32", P(xcat), at(xcat, n_pix), P(g_rec), None, P(dxt), 1, n_pix, 1)     # 2/n (x_tilde - x)
        bwd.add("vp_add_f32", P(dxt), P(dxc), P(dxt), n_pix)
        dz_dec = self._buf("g.dz_dec", B * Z)
        dec_bwd(dec1, P(dxt), grad_of, dz_dec)
        bwd.hook("dec_done")        # both decoder passes' gradients are launched; the decoder's parameters are not read again
        dz = self._buf("g.dz", B * Z)
        bwd.add("vp_add_f32", P(dz_dec), P(dz_pe), P(dz), B * Z)

        # ---- encoder ----
        dmu, dlv = self._buf("g.dmu", B, Z), self._buf("g.dlv", B, Z)
        bwd.add("vp_latent_bwd_f32", P(self.mu), P(self.logvar), P(self.eps), P(dz), None, 1.0, P(dmu), P(dlv), B, Z)
        dhb_a, dhb_b = self._buf("g.dhb_a", B * H1), self._buf("g.dhb_b", B * H1)
        for lin, dsrc, dst in ((enc.l_mu, dmu, dhb_a), (enc.l_var, dlv, dhb_b)):
            lin_wgrad(bwd, dsrc, hb, grad_of(lin.weight), B, Z, H1)
            colsum(bwd, f"enc.{'mu' if lin is enc.l_mu else 'var'}", dsrc, grad_of(lin.bias), B, Z)
            lin_dgrad(bwd, dsrc, lin.weight, dst, B, Z, H1)
        bwd.add("vp_add_f32", P(dhb_a), P(dhb_b), P(dhb_a), B * H1)
        g_eh = self._buf("g.enc_h", B * H1)
        bn_bwd(h, dhb_a, g_eh, None, B, H1, fc_bn, h_mean, h_rstd, h_ws, grad_of)
        lin_wgrad(bwd, g_eh, flat, grad_of(fc_lin.weight), B, H1, F0)
        lin_dgrad(bwd, g_eh, fc_lin.weight, hA, B, H1, F0)
        bwd.hook("enc_dense_done")  # the encoder's dense gradients (fc, l_mu, l_var) are launched, their parameters not read again
        bwd.add("vp_nchw_to_nhwc_f32", P(hA), P(hB), B, size, 8, 8)
        for i in range(L - 1, 0, -1):
            gather_block_bwd(enc_rec[i], B, hB, grad_of, True, f"enc{i}")
        blk, Cin, Cout, Hs, _, c, mean, rstd, ws, (xcol, KC) = enc_rec[0]
        k = next_gs()
        gS = gS2[k]
        bn_bwd(c, hB, None, gS, B * Hs * Hs, Cout, blk.bn, mean, rstd, ws, grad_of)
        dwc = self._buf("enc0.dwc", Cout * KC)
        ws0 = self._ws("enc0.wgws", lib.vp_conv_wgrad_bf16x3_workspace_bytes(B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1))
        bwd.add("vp_conv_wgrad_bf16x3", P(xcol), P(gS), P(dwc), B, Hs, Hs, Hs, Hs, KC, Cout, 1, 1, P(ws0), ws0.numel() * 4,
                flops=50.0 * B * Hs * Hs * Cout, tag="enc0.wgrad")
        bwd.add("vp_unpack_dw_im2col5_f32", P(dwc), P(grad_of(blk.conv.weight)), Cout, 1)

        # ---- finish: batched weight pack at the head of the forward plan, shared dense workspace ----
        self._pack_jobs = (_lib.PackJob * len(pack_jobs))(*pack_jobs)
        fwd.add_first("vp_pack_w5_batch", self._pack_jobs, len(pack_jobs), side=k_pack)
        wsg = self._ws("gemm.ws", gemm_need[0])
        for a in gemm_calls:
            a[13], a[14] = P(wsg), wsg.numel() * 4
        self._fwd, self._fwd_disc, self._bwd = fwd, fwd_disc, bwd
        self._n_side_events = n_side[0]
        self._snap = [torch.empty_like(bn.running_mean) for bn in self._disc_replay] + [torch.empty_like(bn.running_var) for bn in self._disc_replay]
        self._running = [bn.running_mean for bn in self._disc_replay] + [bn.running_var for bn in self._disc_replay]
        self.x_tilde = xcat[B:2 * B]
        self.x_p = xcat[2 * B:]

    # ---- execution ---------------------------------------------------------------------------
    def _side_ctx(self):
        if not self._n_side_events:
            return None
        if not hasattr(self, "_side"):
            self._side = _SideCtx(self._n_side_events)
        return self._side

    @torch.no_grad()
    def forward_backward(self, x: torch.Tensor, targets: torch.Tensor, eps: torch.Tensor, z_p: torch.Tensor, timers: Optional[dict] = None,
                         early_updates: bool = False):
        """Gradients of the summed losses of train.py:61-73 land in the four optimisers' flat gradient arenas.  ``eps`` is the
        reparameterisation noise (models/networks.py:230), ``z_p`` the prior sample (:240).  Outputs (static buffers, overwritten
        by the next step): ``x_tilde``, ``x_p``, ``mu``, ``logvar``, ``disc_class``, ``kl``, ``mse``, ``bce_sums``, ``l1``,
        ``nle_rows``; ``losses()`` assembles train.py's five scalars from them."""
        B = self.B
        self.xcat[:B].copy_(x.reshape(B, 1, self.S, self.S), non_blocking=True)
        self.targets.copy_(targets, non_blocking=True)
        self.eps.copy_(eps, non_blocking=True)
        self.z_p.copy_(z_p, non_blocking=True)
        w = self._conv0_w.detach()
        self._wf.view(1, 25, -1).copy_(w.flip(2, 3).permute(1, 2, 3, 0).reshape(1, 25, -1))
        s = torch.cuda.current_stream().cuda_stream
        side = self._side_ctx() if timers is None else None
        self._fwd.run(s, timers, side=side)
        # the reference runs the discriminator twice per step: the blocks' running statistics move twice (second update replayed
        # from the buffers before / after the single pass: rm2 = (2 - m) rm1 - (1 - m) rm0)
        torch._foreach_copy_(self._snap, self._running)
        self._fwd_disc.run(s, timers, side=side)
        m = float(self._disc_replay[0].momentum)
        torch._foreach_mul_(self._running, 2.0 - m)
        torch._foreach_add_(self._running, self._snap, alpha=-(1.0 - m))
        self._early_done = set()
        self._early_partial = {}
        hooks = None
        if early_updates and side is not None:
            # RMSprop of an arena as soon as its gradients are launched, ON THE SIDE STREAM (behind that arena's weight gradients, which
            # run there; the fork covers the gradients the main stream produced): HBM-bound updates underneath the MFMA-bound backward
            # of the networks that are still to come instead of behind the whole step
            def early(kind):
                def fn(sd):
                    if not self._arena_opts[kind]:          # (an arena shared between sub-networks: updated at the end of the step)
                        return
                    sd.flush()
                    sd.fork.record()
                    sd.stream.wait_event(sd.fork)
                    with torch.cuda.stream(sd.stream):
                        if kind == "dec":
                            a = self._dec_arena
                            _lib.call("vp_add_f32", _ptr(a.flat_grad), _ptr(self._dec_shadow), _ptr(a.flat_grad), a.flat_grad.numel(),
                                      c_void_p(sd.stream.cuda_stream))
                        for o in self._arena_opts[kind]:
                            o.begin_step()
                            o.step_range(0, o.arena.flat_param.numel())
                            self._early_done.add(id(o))
                return fn
            def early_enc_dense(sd):
                if not self._enc_dense:
                    return
                sd.flush()
                sd.fork.record()
                sd.stream.wait_event(sd.fork)
                with torch.cuda.stream(sd.stream):
                    for o, lo in self._enc_dense:
                        o.begin_step()
                        o.step_range(lo, o.arena.flat_param.numel())
                        self._early_partial[id(o)] = lo
            hooks = {"disc_done": early("disc"), "dec_done": early("dec"), "enc_dense_done": early_enc_dense}
        self._bwd.run(s, timers, side=side, hooks=hooks)
        if side is not None:
            side.flush()
            torch.cuda.current_stream().wait_stream(side[0])
        if not any(id(o) in self._early_done for o in self._arena_opts["dec"]):
            a = self._dec_arena
            _lib.call("vp_add_f32", _ptr(a.flat_grad), _ptr(self._dec_shadow), _ptr(a.flat_grad), a.flat_grad.numel(), c_void_p(s))
        for o in self.opts:        # the plan wrote the arenas: a ``.grad`` left None by module.zero_grad() must not read as "no gradient"
            o.arena.adopt_views()
        self._steps_since_sync += 1

    def step(self, x, targets, eps, z_p, timers: Optional[dict] = None):
        """One training iteration of train.py:43-78: forward, losses, backward, gradient all-reduce (several ranks), four RMSprop updates."""
        early = (not parallel.dp_active(self.group)) and timers is None and os.environ.get("VP_GAN_EARLY_UPDATES", "1") != "0"
        self.forward_backward(x, targets, eps, z_p, timers, early_updates=early)
        arenas = []
        for o in self.opts:
            if o.arena.numel and all(o.arena is not q for q in arenas):
                arenas.append(o.arena)
        if parallel.dp_active(self.group):
            works = [parallel.allreduce_flat_grads(a.flat_grad, self.group, async_op=True) for a in arenas]
            for w in works:
                if w is not None:
                    w.wait()
        for o in self.opts:
            if id(o) in self._early_done:      # updated on the side stream during backward (forward_backward's hooks)
                continue
            if id(o) in self._early_partial:   # its dense slice was: the rest now, same step count
                o.step_range(0, self._early_partial[id(o)])
                continue
            o.begin_step()
            o.step_range(0, o.arena.flat_param.numel() if o.arena.numel else 0)

    def losses(self) -> dict:
        """train.py:61-66's scalars from the last step's buffers (host arithmetic on a handful of device scalars; a sync)."""
        kl, mse = float(self.kl.sum()), float(self.mse.sum())
        d = float(self.bce_sums.sum())
        n = self.B * self.S * self.S
        return {"loss_recon": 2.0 * float(self.nle_rows.sum()) / n, "loss_encoder": kl + mse, "loss_discriminator": d,
                "loss_decoder": self.lam * mse - (1.0 - self.lam) * d, "loss_aux": float(self.l1)}

    @property
    def params(self) -> torch.Tensor:
        """DirectDecoder's output cat([r, xy], -1) of the last step"""
        return torch.cat([self.p_r.view(self.B, -1), self.p_xy.view(self.B, -1)], dim=-1)

    @property
    def disc_layer(self) -> torch.Tensor:
        """the "REC" output of the last step in the reference's (C, H, W) flatten order, (3B, C*8*8)"""
        return self.disc_layer_nhwc.permute(0, 3, 1, 2).reshape(3 * self.B, -1)

    def sync_counters(self):
        """Advance BatchNorm ``num_batches_tracked`` buffers (bookkeeping only; kept off the hot path)."""
        n = self._steps_since_sync
        if n:
            for bn, count in self._bn_counts:
                bn.num_batches_tracked.add_(n * count)
            self._steps_since_sync = 0
