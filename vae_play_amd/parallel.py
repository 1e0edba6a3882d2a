"""Data parallelism for the VAE step: one process per GPU, the minibatch sharded by rank,
replicated weights, and ONE sum all-reduce of the flat gradient arena per step
(torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; this is new functionality
defined by BASELINE.json's north_star.  BatchNorm statistics stay per-rank: the
parity definition for W ranks is "the average of W single-shard reference gradients".
The 1/W factor is folded into the fused optimiser kernel (grad_scale), not a separate pass.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(global_batch: int, rank: int, world: int) -> tuple:
    """Rank r owns x[r*B/W : (r+1)*B/W]."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def dp_active(group=None) -> bool:
    """Several ranks, or a one-rank group with VP_DP_FORCE=1 (tools/dp_rccl_selftest.py: drives every collective of
    the data-parallel step through RCCL on a one-GPU box, where a second rank cannot share the card)."""
    if not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("VP_DP_FORCE") == "1"


def allreduce_flat_grads(flat_grad: torch.Tensor, group=None, async_op: bool = False):
    """The step's single collective: SUM all-reduce of the gradient arena."""
    if dp_active(group):
        return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return None


def allgather_rows(out_all: torch.Tensor, local: torch.Tensor, group=None, async_op: bool = False):
    """out_all[(r*B):(r+1)*B] = rank r's ``local`` (B rows).  Used to exchange the two rank-B factors of a dense
    layer's weight gradient (dW = dY^T X) instead of all-reducing the weight-sized gradient itself."""
    if dist.get_backend(group) == "nccl":          # RCCL writes straight into the gathered buffer
        return dist.all_gather_into_tensor(out_all, local.contiguous(), group=group, async_op=async_op)
    world = dist.get_world_size(group)
    chunks = list(out_all.view(world, *local.shape).unbind(0))
    return dist.all_gather(chunks, local, group=group, async_op=async_op)


class CommTrace:
    """Evidence hooks for the first real multi-GPU run (bench.py ``comm``): per collective of a step its name, kind, bytes and
    the device time from "its inputs are final on the producing stream" to "its result is ready" -- an event pair on a
    communicator-side stream that waits for the producer, issues the collective and waits for it (collectives are serialised
    on the backend's own stream, so the pair includes queueing behind earlier buckets: that IS what the step sees) -- and
    ``exposed``: the time the main stream spends blocked in waits for collectives.  Used on instrumented steps only; the
    un-traced step hands its buckets to the backend directly."""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.records = []          # (name, kind, bytes, e0, e1)
        self.waits = []            # (e0, e1) on the waiting stream
        self.steps = 0

    def issue(self, name: str, kind: str, nbytes: int, fn):
        """``fn()`` issues the collective (async) on the current stream context; returns an object with ``wait()``."""
        self.stream.wait_stream(torch.cuda.current_stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self.stream):
            e0.record()
            w = fn()
            if w is not None:
                w.wait()
            e1.record()
        self.records.append((name, kind, int(nbytes), e0, e1))
        return self

    def wait(self):
        """the current stream waits for everything issued so far; the stall is recorded as exposed time"""
        cur = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(cur)
        cur.wait_stream(self.stream)
        e1.record(cur)
        self.waits.append((e0, e1))

    def end_step(self):
        self.steps += 1

    def summary(self) -> dict:
        """after a synchronize: per-bucket averages over the traced steps"""
        torch.cuda.synchronize()
        n = max(self.steps, 1)
        by = {}
        for name, kind, nbytes, e0, e1 in self.records:
            d = by.setdefault(name, {"name": name, "kind": kind, "bytes": nbytes, "ms": 0.0, "calls": 0})
            d["ms"] += e0.elapsed_time(e1)
            d["calls"] += 1
        buckets = []
        for d in by.values():
            d["ms"] = round(d["ms"] / d["calls"], 4)
            d["calls_per_step"] = d.pop("calls") / n
            buckets.append(d)
        exposed = sum(e0.elapsed_time(e1) for e0, e1 in self.waits) / n
        kinds = {}
        for b in buckets:
            kinds[b["kind"]] = kinds.get(b["kind"], 0) + b["calls_per_step"]
        return {"buckets": buckets, "exposed_ms_per_step": round(exposed, 4),
                "collectives_per_step": {k: round(v, 3) for k, v in kinds.items()},
                "bytes_per_step": int(sum(b["bytes"] * b["calls_per_step"] for b in buckets)), "traced_steps": self.steps}


def broadcast_flat_params(flat_param: torch.Tensor, src: int = 0, group=None) -> None:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_param, src=src, group=group)
        from . import ops
        ops.PARAM_EPOCH[0] += 1          # the parameters changed under their views: packed-weight caches are stale


class DataParallelStep:
    """Wraps a flat-arena optimiser: ``dp.step()`` = all-reduce(sum) + fused update with 1/W.

    Usage mirrors the reference loop body (train_BE.py:62-64):
        optim.zero_grad(); loss.backward(); dp.step()
    """

    def __init__(self, optimizer, group=None, broadcast_from: Optional[int] = 0):
        self.optimizer, self.group = optimizer, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        optimizer.grad_scale = 1.0 / self.world
        if broadcast_from is not None:
            broadcast_flat_params(optimizer.flat_param, broadcast_from, group)

    def zero_grad(self) -> None:
        self.optimizer.zero_grad()

    def step(self) -> None:
        self.optimizer.arena.gather_grads()      # gradients that live outside the arena (module.zero_grad()) come home first
        allreduce_flat_grads(self.optimizer.flat_grad, self.group)
        self.optimizer.step()


class DataParallelGroup:
    """Several flat-arena optimisers over one model family (the VAE-GAN's four, the font GAN's three): ``step()`` brings
    every gradient into its arena, sum-all-reduces each arena ONCE (parameters shared between optimisers live in one
    arena only, so their gradient is reduced once too) and then runs the fused updates with the 1/W factor folded in.

        for o in opts: o.zero_grad()
        loss.backward()
        group.step()                      # instead of: for o in opts: o.step()
    """

    def __init__(self, optimizers, group=None, broadcast_from: Optional[int] = 0):
        self.optimizers, self.group = list(optimizers), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        for o in self.optimizers:
            o.grad_scale = 1.0 / self.world
            if broadcast_from is not None:
                broadcast_flat_params(o.flat_param, broadcast_from, group)

    def zero_grad(self) -> None:
        for o in self.optimizers:
            o.zero_grad()

    def step(self, subset=None) -> None:
        """``subset``: the optimisers to step now (a phase of a GAN iteration); default all."""
        opts = self.optimizers if subset is None else list(subset)
        for o in opts:
            if o.arena.numel:
                o.arena.gather_grads()
        works = [allreduce_flat_grads(o.flat_grad, self.group, async_op=True) for o in opts if o.arena.numel]
        # a shared parameter whose owning arena is not part of this phase (the font GAN's style-encoder optimiser steps
        # parameters that live in the generator's arena): reduce its gradient on its own
        owners = {id(o.arena) for o in opts if o.arena.numel}
        for o in opts:
            for q in o.arena.foreign:
                if q.grad is not None and id(getattr(q, "_vp_arena", None)) not in owners:
                    g = q.grad if q.grad.is_contiguous() else q.grad.contiguous()
                    w = allreduce_flat_grads(g, self.group, async_op=True)
                    if g is not q.grad and w is not None:
                        w.wait()
                        q.grad.copy_(g)
                    else:
                        works.append(w)
        for w in works:
            if w is not None:
                w.wait()
        for o in opts:
            o.step()
