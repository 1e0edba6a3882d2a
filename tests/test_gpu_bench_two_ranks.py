"""bench.py's multi-rank line on the one GPU of the box (two gloo ranks, fresh child processes under torch.distributed.run): the
JSON schema the driver parses, the `comm` evidence hooks of round 3 (per-bucket bytes / device time, exposed wait time) and the
exchange forms -- `--dp-overlap 0` must report exactly ONE all-reduce per step (north_star's "single RCCL all-reduce").  No
scaling figure is derived from this run: two ranks share one card.  The real backend is RCCL (`nccl`) under the same code path;
tests/test_gpu_rccl_one_rank.py drives every collective through it with a one-rank group."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

TOP_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "comm"}
ROOF_KEYS = {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_launch_ms"}


def _run(extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, VP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-settle",
           "--img", "64", "--z", "32", "--batch-per-gpu", "8"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _check_common(d):
    assert TOP_KEYS <= set(d), TOP_KEYS - set(d)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert ROOF_KEYS <= set(d["roofline"]), ROOF_KEYS - set(d["roofline"])            # same keys as the one-rank line
    c = d["comm"]
    assert c["world_size_seen"] == 2 and c["backend"] == "gloo"
    assert c["traced_steps"] >= 1 and c["exposed_ms_per_step"] >= 0.0
    for b in c["buckets"]:
        assert b["bytes"] > 0 and b["ms"] >= 0.0 and b["kind"] in ("all_reduce", "all_gather") and b["calls_per_step"] == 1
    assert abs(d["value"] - 16 * 4 / (d["ms_per_step"] * 4e-3)) <= 0.02 * d["value"]   # whole-job images/s from the max-over-ranks time
    return c


def test_single_all_reduce_form():
    c = _check_common(_run(["--dp-overlap", "0"]))
    assert c["all_reduces_per_step"] == 1 and c["all_gathers_per_step"] == 0
    assert len(c["buckets"]) == 1 and c["buckets"][0]["name"] == "whole gradient arena"
    # the one bucket is the whole flat gradient arena (fp32)
    import torch
    import vae_play_amd as V
    n = sum((p.numel() + 63) // 64 * 64 for p in V.VAE(64, 32, 3).parameters())
    assert c["buckets"][0]["bytes"] == 4 * n and c["bytes_per_step"] == 4 * n


@pytest.mark.parametrize("factored", [1, 0])
def test_bucketed_forms(factored):
    c = _check_common(_run(["--dp-factored", str(factored)]))
    names = [b["name"] for b in c["buckets"]]
    assert "decoder" in names and any(n.startswith("encoder.conv") for n in names)
    if factored:
        assert c["all_gathers_per_step"] == 2 and "fc.0 factor: dh" in names and "fc.0 factor: flat" in names
        assert "encoder dense (without fc.0)" in names
    else:
        assert c["all_gathers_per_step"] == 0 and "encoder dense" in names
    assert c["all_reduces_per_step"] == len([b for b in c["buckets"] if b["kind"] == "all_reduce"]) >= 3
