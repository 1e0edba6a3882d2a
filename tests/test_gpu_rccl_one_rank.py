"""The data-parallel step's collectives on the real RCCL backend: a one-rank `nccl` group with VP_DP_FORCE=1 runs the
three bucketed all-reduces and the factored exchange (all-gather x2 + local GEMM) against the side-stream schedule and
must reproduce the single-process step exactly (tools/dp_rccl_selftest.py).  Several ranks: tests/test_parallel_gloo.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_dp_collectives_on_rccl_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_rccl_selftest.py"), "5"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "max_rel_param_diff_vs_single_after_3_steps" in r.stdout
