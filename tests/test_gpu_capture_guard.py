"""hipGraph capture of the autograd front end is refused up front (ops._stream) instead of aborting the process (VERDICT r2 weak #7:
gpurun_out/be_graph.err).  Runs in a child process: the failure mode being guarded against is a core dump."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_autograd_front_end_refuses_graph_capture():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "capture_guard_selftest.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "refused True" in r.stdout and r.stdout.strip().endswith("OK")
