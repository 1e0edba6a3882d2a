"""The pipelined LDS-DMA kernel (csrc/igemm16p.h) against the register-staged kernel (csrc/igemm16.h), through the stand-alone
bench tools/kbench: the 32x32x16 forms must agree BIT FOR BIT on every launch shape (same K order, same hi/lo MFMA sequence),
the 16x16x32 form to rounding (sums run over k in groups of 32 instead of 16)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KBENCH = os.path.join(ROOT, "tools", "kbench", "kbench")


@pytest.mark.parametrize("args", [["B=8", "img=128"], ["B=16", "img=64"], ["B=4", "img=256", "layers=enc4,dec0,dec4,enc1"]])
def test_pipelined_kernel_equals_register_staged_kernel(args):
    if not os.path.exists(KBENCH):
        pytest.fail("tools/kbench/kbench is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([KBENCH, *args, "reps=1", "mode=all", "buf=2", "m16=2", "v1=0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "out-of-range LDS-DMA lanes write ZEROS" in out.stdout, "buffer_load ... lds must zero-fill out-of-range lanes (padding taps)"
    cells = re.findall(r"c\d+[bq]? \d+x\d+ +[\d.]+ us +[\d.]+ TF (bit=\d+|rel=[\d.e+-]+)( !!!)?", out.stdout)
    assert len(cells) >= 20, out.stdout[-2000:]
    for what, bad in cells:
        assert not bad, f"kernel mismatch: {what}\n" + out.stdout[-3000:]
        if what.startswith("bit="):
            assert what == "bit=0", what


SBENCH = os.path.join(ROOT, "tools", "kbench", "sbench")


def test_two_phase_halo_patch_prototype_equals_library_kernel():
    """tools/kbench/scatter5.h (prototype, not in the library: transposed-conv forward with two phases per workgroup from one halo patch)
    against the library's vp_conv5_scatter_bf16x3 on every layer shape it takes: equal to rounding (another summation order over k)."""
    if not os.path.exists(SBENCH):
        pytest.fail("tools/kbench/sbench is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run([SBENCH, "B=4", "img=128", "reps=1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    cells = re.findall(r"new vs old: max\|d\|/rms = ([\d.e+-]+), (\d+) NaN( !!!)?", out.stdout)
    assert len(cells) >= 5, out.stdout[-2000:]
    for rel, nan, bad in cells:
        assert not bad and int(nan) == 0 and float(rel) <= 3e-5, out.stdout[-3000:]
