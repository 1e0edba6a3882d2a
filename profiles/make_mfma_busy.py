"""MFMA-busy share per kernel from one rocprofv3 PMC pass (serial schedule):
    VP_SIDE_WGRAD=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d D -o t -- \
        python3 bench.py --steps 2 --warmup 1 --no-settle --no-cpu-baseline
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 1024 SIMDs).  SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over
all SIMDs (32 per v_mfma_f32_32x32x16_bf16: it reproduces the algorithmic count exactly, 157.29 M for a 53.69-GFLOP layer with
three MFMAs per product); GRBM_GUI_ACTIVE comes back SUMMED over the 8 XCDs (17-18 G cycles per second of kernel time), so the
active cycles of the chip are that sum / 8 -- rocprofv3's own MfmaUtil formula takes the max over instances instead.
usage: python profiles/make_mfma_busy.py D/t_counter_collection.csv > profiles/<name>.md"""
import csv
import re
import sys
from collections import defaultdict

SIMD_NUM, XCDS = 256 * 4, 8
busy, act, dur, n = defaultdict(float), defaultdict(float), defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"^void ", "", r["Kernel_Name"])
    v = float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += v
        n[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[k] += v
print("# MFMA-busy share per kernel (rocprofv3 PMC: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); serial schedule,\n"
      "# `bench.py --steps 2 --warmup 1 --no-settle`, counters summed over all launches of a kernel; durations under PMC collection)\n")
print("| kernel | launches | MFMA busy % of SIMD cycles while the kernel runs | avg us (under PMC) |\n|---|---:|---:|---:|")
rows = sorted(busy, key=lambda k: -busy[k])
tb = sum(busy.values()); ta = sum(act.values())
for k in rows[:18]:
    if act[k] > 0:
        print(f"| `{k[:100]}` | {n[k]} | {100 * busy[k] / (act[k] / XCDS * SIMD_NUM):.1f} | {dur[k] / n[k]:.1f} |")
print(f"\nWhole step (all kernels): MFMA busy {100 * tb / (ta / XCDS * SIMD_NUM):.1f} % of GPU-active cycles x SIMDs.")
