"""Per-kernel roofline table from the round-end measurement set (tools/collect_profiles.sh):
    kernel_stats of the serial schedule (avg duration), the FETCH_SIZE / WRITE_SIZE PMC passes (HBM-side bytes per launch, MB =
    (2 x FETCH_SIZE + WRITE_SIZE) KB / 1024, the gfx950 correction of MI355X_MICROARCH.md) and the MFMA-busy PMC pass.
For every kernel: achieved HBM-side GB/s (bytes / duration; FETCH_SIZE counts fabric requests, Infinity-Cache hits included, so this
is an upper bound of true HBM traffic) against the 8 TB/s peak, and MFMA-busy share of SIMD cycles.  The bound column names the
larger of the two fractions' resource.
usage: python profiles/make_roofline.py <prof dir> <steps in the stats run> <steps in the pmc runs> > profiles/<name>.md"""
import csv
import re
import sys
from collections import defaultdict

d, steps, psteps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
HBM_PEAK, SIMDS, XCDS = 8000.0, 1024, 8


def name(n):
    return re.sub(r"^void ", "", n)


dur, calls = {}, {}
for r in csv.DictReader(open(f"{d}/ks_serial/t_kernel_stats.csv")):
    dur[name(r["Name"])] = float(r["AverageNs"]) / 1e3
    calls[name(r["Name"])] = int(r["Calls"]) / steps
fetch, write, nl = defaultdict(float), defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(f"{d}/fetch/t_counter_collection.csv")):
    fetch[name(r["Kernel_Name"])] += float(r["Counter_Value"]); nl[name(r["Kernel_Name"])] += 1
for r in csv.DictReader(open(f"{d}/write/t_counter_collection.csv")):
    write[name(r["Kernel_Name"])] += float(r["Counter_Value"])
busy, act = defaultdict(float), defaultdict(float)
for r in csv.DictReader(open(f"{d}/mfma/t_counter_collection.csv")):
    k = name(r["Kernel_Name"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += float(r["Counter_Value"])
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[k] += float(r["Counter_Value"])
print("# Per-kernel roofline, bf16x3 step, serial schedule (one MI355X, 128x128x3, z = 128, 32 images)\n")
print("Duration: rocprofv3 --kernel-trace --stats; bytes: separate --pmc FETCH_SIZE / WRITE_SIZE passes (x2-corrected fetch, fabric side:\n"
      "Infinity-Cache hits included); MFMA busy: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (durations under PMC are longer, so the share\n"
      "is of the cycles of THAT run).  Peaks: HBM 8 TB/s; bf16 MFMA 2.5 PFLOP/s dense (= 100 % busy).\n")
print("| us/step | launches | avg us | MB/launch (fetch + write) | GB/s | % of HBM peak | MFMA busy % | bound | kernel |")
print("|---:|---:|---:|---:|---:|---:|---:|---|---|")
rows = sorted(dur, key=lambda k: -dur[k] * calls[k])
for k in rows[:34]:
    n = max(nl.get(k, 0), 1)
    mb = (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) / 1024 / n
    gbs = mb / 1024 / (dur[k] * 1e-6) if dur[k] > 0 else 0.0
    mf = 100 * busy[k] / (act[k] / XCDS * SIMDS) if act.get(k, 0) > 0 else 0.0
    hb = 100 * gbs / HBM_PEAK
    bound = "mfma / power" if mf >= 15 else ("hbm" if hb >= 35 else "latency")
    print(f"| {dur[k] * calls[k]:.1f} | {calls[k]:.0f} | {dur[k]:.1f} | {mb:.1f} | {gbs:.0f} | {hb:.0f} | {mf:.1f} | {bound} | `{k[:96]}` |")
