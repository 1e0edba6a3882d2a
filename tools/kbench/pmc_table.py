"""Per-kernel table from rocprofv3 --pmc CSVs (test/bench infrastructure).
usage: python tools/kbench/pmc_table.py D/..._counter_collection.csv [more.csv ...]
Counters are averaged per launch; SQ_* cycle counters are also shown as a share of SQ_WAVE_CYCLES when that counter
is in the same pass.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles."""
import csv
import re
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    val = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"^void ", "", r["Kernel_Name"])
        k = re.sub(r"vp::", "", k)[:70]
        c = r["Counter_Name"]
        val[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
        if c == list(val[k].keys())[0]:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"## {path}")
    for k in val:
        if "igemm" not in k:
            continue
        d = sorted(dur[k])
        parts = [f"n={len(d)} med_us={d[len(d) // 2]:.1f}"]
        wc = val[k].get("SQ_WAVE_CYCLES", 0.0) / max(1, cnt[k].get("SQ_WAVE_CYCLES", 1))
        for c in val[k]:
            v = val[k][c] / cnt[k][c]
            s = f"{c}={v:.3g}"
            if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and ("CYCLES" in c or "WAIT" in c or "ACTIVE" in c or "CONFLICT" in c):
                s += f"({v / wc:.2f})"
            parts.append(s)
        print(k, "|", " ".join(parts))
