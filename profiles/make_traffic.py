"""HBM-side traffic per kernel from two rocprofv3 PMC passes (separate runs, as MI355X_MICROARCH.md prescribes):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  --output-format csv -d A -o t -- python3 bench.py --steps 2 --warmup 1 --no-settle --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  --output-format csv -d B -o t -- python3 bench.py ...   (VP_SIDE_WGRAD=0: serial schedule)
usage: python profiles/make_traffic.py A/t_counter_collection.csv B/t_counter_collection.csv <steps incl. warm-up> <out.json> > out.md
bytes = 2 * FETCH_SIZE * 1024 (gfx950 reports half of wide streaming reads) + WRITE_SIZE * 1024."""
import csv
import json
import re
import sys
from collections import defaultdict

fa, fb, steps, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def load(path):
    d = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = re.sub(r"^void ", "", r["Kernel_Name"])
        d[k][0] += float(r["Counter_Value"])
        d[k][1] += 1
    return d


F, W = load(fa), load(fb)
# kernels behind each entry point: the register-staged igemm16_kernel, the pipelined igemm16p_kernel (PF16 / PT16) and, for the weight
# gradient, the rows-of-taps wgrad5_kernel (round 3)
fam = {"vp_conv5_gather_bf16x3": ("ProbF16", "PF16"), "vp_conv5_scatter_bf16x3": ("ProbT16", "PT16"),
       "vp_conv5_wgrad_bf16x3": ("ProbW16", "wgrad5_kernel")}
res = {"_provenance": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, serial schedule VP_SIDE_WGRAD=0) of "
       "`python bench.py --steps 2 --warmup 1 --no-settle --no-cpu-baseline` on one MI355X; bytes = 2*FETCH_SIZE*1024 (gfx950 reports "
       "half of wide streaming reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE*1024, averaged over the launches of the igemm16 / "
       "igemm16p / wgrad5 kernels of each family (for the weight gradient: main kernel; its slab reduction is listed separately in the "
       ".md); "
       "FETCH_SIZE counts fabric requests (Infinity-Cache hits included).", "unit": "bytes per launch"}
for ep, tags in fam.items():
    hit = lambda k: any(t in k for t in tags) and ("igemm16" in k or "wgrad5" in k)
    f = sum(v[0] for k, v in F.items() if hit(k))
    n = sum(v[1] for k, v in F.items() if hit(k))
    w = sum(v[0] for k, v in W.items() if hit(k))
    res[ep] = int((2 * f + w) * 1024 / max(n, 1))
tf, tw = sum(v[0] for v in F.values()), sum(v[0] for v in W.values())
res["whole_step_bytes"] = {"fetch_x2": int(2 * tf * 1024 / steps), "write": int(tw * 1024 / steps)}
json.dump(res, open(out_json, "w"), indent=1)
print("# HBM-side traffic per launch (rocprofv3 PMC, separate passes; MB, FETCH_SIZE x2-corrected)\n")
print("| kernel | launches | fetch MB/launch | write MB/launch |\n|---|---:|---:|---:|")
rows = sorted(F.items(), key=lambda kv: -(2 * kv[1][0] + W.get(kv[0], [0, 1])[0]))
for k, (v, n) in rows[:26]:
    w = W.get(k, [0.0, 1])
    print(f"| `{k[:90]}` | {n} | {2 * v * 1024 / n / 1e6:.1f} | {w[0] * 1024 / max(w[1], 1) / 1e6:.1f} |")
print(f"\nWhole step: {2 * tf * 1024 / steps / 1e9:.2f} GB fetched + {tw * 1024 / steps / 1e9:.2f} GB written.")
