"""Turn a rocprofv3 --kernel-trace --stats kernel_stats.csv into the per-step markdown table kept in profiles/.
usage: python profiles/make_summary.py <kernel_stats.csv> <steps> <title> > profiles/<name>.md"""
import csv
import re
import sys

path, steps, title = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
print(f"# {title}\n")
print("| us/step | launches/step | avg us | % | kernel |\n|---:|---:|---:|---:|---|")
for r in rows[:32]:
    name = re.sub(r"^void ", "", r["Name"])[:110]
    print(f"| {float(r['TotalDurationNs']) / steps / 1e3:.1f} | {int(r['Calls']) / steps:.0f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{float(r['Percentage']):.1f} | `{name}` |")
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"\nSum of kernel time per step: {tot:.2f} ms.")
