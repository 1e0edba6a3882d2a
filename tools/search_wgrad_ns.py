"""Coordinate search over the per-shape split counts of the split-bf16 weight-gradient kernels (A/B knob VP_WGRAD_NS), each
candidate against the current heuristic in one process (tools/ab_build.py protocol).  usage: python tools/search_wgrad_ns.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VP_ENV_DYNAMIC", "1")
# base = the heuristic's current split count per "CsxCb" key (tap-pair layers: before the pair rule doubles it)
shapes = {"128x64": 16, "256x128": 12, "512x256": 3, "512x512": 2}
best = {}
for key, base in shapes.items():
    for cand in sorted({max(1, int(round(base * f))) for f in (0.5, 0.67, 1.34, 1.5, 2.0)} - {base}):
        cfg = ",".join([f"{k}:{v}" for k, v in best.items()] + [f"{key}:{cand}"])
        ref = ",".join([f"{k}:{v}" for k, v in best.items()]) or "none:1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), "VP_WGRAD_NS", ref, cfg, "3", "30"],
                           capture_output=True, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if "median" in l]
        print(key, cand, "|", " || ".join(l.strip() for l in lines), flush=True)
        try:
            a, b = (float(l.split("median")[1].split("ms")[0]) for l in lines[-2:])
            if b < a * 0.996:
                best[key] = cand
        except Exception as exc:  # noqa: BLE001
            print("parse error", exc)
print("best overrides:", best)
