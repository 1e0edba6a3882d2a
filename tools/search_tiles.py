"""Coordinate search over the workgroup tile of each split-bf16 conv launch shape (A/B knob VP_TILE_OVERRIDE), each candidate
against the current choice in one process (tools/ab_env.py protocol: the knob is read per launch).
usage: python tools/search_tiles.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VP_ENV_DYNAMIC", "1")      # (inherited by the probe and the ab_env.py children: per-launch knob reads, csrc/env.h)
probe = subprocess.run([sys.executable, "-c", "import os,sys,torch\nsys.path.insert(0, %r)\nimport vae_play_amd as V\nfrom vae_play_amd import optim\nfrom vae_play_amd.engine import FusedVAEStep\ntorch.manual_seed(0)\nvae=V.VAE(128,128,3).cuda()\nopt=optim.Adam(vae.parameters(),lr=1e-4)\nst=FusedVAEStep(vae,opt,32,128,3)\nx,e=torch.rand(32,3,128,128,device='cuda'),torch.randn(32,128,device='cuda')\nst.step(x,e)\nos.environ['VP_TILE_LOG']='1'\nst.step(x,e)\ntorch.cuda.synchronize()" % ROOT],
                       capture_output=True, text=True, timeout=300)
shapes = sorted({m for m in re.findall(r"tile16 (\d+x\d+x\d+)", probe.stderr)})
print("launch shapes:", shapes, flush=True)
best = {}
for key in shapes:
    M, N, gz = (int(v) for v in key.split("x"))
    if gz >= 25:          # weight-gradient family: tiles follow the channel counts
        continue
    for cand in ("128x128", "128x64", "64x64"):
        if (cand == "128x128" and N < 128) or (cand.startswith("128") and M < 128):
            continue
        cfg = ",".join([f"{k}:{v}" for k, v in best.items()] + [f"{key}:{cand}"])
        ref = ",".join([f"{k}:{v}" for k, v in best.items()]) or "none"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_env.py"), "VP_TILE_OVERRIDE", ref, cfg, "3", "30"],
                           capture_output=True, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if "median" in l]
        try:
            a, b = (float(l.split("median")[1].split("ms")[0]) for l in lines[-2:])
        except Exception:  # noqa: BLE001
            print(key, cand, "failed", r.stderr[-200:]); continue
        print(f"{key:18s} {cand:8s} ref {a:.3f} cand {b:.3f} {'<-- better' if b < a * 0.996 else ''}", flush=True)
        if b < a * 0.996:
            best[key] = cand
print("best overrides:", best)
