#!/usr/bin/env python3
"""A/B runs of the distance-field build: bench.py under different REBVIO_HIP_DF / REBVIO_HIP_DF_STRIP settings; prints
frames/s and the per-launch time of the distance-field kernel (HIP events, all-kernel pass)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
settings = sys.argv[1:] or ["tiles", "16,1", "16,2", "16,4", "8,2", "8,4", "32,2", "32,4"]
for s in settings:
    env = dict(os.environ)
    if s == "tiles":
        env["REBVIO_HIP_DF"] = "tiles"
    else:
        env["REBVIO_HIP_DF_STRIP"] = s
    extra = os.environ.get("DF_SWEEP_ARGS", "--steps 2000 --no-cpu-baseline").split()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(s, "FAILED", r.stderr[-500:])
        continue
    d = json.loads(r.stdout.strip().splitlines()[-1])
    k = {n: v for n, v in d["kernel_us_per_frame"].items() if n.startswith("k_df")}
    print(f"{s:8s} fps {d['value']:9.1f}  df {k}  detect-stage {d['stage_us']['detect']['us_per_frame']}", flush=True)
