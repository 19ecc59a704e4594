#!/usr/bin/env python3
"""Distance-field build alone on an idle GPU: detect the bench's frames, then rebuild one map's field N times
(REBVIO_HIP_DF_FORCE) under the per-kernel event profiler. Usage: df_probe.py [S,XB | tiles] ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    for s in sys.argv[1:]:
        env = dict(os.environ, REBVIO_HIP_DF_FORCE="1")
        if s == "tiles":
            env["REBVIO_HIP_DF"] = "tiles"
        else:
            env["REBVIO_HIP_DF_STRIP"] = s
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(f"{s:8s}", r.stdout.strip() or r.stderr[-400:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from rebvio_amd import backend as B, synth  # noqa: E402

cfgs = [(640, 480, 15000, 16000, 1.0), (1280, 960, 60000, 64000, 2.2)] if os.environ.get("DF_PROBE_C3") else [(640, 480, 15000, 16000, 1.0)]
for W, H, kref, kmax, dens in cfgs:
    frames, cam = synth.render_stream(W, H, 12, density=dens)
    ctx = B.Context(B.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=kref, keylines_max=kmax))
    maps = [ctx.detect_u8(frames[i % 12], i * 50000) for i in range(40 if W == 640 else 120)]  # the servo settles
    m = maps[-1]
    n = m.size()
    for _ in range(20):
        ctx.build_distance_field(m)
    torch.cuda.synchronize()
    ctx.profile_reset()
    ctx.profile(True)
    for _ in range(200):
        ctx.build_distance_field(m)
    torch.cuda.synchronize()
    pr = ctx.profile_read()
    ctx.profile(False)
    print(f"{W}x{H} n={n}", {k: round(v[0], 2) for k, v in pr.items() if k.startswith("k_df")}, end="  ")
print()
