import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rebvio_amd import backend as B, synth
frames, cam = synth.render_stream(640, 480, 24)
ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
dev = ctx.upload_frames(frames)
order = synth.pingpong_indices(24, 20000)
npx = 640*480
k = 0
import subprocess
for blk in range(12):
    if blk == 6:
        print("-- 1 s pause --", flush=True); time.sleep(1.0)
    if blk in (3, 9):
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        print(" | ".join(l.strip() for l in r.stdout.splitlines() if "sclk" in l or "Power" in l or "mclk" in l), flush=True)
    prof = blk in (0, 5)
    if prof:
        ctx.profile_reset(); ctx.profile(True, stride=4)
    t0 = time.perf_counter(); ns = []; ms = []
    for _ in range(1000):
        out, n = ctx.push_frame_u8_device(dev + int(order[k]) * npx, k * 50000); k += 1
        if n >= 0: ns.append(n); ms.append(out.klm_num)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"block {blk}: {1000/dt:.0f} fps  keylines {np.mean(ns):.0f}  matches {np.mean(ms):.0f}", flush=True)
    if prof:
        pr = ctx.profile_read(); ctx.profile(False)
        print("   ", {k: round(v[0], 1) for k, v in sorted(pr.items())}, flush=True)
