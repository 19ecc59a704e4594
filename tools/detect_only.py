"""Diagnostic: throughput of the detect chain alone (no tracking), frames resident in HBM."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rebvio_amd import backend as B, synth
frames, cam = synth.render_stream(640, 480, 8)
ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
dev = ctx.upload_frames(frames)
npx = 640*480
maps = []
def run(n):
    for k in range(n):
        m = ctx.detect_u8_device(dev + (k % 8) * npx, k * 50000)
        maps.append(m)
        if len(maps) > 3:
            maps.pop(0).release()
run(200)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(2000)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("detect-only: %.1f us/frame, %.0f fps" % ((t1 - t0) / 2000 * 1e6, 2000 / (t1 - t0)))
