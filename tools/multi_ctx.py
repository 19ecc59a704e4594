#!/usr/bin/env python3
"""B camera streams on ONE GPU inside one process: B contexts (three HIP streams each), one pushing thread per context
(ctypes releases the GIL inside the library). Prints per-stream and aggregate frames/s.
  multi_ctx.py B [steps] [warmup]     env GPU_MAX_HW_QUEUES is forwarded to the runtime (must be set before HIP loads)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B_ = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 600

import torch  # noqa: E402,F401
from rebvio_amd import backend as B, synth  # noqa: E402

W, H = 640, 480
ctxs, devs, orders = [], [], []
for s in range(B_):
    frames, cam = synth.render_stream(W, H, 24, stream_id=s)
    ctx = B.Context(B.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
    devs.append(ctx.upload_frames(frames))
    ctxs.append(ctx)
    orders.append(synth.pingpong_indices(24, warm + steps + 8))
npx = W * H
barrier = threading.Barrier(B_ + 1)
t_end = [0.0] * B_
bad = [0] * B_


def run(s):
    ctx, dev, order = ctxs[s], devs[s], orders[s]
    k = 0
    for _ in range(warm):
        ctx.push_frame_u8_device(dev + int(order[k]) * npx, k * 50000)
        k += 1
    barrier.wait()
    barrier.wait()
    for _ in range(steps):
        out, n = ctx.push_frame_u8_device(dev + int(order[k]) * npx, k * 50000)
        if out.status not in (0, -1):
            bad[s] += 1
        k += 1
    ctx.flush()
    t_end[s] = time.perf_counter()


th = [threading.Thread(target=run, args=(s,)) for s in range(B_)]
for t in th:
    t.start()
barrier.wait()
torch.cuda.synchronize()
t0 = time.perf_counter()
barrier.wait()
for t in th:
    t.join()
torch.cuda.synchronize()
t1 = time.perf_counter()
per = [steps / (te - t0) for te in t_end]
print(f"B={B_} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}: aggregate {B_ * steps / (t1 - t0):.0f} frames/s; per stream "
      f"{[round(p) for p in per]}; non-zero statuses {bad}")
