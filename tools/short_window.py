#!/usr/bin/env python3
"""The driver's bench window (`bench.py --steps 20 --warmup 5`) repeated: W windows of K frames, each bracketed by
torch.cuda.synchronize() on both sides like bench.py's timed region, after the stream has reached its steady state.
Prints the distribution of the windows' rates next to the rate of one long window, and the host time of the pushes.

    short_window.py [K=20] [W=40] [settle=1000]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_a = [a for a in sys.argv[1:] if not a.startswith("--")]
K = int(_a[0]) if len(_a) > 0 else 20
NW = int(_a[1]) if len(_a) > 1 else 40
settle = int(_a[2]) if len(_a) > 2 else 1000

import numpy as np  # noqa: E402
import torch  # noqa: E402
from rebvio_amd import backend as B, shard, synth  # noqa: E402

if "--no-bind" not in sys.argv:  # like bench.py: the CPUs of the GPU's NUMA node (the rate is host-sensitive: unbound -10..20 % on a two-socket box)
    pr = torch.cuda.get_device_properties(0)
    shard.bind_to_gpu_numa_node(f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")

W, H = 640, 480
frames, cam = synth.render_stream(W, H, 24)
ctx = B.Context(B.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
dev = ctx.upload_frames(frames)
npx = W * H
order = synth.pingpong_indices(24, settle + NW * (K + 5) + 4000 + 64)
k = 0


def push():
    global k
    r = ctx.push_frame_u8_device(dev + int(order[k]) * npx, k * 50000)
    k += 1
    return r


for _ in range(settle):
    push()
torch.cuda.synchronize()
for _ in range(8):
    push()
rates, push_us, pairs = [], [], []
for w in range(NW):
    for _ in range(5):  # the driver's --warmup 5, directly in front of the window
        push()
    torch.cuda.synchronize()
    p0 = ctx.pairs_started()
    t0 = time.perf_counter()
    for _ in range(K):
        push()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rates.append(K / (t2 - t0))
    push_us.append((t1 - t0) / K * 1e6)
    pairs.append(ctx.pairs_started() - p0)
torch.cuda.synchronize()
for _ in range(200):
    push()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3000):
    push()
torch.cuda.synchronize()
long_rate = 3000 / (time.perf_counter() - t0)
ctx.flush()
r = np.array(rates)
print("windows of %d frames x %d: frames/s min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f | host us per push (median) %.1f | pairs started per window %.1f | long window %.0f"
      % (K, NW, r.min(), np.percentile(r, 10), np.median(r), np.percentile(r, 90), r.max(), float(np.median(push_us)), float(np.mean(pairs)), long_rate))
