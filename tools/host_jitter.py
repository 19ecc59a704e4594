#!/usr/bin/env python3
"""Run-to-run spread of the streaming rate inside ONE process: N windows of K frames each (bracketed by device synchronisations),
with the mean host time of a push, for the current environment knobs. host_jitter.py [N=12] [K=2000] [--bind]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 12
K = int(args[1]) if len(args) > 1 else 2000

import numpy as np  # noqa: E402
import torch  # noqa: E402
from rebvio_amd import backend as B, shard, synth  # noqa: E402

node = -2
if "--bind" in sys.argv:
    pr = torch.cuda.get_device_properties(0)
    node = shard.bind_to_gpu_numa_node(f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
W, H = 640, 480
frames, cam = synth.render_stream(W, H, 24)
ctx = B.Context(B.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
dev = ctx.upload_frames(frames)
npx = W * H
order = synth.pingpong_indices(24, 1000 + N * K + 64)
k = 0
for _ in range(1000):
    ctx.push_frame_u8_device(dev + int(order[k]) * npx, k * 50000)
    k += 1
torch.cuda.synchronize()
rates = []
for w in range(N):
    t0 = time.perf_counter()
    for _ in range(K):
        ctx.push_frame_u8_device(dev + int(order[k]) * npx, k * 50000)
        k += 1
    torch.cuda.synchronize()
    rates.append(K / (time.perf_counter() - t0))
ctx.flush()
print("numa node %d | %d windows of %d frames: " % (node, N, K) + " ".join("%.0f" % r for r in rates) + " | cpu now %s" % sorted(os.sched_getaffinity(0))[:4])
