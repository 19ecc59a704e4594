#!/usr/bin/env python3
"""Frames/s of rebvio_hip_batch_* for B lanes on one GPU: batch_rate.py B [steps] [warmup] [--prof]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 800
prof = "--prof" in sys.argv

import torch  # noqa: E402,F401
from rebvio_amd import backend as B, shard, synth  # noqa: E402

if "--no-bind" not in sys.argv:  # like bench.py: the CPUs of the GPU's NUMA node
    _pr = torch.cuda.get_device_properties(0)
    shard.bind_to_gpu_numa_node(f"{_pr.pci_domain_id:04x}:{_pr.pci_bus_id:02x}:{_pr.pci_device_id:02x}.0")

W, H = 640, 480
cam = synth.Camera.for_size(W, H)
bat = B.Batch(B.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000), Bn)
devs = [bat.lanes[l].upload_frames(synth.render_stream(W, H, 24, stream_id=l)[0]) for l in range(Bn)]
order = synth.pingpong_indices(24, warm + steps + 64)
npx = W * H
k = 0
for _ in range(warm):
    bat.push_u8_device([d + int(order[k]) * npx for d in devs], k * 50000)
    k += 1
torch.cuda.synchronize()
if prof:
    c0 = bat.lanes[0]
    c0.profile_reset()
    c0.profile(True)
    for _ in range(24):
        bat.push_u8_device([d + int(order[k]) * npx for d in devs], k * 50000)
        k += 1
    torch.cuda.synchronize()
    pr = c0.profile_read()
    c0.profile(False)
    print({n: round(v[0] * v[1] / 24, 1) for n, v in sorted(pr.items(), key=lambda kv: -kv[1][0] * kv[1][1])})
bad = 0
kl = []
t0 = time.perf_counter()
for _ in range(steps):
    outs, nks = bat.push_u8_device([d + int(order[k]) * npx for d in devs], k * 50000)
    bad += sum(1 for l in range(Bn) if outs[l].status not in (0, -1))
    kl.append(nks[0])
    k += 1
torch.cuda.synchronize()
t1 = time.perf_counter()
bat.flush()
print(f"lanes {Bn}: {Bn * steps / (t1 - t0):.0f} frames/s aggregate, {(t1 - t0) / steps * 1e6:.1f} us per step, keylines(lane 0) {kl[-1]}, bad statuses {bad}")
