#!/usr/bin/env python3
"""Aggregate frames/s of G independent batches of B lanes each in ONE process, one driving thread per batch (diagnostic:
do two half-size lock-step groups hide each other's host hand-over?).  multi_batch.py G B [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rebvio_amd import backend as B, synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
warm = 600
W, H = 640, 480
cam = synth.Camera.for_size(W, H)
bats, devs = [], []
for g in range(G):
    bat = B.Batch(B.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000), L)
    bats.append(bat)
    devs.append([bat.lanes[l].upload_frames(synth.render_stream(W, H, 24, stream_id=g * L + l)[0]) for l in range(L)])
order = synth.pingpong_indices(24, warm + steps + 64)
npx = W * H
bad = [0] * G
def run(g, n0, n):
    for k in range(n0, n0 + n):
        outs, _ = bats[g].push_u8_device([d + int(order[k]) * npx for d in devs[g]], k * 50000)
        bad[g] += sum(1 for o in outs if o.status not in (0, -1))
ths = [threading.Thread(target=run, args=(g, 0, warm)) for g in range(G)]
[t.start() for t in ths]; [t.join() for t in ths]
torch.cuda.synchronize()
t0 = time.perf_counter()
ths = [threading.Thread(target=run, args=(g, warm, steps)) for g in range(G)]
[t.start() for t in ths]; [t.join() for t in ths]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{G} batches x {L} lanes: {G * L * steps / dt:.0f} frames/s aggregate, {dt / steps * 1e6:.1f} us per step, bad statuses {sum(bad)}")
for b in bats:
    b.flush(); b.close()
