#!/usr/bin/env python3
"""Which kernels change between frame ~1000 and frame ~9000 of a stream (the rate drifts from ~13.9k to ~13.0k frames/s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rebvio_amd import backend as B, shard, synth
pr = torch.cuda.get_device_properties(0)
shard.bind_to_gpu_numa_node(f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
W, H = 640, 480
frames, cam = synth.render_stream(W, H, 24)
ctx = B.Context(B.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
dev = ctx.upload_frames(frames)
npx = W * H
order = synth.pingpong_indices(24, 12000)
k = 0
def run(n):
    global k
    kl, mt = [], []
    for _ in range(n):
        out, nk = ctx.push_frame_u8_device(dev + int(order[k]) * npx, k * 50000)
        k += 1
        if out.status == 0:
            kl.append(nk); mt.append(out.klm_num)
    torch.cuda.synchronize()
    return (np.mean(kl) if kl else 0, np.mean(mt) if mt else 0)
for target in (1000, 3000, 5000, 9000):
    run(target - k)
    t0 = time.perf_counter(); a = run(1000); rate = 1000 / (time.perf_counter() - t0)
    ctx.profile_reset(); ctx.profile(True)
    run(48)
    p = ctx.profile_read(); ctx.profile(False)
    print("frame %d: %.0f frames/s, keylines %.0f, matches %.0f | " % (k, rate, a[0], a[1]) + "  ".join("%s %.1f" % (n.replace("k_", "")[:18], v[0] * v[1] / 48) for n, v in sorted(p.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:12]))
ctx.flush()
