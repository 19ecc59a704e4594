#!/usr/bin/env python3
"""Statuses / accept masks of the pairs around a rebvio_hip_flush() in the middle of a stream (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rebvio_amd import backend as B, synth
frames, cam = synth.render_stream(640, 480, 24)
ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
dev = ctx.upload_frames(frames)
order = synth.pingpong_indices(24, 400)
k = 0
for seg in range(3):
    rec = []
    for _ in range(40):
        out, n = ctx.push_frame_u8_device(dev + int(order[k]) * 640 * 480, k * 50000)
        k += 1
        if out.status >= 0:
            rec.append((out.status, format(out.lm_accept_mask, "05b"), out.klm_num, round(float(out.F), 3), round(float(out.sigma_rho_min), 5), [round(float(v), 6) for v in out.Vg], n))
    ctx.flush()
    print("segment", seg, rec[:3])
