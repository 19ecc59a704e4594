#!/usr/bin/env python3
"""Histogram of minimizeVel's accept masks (bit i = LM iteration i accepted, core.cpp:172-183) over a long stream."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rebvio_amd import backend as B, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for sid in (0, 3):
    frames, cam = synth.render_stream(640, 480, 24, stream_id=sid)
    ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
    dev = ctx.upload_frames(frames)
    order = synth.pingpong_indices(24, n)
    h = collections.Counter()
    for k, i in enumerate(order):
        out, _ = ctx.push_frame_u8_device(dev + int(i) * 640 * 480, k * 50000)
        if out.status >= 0:
            h[format(out.lm_accept_mask, "05b")] += 1
    ctx.flush(); ctx.close()
    print("stream", sid, dict(h))
