"""Diagnostic: frames/s when every u8 frame is copied host -> device (synchronous hipMemcpy from pageable memory, the
slowest way to hand a frame over) right before it is pushed, against frames resident in HBM (what bench.py's `value` is)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from rebvio_amd import backend as B, synth

frames, cam = synth.render_stream(640, 480, 24)
order = synth.pingpong_indices(24, 5000)
npx = 640 * 480


def run(copy):
    ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
    dev = ctx.upload_frames(frames)
    ring = ctx.upload_frames(np.zeros((8, 480, 640), np.uint8))
    def push(k):
        i = int(order[k])
        if copy:
            dst = ring + (k % 8) * npx
            B._chk(B.lib().rebvio_hip_device_upload(ctx.h, C.c_void_p(dst), frames[i].ctypes.data, npx))
            ctx.push_frame_u8_device(dst, k * 50000)
        else:
            ctx.push_frame_u8_device(dev + i * npx, k * 50000)
    for k in range(100):
        push(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(100, 2100):
        push(k)
    ctx.flush()
    torch.cuda.synchronize()
    return 2000 / (time.perf_counter() - t0)


print("frames resident in HBM : %.0f frames/s" % run(False))
print("H2D copy per frame     : %.0f frames/s" % run(True))
