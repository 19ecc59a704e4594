import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from rebvio_amd import backend as B, synth
frames, cam = synth.render_stream(640, 480, 24)
ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
dev = ctx.upload_frames(frames)
order = synth.pingpong_indices(24, 20000)
hs = [collections.Counter() for _ in range(10)]
for k, i in enumerate(order):
    out, _ = ctx.push_frame_u8_device(dev + int(i) * 640 * 480, k * 50000)
    if out.status >= 0:
        hs[min(k // 2000, 9)][format(out.lm_accept_mask, "05b")] += 1
ctx.flush(); ctx.close()
for j, h in enumerate(hs):
    print("frames %5d..%5d" % (j * 2000, j * 2000 + 1999), dict(h))
