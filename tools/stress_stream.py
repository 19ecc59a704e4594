"""Stress aid: the bench's streaming loop with its profiling phases, repeated; reports any device-side time-out."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rebvio_amd import backend as B, synth

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    frames, cam = synth.render_stream(640, 480, 24)
    ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
    dev = ctx.upload_frames(frames)
    order = synth.pingpong_indices(24, 1 << 20)
    k = 0
    t0 = time.time()
    for r in range(rounds):
        try:
            for _ in range(50):
                ctx.push_frame_u8_device(dev + int(order[k % len(order)]) * 640 * 480, k * 50000); k += 1
            torch.cuda.synchronize()
            ctx.profile_reset(); ctx.profile(True)
            for _ in range(24):
                ctx.push_frame_u8_device(dev + int(order[k % len(order)]) * 640 * 480, k * 50000); k += 1
            torch.cuda.synchronize()
            prof = ctx.profile_read(); ctx.profile(False)
            ctx.profile_reset(); ctx.profile(True, only="k_lm_chain<512>", stride=8)
            for _ in range(1000):
                ctx.push_frame_u8_device(dev + int(order[k % len(order)]) * 640 * 480, k * 50000); k += 1
            torch.cuda.synchronize()
            ctx.profile_read(); ctx.profile(False)
            ctx.flush()
        except Exception as e:
            print(f"round {r} frame {k}: {e}", flush=True)
            return 1
        print(f"round {r} ok, {k} frames, {time.time() - t0:.1f} s", flush=True)
    return 0

sys.exit(main())
