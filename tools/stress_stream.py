"""Stress aid: the bench's streaming loop with its profiling phases, repeated; reports any device-side time-out.
stress_stream.py [rounds] [jumps]: jumps = 1 replays the ping-pong order in random steps of 1..6 frames, so that minimizeVel's
accept masks vary and the speculative LM kernel's roll-back path runs under the streaming driver; prints the mask histogram."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rebvio_amd import backend as B, synth

def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    jumps = len(sys.argv) > 2 and int(sys.argv[2]) != 0
    import collections
    masks = collections.Counter()
    statuses = collections.Counter()
    frames, cam = synth.render_stream(640, 480, 24)
    ctx = B.Context(B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000))
    dev = ctx.upload_frames(frames)
    order = synth.pingpong_indices(24, 1 << 20)
    if jumps:
        rng = np.random.default_rng(7)
        order = order[np.cumsum(rng.integers(1, 7, size=1 << 20)) % len(order)]
    k = 0

    def push():
        nonlocal k
        out, _ = ctx.push_frame_u8_device(dev + int(order[k % len(order)]) * 640 * 480, k * 50000)
        k += 1
        if out.status >= 0:
            masks[out.lm_accept_mask] += 1
        statuses[out.status] += 1

    t0 = time.time()
    for r in range(rounds):
        try:
            for _ in range(50):
                push()
            torch.cuda.synchronize()
            ctx.profile_reset(); ctx.profile(True)
            for _ in range(24):
                push()
            torch.cuda.synchronize()
            prof = ctx.profile_read(); ctx.profile(False)
            ctx.profile_reset(); ctx.profile(True, only="k_lm_chain*", stride=8)
            for _ in range(1000):
                push()
            torch.cuda.synchronize()
            ctx.profile_read(); ctx.profile(False)
            ctx.flush()
        except Exception as e:
            print(f"round {r} frame {k}: {e}", flush=True)
            return 1
        if r % 20 == 19 or r == rounds - 1:
            print(f"round {r} ok, {k} frames, {time.time() - t0:.1f} s", flush=True)
    print("accept masks:", {format(m, "05b"): n for m, n in sorted(masks.items(), key=lambda kv: -kv[1])})
    print("statuses:", dict(statuses))
    return 0

sys.exit(main())
