"""Create / use / destroy many contexts in one process and report the device memory that stays allocated.
    python tools/leak_probe.py [--rounds 100]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=100)
    a = ap.parse_args()
    import torch
    from rebvio_amd import backend as B
    from rebvio_amd import synth
    B.lib()
    frames, cam = synth.render_stream(640, 480, 4)
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000)
    used = []
    for r in range(a.rounds):
        ctx = B.Context(B.default_params(480, 640, **kw))
        dev = ctx.upload_frames(frames)
        for k in range(8):
            ctx.push_frame_u8_device(dev + (k % 4) * 640 * 480, k * 50000)
        ctx.flush()
        ctx.close()
        del ctx
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used.append((total - free) / 2**20)
        if r % 10 == 0:
            print(f"round {r}: {used[-1]:.1f} MiB in use", flush=True)
    print(f"first {used[0]:.1f} MiB, after 10 {used[min(10, len(used) - 1)]:.1f} MiB, last {used[-1]:.1f} MiB")
    return 0 if used[-1] - used[min(10, len(used) - 1)] < 64 else 1


if __name__ == "__main__":
    sys.exit(main())
