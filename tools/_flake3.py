import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rebvio_amd import backend as B, synth
W, H = 640, 480
frames, cam = synth.render_stream(W, H, 8)
kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000)
rng = np.random.default_rng(3)
order = np.cumsum(rng.integers(1, 6, size=160)) % len(frames)
npx = W * H
def run(mode):
    if mode: os.environ["REBVIO_HIP_LM"] = mode
    else: os.environ.pop("REBVIO_HIP_LM", None)
    ctx = B.Context(B.default_params(H, W, **kw))
    dev = ctx.upload_frames(frames)
    recs = []
    for k, i in enumerate(order):
        out, n = ctx.push_frame_u8_device(dev + int(i) * npx, k * 50000)
        if out.status >= 0: recs.append((out.status, out.lm_accept_mask, tuple(out.Vg), out.klm_num, out.reg_num, n))
    for o, n in ctx.flush():
        recs.append((o.status, o.lm_accept_mask, tuple(o.Vg), o.klm_num, o.reg_num, n))
    ctx.close()
    return recs
ref = run("seq")
print("pairs", len(ref), "masks", sorted({format(m, "05b") for _, m, *_ in ref}))
bad = 0
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["spec3", "", "spec", "mix7"]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    for mode in modes:
        r = run(mode or None)
        if r != ref:
            bad += 1
            d = [i for i, (x, y) in enumerate(zip(r, ref)) if x != y]
            print("MISMATCH it", it, "mode", mode or "auto", "first diffs", d[:5], r[d[0]], ref[d[0]])
print("mismatches:", bad)
