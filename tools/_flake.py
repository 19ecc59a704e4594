import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from rebvio_amd import backend as B, synth
W, H = 192, 144
frames, cam = synth.render_stream(W, H, 12, density=1.0)
other, _ = synth.render_stream(W, H, 1, stream_id=7)
kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=1500, keylines_max=2500, global_min_matches_threshold=900)
seq = np.concatenate([frames[:3], other, frames[3:12]])
npx = W * H
def run(mode):
    if mode: os.environ["REBVIO_HIP_LM"] = mode
    else: os.environ.pop("REBVIO_HIP_LM", None)
    ctx = B.Context(B.default_params(H, W, **kw))
    dev = ctx.upload_frames(seq)
    recs = []
    rng = np.random.default_rng(hash(mode) & 0xFFFF if mode else 0)
    for i in range(len(seq)):
        if mode and mode != "seq" and rng.random() < 0.5:
            t_end = time.perf_counter() + rng.random() * 4e-4
            while time.perf_counter() < t_end:
                pass
        out, n = ctx.push_frame_u8_device(dev + i * npx, i * 50000)
        if out.status >= 0: recs.append((out.status, out.lm_accept_mask, tuple(out.Vg)))
    for o, n in ctx.flush():
        recs.append((o.status, o.lm_accept_mask, tuple(o.Vg)))
    ctx.close()
    return recs
ref = run("seq")
print("seq:", [(s, format(m, "05b")) for s, m, _ in ref])
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    for mode in ("mix%d" % (it + 1), None):
        os.environ["REBVIO_HIP_GROUP"] = str(1 + it % 6)
        os.environ["REBVIO_HIP_LEAD"] = str(3 + (it // 6) % 5)
        r = run(mode)
        if r != ref:
            bad += 1
            print("MISMATCH", it, mode, [(s, format(m, "05b")) for s, m, _ in r])
print("mismatches:", bad)
