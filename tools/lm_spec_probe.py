#!/usr/bin/env python3
"""Which pair steps give minimizeVel accept masks other than 00001, and do the speculative and the sequential persistent
kernels agree on them bit for bit? Sweeps frame skips / parameter variations on the synthetic streams (stand-alone pairs)."""
import collections, os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def run(mode):
    os.environ["REBVIO_HIP_LM"] = mode
    import torch  # noqa
    from rebvio_amd import backend as B, synth
    res = {}
    for sid in (0, 1, 3):
        frames, cam = synth.render_stream(640, 480, 12, stream_id=sid)
        for variant, kw in (("default", {}), ("rw0.5", dict(reweight_distance=0.5)), ("sr10", dict(search_range=10.0)),
                            ("it7", dict(iterations=7)), ("it3", dict(iterations=3))):
            try:
                P = B.default_params(480, 640, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=15000, keylines_max=16000, **kw)
            except TypeError as e:
                print("variant", variant, "unsupported:", e, file=sys.stderr); continue
            ctx = B.Context(P)
            for skip in (1, 2, 4, 7):
                for i in range(skip, len(frames)):
                    # fresh maps per pair: forwardMatch keys and depth state of a map belong to ONE pair step
                    maps = [ctx.detect_u8(frames[i - skip], 0), ctx.detect_u8(frames[i], skip * 50000)]
                    o = ctx.track_pair(maps[0], maps[1])
                    v = np.concatenate([np.array(o.Vg), np.array(o.P_Vg), [o.F, o.sigma_rho_min], np.array(o.Xv), np.array(o.W_Xv),
                                        [o.klm_num, o.kf_matches, o.reg_num, o.lm_accept_mask, o.status]]).astype(np.float32)
                    res[f"{sid}/{variant}/{skip}/{i}"] = (int(o.lm_accept_mask), v.tobytes().hex())
                    for m in maps:
                        m.release()
            ctx.close()
    return res

if len(sys.argv) > 1:
    json.dump(run(sys.argv[1]), open(sys.argv[2], "w"))
else:
    out = {}
    for mode in ("seq", "spec"):
        f = f"/tmp/lm_probe_{mode}.json"
        subprocess.run([sys.executable, __file__, mode, f], check=True)
        out[mode] = json.load(open(f))
    h = collections.Counter(format(m, "07b") for m, _ in out["seq"].values())
    print("masks (seq):", dict(h))
    bad = [k for k in out["seq"] if out["seq"][k] != out["spec"][k]]
    print("pairs:", len(out["seq"]), "differing:", len(bad), bad[:10])
    other = [k for k, (m, _) in out["seq"].items() if m != 1]
    print("pairs with mask != 1:", other[:40])
