"""Randomised differential run of the HIP path against the CPU oracle (run on the GPU box; test infrastructure like tests/).

Per trial: a random frame size (incl. widths/heights that are multiples of no tile size), scene seed, texture density, keyline
budget (so that truncation at keylines_max and the threshold servo both get exercised) and velocity; every stage whose result
is specified bit-exactly (tests/test_parity_gpu.py header) is compared bit for bit:
  detection (all keyline fields, dense mask, servo + auto threshold) on every frame, distance field, rotateKeylines +
  estimateQuantile, tryVel per-keyline results, forwardMatch, directedMatch, regularize1Iter, depth EKF.

    python tools/fuzz_parity.py --trials 60 --seed 1 [--out gpurun_out/fuzz.json]
Exit status 1 on the first mismatch (the trial's parameters are printed so that it can be replayed with --only).

--stateful: whole streams instead of stages. Per trial 8-24 frames of a random stream, the state carried INDEPENDENTLY on
both sides, the oracle adding its keyline sums in the kernels' order (oracle_py.Oracle.set_sum_order("device"), a diagnostic
of the restatement): every word of every pair record through the per-pair API and through the streaming driver, and every
keyline field of the newest map after the last pair, bit for bit (tests/test_parity_gpu.py::
test_whole_pipeline_is_bit_identical_with_the_sums_in_one_order on random sizes, densities and budgets).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SIZES = [(640, 480), (320, 240), (752, 480), (420, 293), (192, 144), (256, 64), (132, 257), (1280, 960), (96, 80), (1000, 75)]


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def keylines_equal(ko, kg, what):
    if len(ko) != len(kg):
        raise AssertionError(f"{what}: size {len(ko)} vs {len(kg)}")
    for f in ko.dtype.names:
        if not bits_equal(ko[f], kg[f]):
            bad = np.nonzero((ko[f] != kg[f]).reshape(len(ko), -1).any(1))[0]
            raise AssertionError(f"{what}: field {f} differs at {len(bad)} keylines, first {bad[:5]}: {ko[f][bad[:3]]} vs {kg[f][bad[:3]]}")


def trial(O, B, synth, t):
    W, H = t["size"]
    frames, cam = synth.render_stream(W, H, t["frames"], stream_id=t["stream"], density=t["density"])
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=t["kref"], keylines_max=t["kmax"])
    orc = O.Oracle(O.default_params(H, W, **kw))
    ctx = B.Context(B.default_params(H, W, **kw))
    oms, gms = [], []
    stats = {}
    for i in range(t["frames"]):
        om = orc.detect_u8(frames[i], i * 50000)
        gm = ctx.detect_u8(frames[i], i * 50000)
        keylines_equal(om.keylines(), gm.keylines(), f"detect frame {i}")
        assert np.array_equal(om.mask(H, W), gm.mask()), f"mask frame {i}"
        thr, auto, cnt = ctx.detector_state()
        assert np.float32(thr) == np.float32(orc.threshold) and np.float32(auto) == np.float32(orc.auto_threshold), "servo"
        assert cnt == om.size()
        oms.append(om)
        gms.append(gm)
        if len(oms) > 2:
            oms.pop(0)
            gms.pop(0).release()
        if i >= 1 and i < t["frames"] - 1:
            orc.track_pair(oms[0], oms[1])  # realistic depths / matches for the pair under test
    stats["keylines"] = oms[1].size()
    if oms[0].size() < 8 or oms[1].size() < 8:
        return stats
    om_old, om_new = oms
    gm_old, gm_new = gms
    gm_old.upload(om_old.keylines())
    gm_new.upload(om_new.keylines())
    # distance field of the new map
    orc.build_distance_field(om_new)
    ctx.build_distance_field(gm_new)
    ido, dso = orc.distance_field()
    idg, dsg = ctx.distance_field()
    assert np.array_equal(ido, idg), "distance field ids"
    assert np.array_equal(dso[ido >= 0], dsg[idg >= 0]), "distance field distances"
    # rotateKeylines + estimateQuantile
    a, b = t["rot"]
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    R = (R @ np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]], np.float32)).astype(np.float32)
    orc.rotate(om_old, R)
    ctx.rotate(gm_old, R)
    keylines_equal(om_old.keylines(), gm_old.keylines(), "rotate")
    for pct, bins in ((0.9, 100), (0.5, 37)):
        assert np.float32(orc.quantile(om_old, pct, bins)) == np.float32(ctx.quantile(gm_old, pct, bins)), "quantile"
    # tryVel: per-keyline outputs
    n = om_old.size()
    srm = orc.quantile(om_old)
    res_o = np.zeros(n, np.float32)
    res_g = np.zeros(n, np.float32)
    for vel in ([0, 0, 0], t["vel"]):
        orc.try_vel(om_old, vel, srm, res_o)
        ctx.try_vel(gm_old, vel, srm, res_g)
        assert np.array_equal(om_old.keylines()["match_id_forward"], gm_old.keylines()["match_id_forward"]), "tryVel match_id_forward"
        assert bits_equal(res_o, res_g), f"tryVel residuals differ at {(res_o != res_g).sum()}"
    # minimizeVel on the oracle, forwardMatch on both from the oracle's state
    ro = orc.minimize_vel(om_old)
    gm_old.upload(om_old.keylines())
    orc.forward_match(om_old, om_new)
    ctx.forward_match(gm_old, gm_new)
    keylines_equal(om_new.keylines(), gm_new.keylines(), "forwardMatch")
    V, Rvel = ro["vel"], ro["Rvel"]
    if not (np.isfinite(V).all() and np.isfinite(Rvel).all()):
        V = np.asarray(t["vel"], np.float32)
        Rvel = np.eye(3, dtype=np.float32) * 1e-4
    c = t["rot"][0] * 0.25
    Rb = np.array([[np.cos(c), 0, np.sin(c)], [0, 1, 0], [-np.sin(c), 0, np.cos(c)]], np.float32)
    no, kfo = orc.directed_match(om_new, om_old, V, Rvel, Rb)
    ng, kfg = ctx.directed_match(gm_new, gm_old, V, Rvel, Rb)
    assert (no, kfo) == (ng, kfg), f"directedMatch counts {(no, kfo)} vs {(ng, kfg)}"
    keylines_equal(om_new.keylines(), gm_new.keylines(), "directedMatch")
    assert orc.regularize(om_new) == ctx.regularize(gm_new), "regularize count"
    keylines_equal(om_new.keylines(), gm_new.keylines(), "regularize")
    orc.update_inverse_depth(V)
    ctx.update_inverse_depth(V)
    keylines_equal(om_new.keylines(), gm_new.keylines(), "depth EKF")
    stats["matches"] = int(no)
    return stats


def record_words(po):
    out = []
    for name, _ in type(po)._fields_:
        v = getattr(po, name)
        a = np.array(v) if hasattr(v, "__len__") else np.array([v])
        if a.dtype.kind == "f":
            a = a.astype(np.float32)
            a[np.isnan(a)] = np.float32(np.nan)  # (a NaN is a NaN: sign and payload of one are not part of the result)
            out.append(a.view(np.uint32))
        else:
            out.append(a.astype(np.int64).astype(np.uint32))
    return np.concatenate(out)


def trial_stream(O, B, synth, t):
    W, H = t["size"]
    nf = t["stream_frames"]
    frames, cam = synth.render_stream(W, H, 8, stream_id=t["stream"], density=t["density"])
    order = synth.pingpong_indices(8, nf)
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=t["kref"], keylines_max=t["kmax"])
    orc = O.Oracle(O.default_params(H, W, **kw))
    orc.set_sum_order("device")
    gpu = B.Context(B.default_params(H, W, **kw))
    mo, mg, rec = [], [], []
    statuses = set()
    for k, i in enumerate(order):
        mo.append(orc.detect_u8(frames[i], k * 50000))
        mg.append(gpu.detect_u8(frames[i], k * 50000))
        if len(mo) > 2:
            mo.pop(0)
            mg.pop(0).release()
        if k == 0:
            continue
        po = orc.track_pair(mo[0], mo[1])
        pg = gpu.track_pair(mg[0], mg[1])
        wo, wg = record_words(po), record_words(pg)
        bad = np.flatnonzero(wo != wg)
        assert bad.size == 0, (f"pair {k}: record words {bad[:8]} differ: {wo[bad[:8]].view(np.float32)} vs {wg[bad[:8]].view(np.float32)} "
                               f"(W_Xv diagonal {np.array(po.W_Xv).reshape(6, 6).diagonal()})")
        rec.append(wo)
        statuses.add(int(po.status))
    keylines_equal(mo[1].keylines(), mg[1].keylines(), "newest map after the last pair")
    stats = dict(pairs=len(rec), keylines=mo[1].size(), matches=int(po.klm_num), statuses=sorted(statuses))
    gpu.close()
    ctx = B.Context(B.default_params(H, W, **kw))
    dev = ctx.upload_frames(frames)
    got = []
    for k, i in enumerate(order):
        out, _ = ctx.push_frame_u8_device(dev + int(i) * W * H, k * 50000)
        if out.status >= 0:
            got.append(record_words(out))
    for out, _ in ctx.flush():
        got.append(record_words(out))
    ctx.close()
    assert len(got) == len(rec), f"streaming driver: {len(got)} records for {len(rec)} pairs"
    for k, (wo, wg) in enumerate(zip(rec, got)):
        bad = np.flatnonzero(wo != wg)
        assert bad.size == 0, (f"streaming driver, pair {k + 1}: record words {bad[:8]} differ: "
                               f"{wo[bad[:8]].view(np.float32)} vs {wg[bad[:8]].view(np.float32)}")
    return stats


def make_trial(rng, k, stateful=False):
    W, H = SIZES[int(rng.integers(0, len(SIZES)))] if k % 7 else (640, 480)
    if stateful and (W, H) == (1000, 75):
        # A 1000 x 75 strip leaves extRotVel's 6x6 system nearly singular (translation along the strip against rotation about the
        # axis across it: |X| ~ 3 where a frame gives 1e-2). There the two stand-ins for the reference's SVD back-substitution -
        # the restatement's Jacobi pseudo-inverse, the library's LDL^T in double with that pseudo-inverse as its fall-back -
        # no longer round to the same floats (observed: 5e-6 of |X|, seed 11 trial 9), and a stream's states part from that
        # pair on. The stage-wise fuzz keeps the size; whole streams are compared where the pose is observable.
        W, H = 420, 293
    if (W, H) == (1280, 960) and rng.random() < 0.5:
        W, H = 640, 480
    px = W * H
    base = max(64, int(px * 0.05))
    kref = int(base * rng.uniform(0.3, 1.2))
    kmax = min(int(kref * rng.uniform(1.02, 1.5)), 65536)  # rebvio_hip_create: keylines_max <= 65536
    kref = min(kref, kmax)
    return dict(size=(W, H), frames=int(rng.integers(4, 7)), stream=int(rng.integers(0, 1 << 20)), density=float(rng.uniform(0.4, 2.6)),
                kref=kref, kmax=kmax, rot=(float(rng.normal(0, 0.004)), float(rng.normal(0, 0.003))),
                vel=[float(x) for x in rng.normal(0, 0.012, 3)], stream_frames=int(rng.integers(8, 25)) if (W, H) != (1280, 960) else 8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="replay one trial index")
    ap.add_argument("--out", default="")
    ap.add_argument("--stateful", action="store_true", help="whole streams against the oracle with device-ordered sums (see the header)")
    a = ap.parse_args()
    import torch  # noqa: F401
    from oracle import oracle_py as O
    from rebvio_amd import backend as B
    from rebvio_amd import synth
    O.build()
    O.lib()
    B.lib()
    rng = np.random.default_rng(a.seed)
    trials = [make_trial(rng, k, a.stateful) for k in range(a.trials)]
    log = []
    t0 = time.time()
    for k, t in enumerate(trials):
        if a.only >= 0 and k != a.only:
            continue
        try:
            st = trial_stream(O, B, synth, t) if a.stateful else trial(O, B, synth, t)
        except AssertionError as e:
            print(f"trial {k} FAILED: {e}\n  parameters: {json.dumps(t)}", flush=True)
            if a.out:
                json.dump({"failed": k, "error": str(e), "trial": t, "passed": log}, open(a.out, "w"), indent=1)
            return 1
        log.append(dict(trial=k, size=t["size"], density=round(t["density"], 2), kref=t["kref"], kmax=t["kmax"], **st))
        print(f"trial {k:3d} ok  {t['size'][0]}x{t['size'][1]} density {t['density']:.2f} kref {t['kref']} kmax {t['kmax']} -> {st}  "
              f"[{time.time() - t0:.0f} s]", flush=True)
    if a.out:
        json.dump({"failed": None, "passed": log, "seed": a.seed}, open(a.out, "w"), indent=1)
    print(f"{len(log)} {'streams' if a.stateful else 'trials'} bit-exact")
    return 0


if __name__ == "__main__":
    sys.exit(main())
