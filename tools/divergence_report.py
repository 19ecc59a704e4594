#!/usr/bin/env python3
"""SURVEY.md H4's divergence report for a stream: GPU pipeline vs CPU oracle, state carried independently on both sides.
Per pair: do the Levenberg-Marquardt accept / reject decisions agree (same path through minimizeVel, core.cpp:166-185), and how
far apart are the translations; first pair at which a decision differs. Also against the oracle run with double-accumulated
sums (what the translation would be without the summation rounding of either side), and against the oracle run with its
keyline sums in the DEVICE's order (oracle_py.Oracle.set_sum_order("device")): what is left then is the 6x6 solve of
extRotVel / the glue algebra, where the restatement and the library each follow the reference's SVD in their own way.
  divergence_report.py [n_frames] [stream_id ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def run(n, sid, W=640, H=480, kref=15000, kmax=16000, density=1.0, with_dev=False, **extra):
    import torch  # noqa: F401
    from oracle import oracle_py as O
    from rebvio_amd import backend as B, synth
    frames, cam = synth.render_stream(W, H, 8, stream_id=sid, density=density)
    order = synth.pingpong_indices(8, n)
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=kref, keylines_max=kmax, **extra)

    def oracle(sum_order):
        orc = O.Oracle(O.default_params(H, W, **kw))
        orc.set_sum_order(sum_order)
        prev, rec = None, []
        for k, i in enumerate(order):
            m = orc.detect_u8(frames[i], k * 50000)
            if prev is not None:
                po = orc.track_pair(prev, m)
                rec.append((np.array(po.Vg), po.lm_accept_mask, po.klm_num, po.status))
            prev = m
        return rec

    ref, wide = oracle("reference"), oracle("wide")
    dev_order = oracle("device") if with_dev else None
    ctx = B.Context(B.default_params(H, W, **kw))
    dev = ctx.upload_frames(frames)
    got = []
    for k, i in enumerate(order):
        out, _ = ctx.push_frame_u8_device(dev + int(i) * W * H, k * 50000)
        if out.status >= 0:
            got.append((np.array(out.Vg), out.lm_accept_mask, out.klm_num, out.status))
    for out, _ in ctx.flush():  # the records still on their way when the input ended
        got.append((np.array(out.Vg), out.lm_accept_mask, out.klm_num, out.status))
    ctx.close()
    if with_dev:
        return ref, wide, got, dev_order
    return ref, wide, got


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def analyse(ref, wide, got):
    m = min(len(ref), len(got))
    rows = []
    first = None
    for k in range(m):
        same = ref[k][1] == got[k][1]
        if not same and first is None:
            first = k
        rows.append((k, ref[k][1], got[k][1], rel(got[k][0], ref[k][0]), rel(got[k][0], wide[k][0]), rel(ref[k][0], wide[k][0]), ref[k][2], got[k][2]))
    return first, rows


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    sids = [int(s) for s in sys.argv[2:]] or [0, 1, 2]
    for sid in sids:
        ref, wide, got, devo = run(n, sid, with_dev=True)
        first, rows = analyse(ref, wide, got)
        print(f"stream {sid}: {len(rows)} pairs compared; first pair whose LM accept mask differs from the oracle's: {first}")
        print("  pair  mask(oracle/gpu)  |Vg gpu-oracle|/|Vg|  |Vg gpu-wide|/|Vg|  |Vg oracle-wide|/|Vg|  matches(oracle/gpu)")
        for k, mo, mg, d1, d2, d3, ko, kg in rows:
            print(f"  {k:4d}  {mo:05b}/{mg:05b}{' *' if mo != mg else '  '}      {d1:9.2e}           {d2:9.2e}          {d3:9.2e}        {ko}/{kg}")
        before = [r for r in rows if first is None or r[0] < first]
        after = [r for r in rows if first is not None and r[0] >= first]
        print(f"  while the LM paths agree ({len(before)} pairs): max gpu-oracle {max((r[3] for r in before), default=0):.2e}, "
              f"max gpu-wide {max((r[4] for r in before), default=0):.2e}, max oracle-wide {max((r[5] for r in before), default=0):.2e}")
        if after:
            print(f"  after the first differing decision ({len(after)} pairs): max gpu-oracle {max(r[3] for r in after):.2e}, "
                  f"max gpu-wide {max(r[4] for r in after):.2e}, max oracle-wide {max(r[5] for r in after):.2e}")
        md = min(len(devo), len(got))
        same = [devo[k][1] == got[k][1] and devo[k][2] == got[k][2] for k in range(md)]
        dd = [rel(got[k][0], devo[k][0]) for k in range(md)]
        bit = [bool(np.array_equal(np.float32(got[k][0]).view(np.uint32), np.float32(devo[k][0]).view(np.uint32))) for k in range(md)]
        nb = next((k for k in range(md) if not bit[k]), None)
        print(f"  against the oracle with its sums in the device's order ({md} pairs): LM masks and match counts equal on {sum(same)}, "
              f"translation bit-identical on {sum(bit)} (first pair that is not: {nb}), max |Vg gpu-oracle|/|Vg| {max(dd):.2e}")
