#!/usr/bin/env python3
"""Generates tests/golden/pair_160x120.npz with the CPU oracle (the reference itself cannot run here: TooN / OpenCV
absent). Inputs (three u8 frames) and the oracle's outputs for detect x3 and two tracking steps.
Re-run only when the oracle's defined semantics change:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from rebvio_amd import synth  # noqa: E402

W, H = 160, 120
KW = dict(keylines_ref=700, keylines_max=900, global_min_matches_threshold=50)


def main():
    O.build()
    frames, cam = synth.render_stream(W, H, 3, density=1.0)
    orc = O.Oracle(O.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, **KW))
    out = dict(frames=frames, cam=np.array([cam.fm, cam.cx, cam.cy], np.float32))
    maps = []
    for i in range(3):
        m = orc.detect_u8(frames[i], i * 50000)
        maps.append(m)
        out[f"det{i}_keylines"] = m.keylines()
        out[f"det{i}_mask"] = m.mask(H, W)
        out[f"det{i}_threshold"] = np.float32(m.threshold)
        out[f"det{i}_servo"] = np.float32(orc.threshold)
        if i == 0:
            ss = orc.scale_space(frames[0].astype(np.float32) * np.float32(3.0))
            out["dog0"], out["mag0"] = ss["dog"], ss["mag"]
            orc.build_distance_field(m)
            out["df0_id"], out["df0_dist"] = orc.distance_field()
        if i >= 1:
            po = orc.track_pair(maps[i - 1], m)
            for f in ("Vg", "P_Vg", "Xv", "Xgv", "V", "R"):
                out[f"pair{i}_{f}"] = np.array(getattr(po, f), np.float32)
            out[f"pair{i}_W_Xv"] = np.array(po.W_Xv, np.float32)
            out[f"pair{i}_ints"] = np.array([po.klm_num, po.kf_matches, po.reg_num, po.lm_accept_mask, po.status], np.int32)
            out[f"pair{i}_F_srm"] = np.array([po.F, po.sigma_rho_min], np.float32)
            out[f"pair{i}_new_keylines"] = m.keylines()
            out[f"pair{i}_old_keylines"] = maps[i - 1].keylines()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pair_160x120.npz"), **out)
    print("keylines:", [len(out[f"det{i}_keylines"]) for i in range(3)], "klm:", out["pair1_ints"], out["pair2_ints"])


if __name__ == "__main__":
    main()
