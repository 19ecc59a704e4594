"""Committed golden vectors (tests/golden/pair_160x120.npz, made by tests/golden/make_golden.py with the oracle):
  * CPU: the oracle must keep reproducing them bit for bit (pins the restatement against accidental change);
  * GPU: the HIP path through the C-ABI on the same inputs - exact for detection, toleranced for the tracked sums."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
W, H = 160, 120
KW = dict(keylines_ref=700, keylines_max=900, global_min_matches_threshold=50)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "pair_160x120.npz"))


def _params(mod, gold):
    fm, cx, cy = (float(x) for x in gold["cam"])
    return mod.default_params(H, W, fm=fm, cx=cx, cy=cy, **KW)


def _eq(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.names:
        return all(_eq(a[f], b[f]) for f in a.dtype.names)
    if a.dtype.kind == "f":
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def test_oracle_reproduces_golden(orc_mod, gold):
    orc = orc_mod.Oracle(_params(orc_mod, gold))
    frames = gold["frames"]
    maps = []
    for i in range(3):
        m = orc.detect_u8(frames[i], i * 50000)
        maps.append(m)
        assert _eq(m.keylines(), gold[f"det{i}_keylines"]), f"detect {i}"
        assert np.array_equal(m.mask(H, W), gold[f"det{i}_mask"])
        assert np.float32(m.threshold) == gold[f"det{i}_threshold"]
        assert np.float32(orc.threshold) == gold[f"det{i}_servo"]
        if i == 0:
            ss = orc.scale_space(frames[0].astype(np.float32) * np.float32(3.0))
            assert _eq(ss["dog"], gold["dog0"]) and _eq(ss["mag"], gold["mag0"])
            orc.build_distance_field(m)
            ids, dist = orc.distance_field()
            assert np.array_equal(ids, gold["df0_id"])
            assert np.array_equal(dist[ids >= 0], gold["df0_dist"][ids >= 0])
        if i >= 1:
            po = orc.track_pair(maps[i - 1], m)
            assert _eq(np.array(po.Vg, np.float32), gold[f"pair{i}_Vg"])
            assert _eq(np.array(po.Xv, np.float32), gold[f"pair{i}_Xv"])
            assert [po.klm_num, po.kf_matches, po.reg_num, po.lm_accept_mask, po.status] == list(gold[f"pair{i}_ints"])
            assert _eq(m.keylines(), gold[f"pair{i}_new_keylines"])


@pytest.mark.gpu
def test_hip_matches_golden(gold):
    import torch  # noqa: F401
    from rebvio_amd import backend as B
    ctx = B.Context(_params(B, gold))
    frames = gold["frames"]
    maps = []
    for i in range(3):
        m = ctx.detect_u8(frames[i], i * 50000)
        maps.append(m)
        assert _eq(m.keylines(), gold[f"det{i}_keylines"]), f"detect {i}"
        assert np.array_equal(m.mask(), gold[f"det{i}_mask"])
        assert np.float32(m.threshold) == gold[f"det{i}_threshold"]
        assert np.float32(ctx.detector_state()[0]) == gold[f"det{i}_servo"]
        if i == 0:
            ss = ctx.scale_space(frames[0].astype(np.float32) * np.float32(3.0))
            assert _eq(ss["dog"], gold["dog0"]) and _eq(ss["mag"], gold["mag0"])
            ctx.build_distance_field(m)
            ids, dist = ctx.distance_field()
            assert np.array_equal(ids, gold["df0_id"])
            assert np.array_equal(dist[ids >= 0], gold["df0_dist"][ids >= 0])
        if i >= 1:
            # sync the GPU maps to the golden state of the previous step so that each pair is compared on identical inputs
            if i == 2:
                maps[1].upload(gold["pair1_new_keylines"])
            po = ctx.track_pair(maps[i - 1], m)
            vg = gold[f"pair{i}_Vg"]
            assert np.abs(np.array(po.Vg) - vg).max() <= 1e-6 + 2e-3 * np.abs(vg).max()  # fp32 sum order (900 terms)
            ints = list(gold[f"pair{i}_ints"])
            assert po.status == ints[4] and po.lm_accept_mask == ints[3]
            assert abs(po.klm_num - ints[0]) <= 0.01 * ints[0] and abs(po.reg_num - ints[2]) <= 0.01 * ints[2]
            kg = m.keylines()
            ko = gold[f"pair{i}_new_keylines"]
            assert (kg["match_id"] == ko["match_id"]).mean() >= 0.99
