"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Bars:
  * bit-exact: scale space (DoG, squared gradient), keyline set + order + every keyline field, dense mask,
    edge chaining ids, auto threshold, distance-field ids/distances, rotateKeylines, estimateQuantile,
    match_id_forward / residuals of tryVel, forwardMatch, directedMatch, regularize1Iter, depth EKF
    (all per-keyline fp32 arithmetic is evaluated in the reference's order, fp contraction off);
  * tolerance (stated at each assert): the fp32 sums over ~15k keylines (tryVel: score/JtJ/JtF, extRotVel:
    JtJ/JtF) — the oracle adds sequentially in index order, the GPU uses a fixed butterfly/tree order;
  * and bit-exact again, sums and everything downstream of them (the LM run, the glue, whole streams with the state
    carried independently), once the oracle adds those sums in the kernels' order - Oracle.set_sum_order("device"), a
    diagnostic of the restatement: test_lm_sums_in_device_order_are_bit_exact,
    test_whole_pipeline_is_bit_identical_with_the_sums_in_one_order. The tolerances above are that order and nothing else.
"""
import os

import numpy as np
import pytest

from conftest import params_for

pytestmark = pytest.mark.gpu

REL_SUM = 2e-4   # relative tolerance of a 15k-term fp32 sum, sequential vs tree order
KW_C2 = dict(keylines_ref=15000, keylines_max=16000)


@pytest.fixture(scope="module")
def B():
    import torch  # noqa: F401  (the bench process has torch's HIP runtime loaded first; do the same here)
    from rebvio_amd import backend
    backend.lib()
    return backend


def _bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.dtype.kind == "f":
        return np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


def assert_keylines_equal(ko, kg, fields=None, what=""):
    assert len(ko) == len(kg), f"{what}: size {len(ko)} vs {len(kg)}"
    for f in (fields or ko.dtype.names):
        if not _bits_equal(ko[f], kg[f]):
            bad = np.nonzero((ko[f] != kg[f]).reshape(len(ko), -1).any(1))[0]
            raise AssertionError(f"{what}: field {f} differs at {len(bad)} keylines, first {bad[:5]}: "
                                 f"{ko[f][bad[:3]]} vs {kg[f][bad[:3]]}")


def run_stream(ctx, dev, order, npx, k0=0, ts_step=50000):
    """Push order[] through the streaming driver and flush: every pair's (record, keyline count), in pair order - len(order) - 1
    of them. Records come back several pushes late (the pair step runs on the device from end to end, pairs are queued in
    groups and the host reads them up to fifteen pairs behind); the flush tracks the pairs not started yet and delivers the
    rest."""
    recs = []
    for k, i in enumerate(order):
        out, n = ctx.push_frame_u8_device(dev + int(i) * npx, (k0 + k) * ts_step)
        if out.status >= 0:
            recs.append((out, n))
    recs.extend(ctx.flush())
    return recs


def pair_tuple(out, n=None):
    t = (tuple(out.Vg), tuple(out.P_Vg), out.F, tuple(out.Xv), tuple(out.W_Xv), tuple(out.Xgv), tuple(out.V), tuple(out.R),
         tuple(out.P_V), out.sigma_rho_min, out.ext_ok, out.klm_num, out.kf_matches, out.reg_num, out.lm_accept_mask, out.status)
    return t if n is None else t + (n,)


class Pair:
    """Oracle and GPU contexts driven over the same frames; keeps the last two maps of each."""

    def __init__(self, O, B, frames, cam, **kw):
        self.O, self.B = O, B
        self.frames, self.cam = frames, cam
        self.orc = O.Oracle(params_for(O, cam, **kw))
        self.ctx = B.Context(params_for(B, cam, **kw))
        self.om = []
        self.gm = []

    def detect(self, i):
        om = self.orc.detect_u8(self.frames[i], i * 50000)
        gm = self.ctx.detect_u8(self.frames[i], i * 50000)
        self.om.append(om)
        self.gm.append(gm)
        if len(self.om) > 2:
            self.om.pop(0)
            self.gm.pop(0).release()
        return om, gm

    def sync_gpu_from_oracle(self):
        for om, gm in zip(self.om, self.gm):
            gm.upload(om.keylines())


def warm(O, B, frames, cam, n_pairs, **kw):
    """Run n_pairs oracle tracking steps so that depths/matches are realistic; the GPU only detects."""
    P = Pair(O, B, frames, cam, **kw)
    P.detect(0)
    for i in range(1, n_pairs + 1):
        P.detect(i)
        P.orc.track_pair(P.om[0], P.om[1])
    P.detect(n_pairs + 1)  # pair under test: om[0] (tracked once as "new"), om[1] fresh
    P.sync_gpu_from_oracle()
    return P


# --------------------------------------------------------------------------------------------------------
def test_scale_space_bit_exact(orc_mod, B, c2_stream):
    frames, cam = c2_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam))
    ctx = B.Context(params_for(B, cam))
    for i in (0, 3):
        img = frames[i].astype(np.float32) * np.float32(3.0)
        so, sg = orc.scale_space(img), ctx.scale_space(img)
        for k in ("scale0", "scale1", "dog", "mag"):
            assert _bits_equal(so[k], sg[k]), f"{k} differs in {(so[k] != sg[k]).sum()} pixels"


@pytest.mark.parametrize("sigma,n", [(3.56359, 1), (3.56359, 2), (2.2, 4), (3.0, 5), (4.49, 3)])
def test_fast_gaussian_with_any_number_of_box_passes(orc_mod, B, c2_stream, sigma, n):
    """FastGaussian(camera, sigma, n).smooth for n other than the 3 the reference itself constructs (scale_space.cpp:14-41 takes
    any n, :186 uses 3): createIntegralImage, n - 1 x (average + createIntegralImage), average - bit-exact against the
    restatement, whose Kovesi widths the device entry is handed."""
    frames, cam = c2_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam))
    ctx = B.Context(params_for(B, cam))
    img = frames[2].astype(np.float32) * np.float32(3.0)
    so, widths = orc.smooth(img, sigma, n)
    assert len(widths) == n and all(3 <= w <= 11 and w % 2 == 1 for w in widths), widths
    sg = ctx.smooth(img, widths)
    assert _bits_equal(so, sg), f"n = {n}, widths {widths}: {(so != sg).sum()} pixels differ"
    with pytest.raises(B.HipError, match="3..11"):
        ctx.smooth(img, [13] * n)


@pytest.mark.parametrize("width", [2048, 2044, 1300])
def test_first_row_pass_is_exact_on_wide_bright_frames(orc_mod, B, width):
    """The first row pass of a MONO8 frame is a parallel prefix (its partial sums are integers below 2^24, any order gives
    the sequential bits: detect.hip, wave_prefix_exact). Widest frames the wave form takes (2048 columns, eight chunks per
    lane), a width that fills the last lanes partially, and near-white pixels so that the row sums are as large as they get."""
    h = 48
    rng = np.random.default_rng(width)
    frames = [rng.integers(200, 256, (h, width)).astype(np.uint8) for _ in range(2)]
    frames[1][:, ::3] = 255
    kw = dict(keylines_ref=3000, keylines_max=6000)
    orc = orc_mod.Oracle(orc_mod.default_params(h, width, **kw))
    ctx = B.Context(B.default_params(h, width, **kw))
    for i, f in enumerate(frames):
        om, gm = orc.detect_u8(f, i * 50000), ctx.detect_u8(f, i * 50000)
        assert_keylines_equal(om.keylines(), gm.keylines(), what=f"{width} wide, frame {i}")
        assert np.array_equal(om.mask(h, width), gm.mask())
        assert om.threshold == gm.threshold


@pytest.mark.parametrize("shape", [(64, 48), (100, 36), (752, 480), (642, 480), (65, 49), (131, 67)])
def test_scale_space_ragged_sizes(orc_mod, B, shape):
    """Widths that are not multiples of the 64-lane tiles / 16-row strips, EuRoC's 752x480, and widths that are not a
    multiple of 4 (642, 65, 131: the integral images get a padded row pitch internally, the reference has no such limit,
    scale_space.cpp:48-67)."""
    w, h = shape
    rng = np.random.default_rng(w * 1000 + h)
    img = (rng.integers(0, 256, (h, w)).astype(np.float32)) * np.float32(3.0)
    pO = orc_mod.default_params(h, w)
    pB = B.default_params(h, w)
    so, sg = orc_mod.Oracle(pO).scale_space(img), B.Context(pB).scale_space(img)
    for k in ("scale0", "scale1", "dog", "mag"):
        assert _bits_equal(so[k], sg[k]), k


def test_detect_sequence_bit_exact(orc_mod, B, c2_stream):
    """Keyline index set, order, every field, masks, servo and auto thresholds over consecutive frames
    (includes truncation at keylines_max in the first frames)."""
    frames, cam = c2_stream
    P = Pair(orc_mod, B, frames, cam, **KW_C2)
    for i in range(len(frames)):
        om, gm = P.detect(i)
        assert om.size() == gm.size()
        assert_keylines_equal(om.keylines(), gm.keylines(), what=f"frame {i}")
        assert np.array_equal(om.mask(cam.height, cam.width), gm.mask())
        thr, auto, cnt = P.ctx.detector_state()
        assert np.float32(thr) == np.float32(P.orc.threshold)
        assert np.float32(auto) == np.float32(P.orc.auto_threshold)
        assert cnt == om.size()
        assert np.float32(gm.threshold) == np.float32(om.threshold)


def test_detect_constant_and_empty(orc_mod, B):
    """A constant image has no keylines: empty map, threshold carried over."""
    w, h = 128, 96
    img = np.full((h, w), 300.0, np.float32)
    orc = orc_mod.Oracle(orc_mod.default_params(h, w))
    ctx = B.Context(B.default_params(h, w))
    om, gm = orc.detect(img), ctx.detect(img)
    assert om.size() == 0 and gm.size() == 0
    assert (gm.mask() == -1).all()
    assert np.float32(gm.threshold) == np.float32(om.threshold)


def test_detect_reuses_the_previous_frames_map(orc_mod, B, small_stream):
    """Pool exhausted: every detect gets the map the previous frame has just released (prev_st aliases the map's own
    scalars inside k_keyline_emit). The auto threshold carried into an EMPTY frame must still be the previous frame's
    tuneThreshold value, not the reset min/max (+inf)."""
    frames, cam = small_stream
    kw = dict(keylines_ref=1500, keylines_max=2000, map_pool=4)
    orc = orc_mod.Oracle(params_for(orc_mod, cam, **{k: v for k, v in kw.items() if k != "map_pool"}))
    ctx = B.Context(params_for(B, cam, **kw))
    held = [ctx.detect_u8(frames[i]) for i in range(3)]  # three of the four pooled maps stay in use
    for i in range(3):
        orc.detect_u8(frames[i])
    blank = np.full((cam.height, cam.width), 100, np.uint8)
    for i, f in enumerate([frames[3], frames[4], blank, frames[5], blank, blank]):
        om, gm = orc.detect_u8(f), ctx.detect_u8(f)
        assert om.size() == gm.size(), i
        assert_keylines_equal(om.keylines(), gm.keylines())
        assert np.float32(gm.threshold) == np.float32(om.threshold), (i, gm.threshold, om.threshold)
        assert np.isfinite(gm.threshold)
        gm.release()
    assert len([m for m in held if m.h]) == 3


def test_detect_small_keylines_max(orc_mod, B, small_stream):
    """Early truncation: only the first keylines_max raster-ordered candidates survive, mask cleared after."""
    frames, cam = small_stream
    kw = dict(keylines_ref=150, keylines_max=200)
    orc = orc_mod.Oracle(params_for(orc_mod, cam, **kw))
    ctx = B.Context(params_for(B, cam, **kw))
    om, gm = orc.detect_u8(frames[0]), ctx.detect_u8(frames[0])
    assert om.size() == 200 == gm.size()
    assert_keylines_equal(om.keylines(), gm.keylines())
    assert np.array_equal(om.mask(cam.height, cam.width), gm.mask())


def test_distance_field_exact(orc_mod, B, c2_stream):
    frames, cam = c2_stream
    P = Pair(orc_mod, B, frames, cam, **KW_C2)
    for i in range(3):
        om, gm = P.detect(i)
    P.orc.build_distance_field(om)
    P.ctx.build_distance_field(gm)
    ido, dso = P.orc.distance_field()
    idg, dsg = P.ctx.distance_field()
    assert np.array_equal(ido, idg)
    sel = ido >= 0
    assert np.array_equal(dso[sel], dsg[sel])
    assert sel.sum() > 100000


def test_rotate_and_quantile_bit_exact(orc_mod, B, c2_stream):
    frames, cam = c2_stream
    P = warm(orc_mod, B, frames, cam, 2, **KW_C2)
    a = 0.004
    R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    R = (R @ np.array([[1, 0, 0], [0, np.cos(0.002), -np.sin(0.002)], [0, np.sin(0.002), np.cos(0.002)]], np.float32)).astype(np.float32)
    om, gm = P.om[0], P.gm[0]
    P.orc.rotate(om, R)
    P.ctx.rotate(gm, R)
    assert_keylines_equal(om.keylines(), gm.keylines(), what="rotate")
    for pct, bins in ((0.9, 100), (0.5, 37), (0.99, 128)):
        assert np.float32(P.orc.quantile(om, pct, bins)) == np.float32(P.ctx.quantile(gm, pct, bins))


def _sums_close(a, b, scale=None, rel=REL_SUM):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    s = scale if scale is not None else max(np.abs(a).max(), 1e-30)
    assert np.abs(a - b).max() <= rel * s, f"max diff {np.abs(a - b).max()} vs scale {s}"


def test_try_vel_parity(orc_mod, B, c2_stream):
    """tryVel: match_id_forward and residuals exact (incl. the carry-forward rule), sums within REL_SUM."""
    frames, cam = c2_stream
    P = warm(orc_mod, B, frames, cam, 3, **KW_C2)
    om_old, om_new = P.om
    gm_old, gm_new = P.gm
    P.orc.build_distance_field(om_new)
    P.ctx.build_distance_field(gm_new)
    n = om_old.size()
    srm = P.orc.quantile(om_old)
    res_o = np.zeros(n, np.float32)
    res_g = np.zeros(n, np.float32)
    nmatched = []
    for vel in ([0, 0, 0], [-0.011, -0.005, -0.003], [0.02, 0.01, -0.9]):
        so, Jo, Fo = P.orc.try_vel(om_old, vel, srm, res_o)
        sg, Jg, Fg = P.ctx.try_vel(gm_old, vel, srm, res_g)
        ko, kg = om_old.keylines(), gm_old.keylines()
        assert np.array_equal(ko["match_id_forward"], kg["match_id_forward"])
        assert _bits_equal(res_o, res_g), f"residuals differ at {(res_o != res_g).sum()}"
        assert abs(so - sg) <= REL_SUM * abs(so)
        # off-diagonal / signed sums cancel: scale by the diagonal magnitude
        _sums_close(Jo, Jg, scale=np.abs(np.diag(Jo)).max())
        _sums_close(Fo, Fg, scale=np.sqrt(so * np.abs(np.diag(Jo)).max()))
        nmatched.append(int((ko["match_id_forward"] >= 0).sum()))
    assert nmatched[0] > 1000 and nmatched[1] > 1000  # the third velocity sends most keylines out of view


def test_minimize_vel_parity(orc_mod, B, c2_stream):
    frames, cam = c2_stream
    P = warm(orc_mod, B, frames, cam, 3, **KW_C2)
    om_old, om_new = P.om
    gm_old, gm_new = P.gm
    P.orc.build_distance_field(om_new)
    P.ctx.build_distance_field(gm_new)
    ro = P.orc.minimize_vel(om_old)
    rg = P.ctx.minimize_vel(gm_old)
    assert np.float32(ro["sigma_rho_min"]) == np.float32(rg["sigma_rho_min"])
    assert ro["accept_mask"] == rg["accept_mask"]
    # single LM run on identical inputs: fp32 sum order is the only difference
    assert np.abs(ro["vel"] - rg["vel"]).max() <= 1e-6 + 1e-3 * np.abs(ro["vel"]).max()
    assert abs(ro["F"] - rg["F"]) <= 1e-3 * abs(ro["F"])
    assert np.abs(ro["Rvel"] - rg["Rvel"]).max() <= 2e-3 * np.abs(ro["Rvel"]).max()
    ko, kg = om_old.keylines(), gm_old.keylines()
    same = (ko["match_id_forward"] == kg["match_id_forward"]).mean()
    assert same >= 0.999, same


def test_forward_match_and_ext_rot_vel(orc_mod, B, c2_stream):
    frames, cam = c2_stream
    P = warm(orc_mod, B, frames, cam, 3, **KW_C2)
    om_old, om_new = P.om
    gm_old, gm_new = P.gm
    P.orc.build_distance_field(om_new)
    P.ctx.build_distance_field(gm_new)
    ro = P.orc.minimize_vel(om_old)
    # make the GPU old map carry the oracle's match_id_forward, then compare forwardMatch exactly
    gm_old.upload(om_old.keylines())
    P.orc.forward_match(om_old, om_new)
    P.ctx.forward_match(gm_old, gm_new)
    assert_keylines_equal(om_new.keylines(), gm_new.keylines(), what="forwardMatch")
    assert (om_new.keylines()["match_id"] >= 0).sum() > 1000
    eo = P.orc.ext_rot_vel(ro["vel"])
    eg = P.ctx.ext_rot_vel(ro["vel"])
    d = np.sqrt(np.abs(np.diag(eo["Wx"])))
    _sums_close(eo["Wx"] / np.outer(d, d), eg["Wx"] / np.outer(d, d), scale=1.0)
    _sums_close(eo["JtF"] / d, eg["JtF"] / d, scale=np.abs(eo["JtF"] / d).max() + 1.0)
    assert eo["ok"] == eg["ok"] == 1
    # solution of a 6x6 system with condition ~1e4: looser
    assert np.abs(eo["X"] - eg["X"]).max() <= 5e-3 * np.abs(eo["X"]).max() + 1e-6


@pytest.mark.parametrize("config", ["c2", "c3"], ids=["c2-16k(spec kernel)", "c3-64k(chain kernel)"])
def test_lm_sums_in_device_order_are_bit_exact(orc_mod, B, c2_stream, c3_stream, config):
    """What the tolerances of the three tests above ARE: tryVel, minimizeVel and extRotVel reduce ten / twenty-seven fp32 sums
    over all keylines, the reference adds them in index order, the kernels in a tree (a wave's DPP scan, the four waves of a
    group, the groups dealt to sixteen lanes) - and nothing else differs. With the oracle's sums taken in the kernels' order
    (set_sum_order("device"), a diagnostic of the restatement: oracle/rebvio_oracle.cpp struct Acc) every one of those outputs
    comes back BIT FOR BIT on synced inputs: the score and the 3x3 / 3 sums of tryVel at three velocities, the whole
    Levenberg-Marquardt run (velocity, score, covariance, accept mask, every forward match), extRotVel's 6x6 and 6 sums."""
    frames, cam = c2_stream if config == "c2" else c3_stream
    kw = KW_C2 if config == "c2" else KW_C3
    P = warm(orc_mod, B, frames, cam, 3, **kw)
    P.orc.set_sum_order("device")
    om_old, om_new = P.om
    gm_old, gm_new = P.gm
    P.orc.build_distance_field(om_new)
    P.ctx.build_distance_field(gm_new)
    n = om_old.size()
    assert n > (50000 if config == "c3" else 10000)
    ro = P.orc.minimize_vel(om_old)
    rg = P.ctx.minimize_vel(gm_old)
    assert ro["accept_mask"] == rg["accept_mask"]
    for k in ("vel", "F", "Rvel", "sigma_rho_min"):
        assert _bits_equal(np.float32(ro[k]), np.float32(rg[k])), (k, ro[k], rg[k])
    assert np.array_equal(om_old.keylines()["match_id_forward"], gm_old.keylines()["match_id_forward"])
    P.orc.forward_match(om_old, om_new)
    P.ctx.forward_match(gm_old, gm_new)
    assert_keylines_equal(om_new.keylines(), gm_new.keylines(), what="forwardMatch")
    eo = P.orc.ext_rot_vel(ro["vel"])
    eg = P.ctx.ext_rot_vel(ro["vel"])
    assert _bits_equal(np.float32(eo["Wx"]), np.float32(eg["Wx"])), np.abs(eo["Wx"] - eg["Wx"]).max()
    assert _bits_equal(np.float32(eo["JtF"]), np.float32(eg["JtF"])), (eo["JtF"], eg["JtF"])
    # single evaluations, residuals carried from one to the next on both sides
    srm = P.orc.quantile(om_old)
    res_o = np.zeros(n, np.float32)
    res_g = np.zeros(n, np.float32)
    for vel in ([0, 0, 0], [-0.011, -0.005, -0.003], [0.02, 0.01, -0.9]):
        so, Jo, Fo = P.orc.try_vel(om_old, vel, srm, res_o)
        sg, Jg, Fg = P.ctx.try_vel(gm_old, vel, srm, res_g)
        assert _bits_equal(res_o, res_g)
        assert _bits_equal(np.float32([so]), np.float32([sg])), (vel, so, sg)
        assert _bits_equal(np.float32(Jo), np.float32(Jg)), (vel, Jo, Jg)
        assert _bits_equal(np.float32(Fo), np.float32(Fg)), (vel, Fo, Fg)


@pytest.fixture(scope="module")
def c3_stream():
    """5 frames of BASELINE config 3: 1280x960, ~58k keylines of a 64 000-keyline budget."""
    from rebvio_amd import synth
    return synth.render_stream(1280, 960, 5, density=2.0)


KW_C3 = dict(keylines_ref=60000, keylines_max=64000, threshold=0.006)


@pytest.mark.parametrize("config,head", [("c2", None), ("c2", "compact4"), ("c2", "compact1"), ("c3", None), ("c3", "compact4"), ("c3", "compact8")],
                         ids=["c2-default(compact8)", "c2-compact4", "c2-compact1", "c3-64k-default(compact1)", "c3-64k-compact4", "c3-64k-compact8"])
def test_directed_match_regularize_ekf_bit_exact(orc_mod, B, c2_stream, c3_stream, monkeypatch, config, head):
    """directedMatch / searchMatch (edge_map.cpp:101-218), regularize1Iter, depth EKF on maps synced from the oracle: every
    keyline field bit-exact, counters equal - for every form of the one-launch directedMatch kernel k_directed_match_c (REBVIO_HIP_DM_HEAD,
    read when the context is created): 8 / 4 / 1 lanes per keyline (<512, 8> the default up to 32 768 keylines, <64, 1> beyond and
    for batches of four lanes or more, <256, 4> the alternative), each also where it is not the default, at BASELINE config 2 and
    on a 64 000-keyline map of config 3 (1280x960)."""
    frames, cam = c2_stream if config == "c2" else c3_stream
    kw = KW_C2 if config == "c2" else KW_C3
    if head:
        monkeypatch.setenv("REBVIO_HIP_DM_HEAD", head)
    else:
        monkeypatch.delenv("REBVIO_HIP_DM_HEAD", raising=False)
    P = warm(orc_mod, B, frames, cam, 3, **kw)
    om_old, om_new = P.om
    gm_old, gm_new = P.gm
    if config == "c3":
        assert om_new.size() > 50000
    P.orc.build_distance_field(om_new)
    P.ctx.build_distance_field(gm_new)
    ro = P.orc.minimize_vel(om_old)
    P.orc.forward_match(om_old, om_new)
    gm_old.upload(om_old.keylines())
    gm_new.upload(om_new.keylines())
    V = ro["vel"]
    Rvel = ro["Rvel"]
    a = 0.0007
    Rb = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    no, kfo = P.orc.directed_match(om_new, om_old, V, Rvel, Rb)
    ng, kfg = P.ctx.directed_match(gm_new, gm_old, V, Rvel, Rb)
    assert (no, kfo) == (ng, kfg)
    assert no > (5000 if config == "c2" else 20000)
    assert_keylines_equal(om_new.keylines(), gm_new.keylines(), what="directedMatch")
    ro_n, rg_n = P.orc.regularize(om_new), P.ctx.regularize(gm_new)
    assert ro_n == rg_n and ro_n > 1000
    assert_keylines_equal(om_new.keylines(), gm_new.keylines(), what="regularize")
    P.orc.update_inverse_depth(V)
    P.ctx.update_inverse_depth(V)
    assert_keylines_equal(om_new.keylines(), gm_new.keylines(), what="depth EKF")


@pytest.mark.parametrize("head", [None, "compact4", "compact1"], ids=["compact8", "compact4", "compact1"])
def test_directed_match_beyond_one_probe_window(orc_mod, B, c2_stream, monkeypatch, head):
    """searchMatch with max_radius = 100 (the reference's callers pass 40, rebvio.cpp:245): t_steps reaches 103, so the long searches
    of k_directed_match_c run through three windows of 40 probe steps - the +-1.0f chains continue from window to window in the
    owners' registers, a keyline found in one window is not probed in the next - in every instantiation. Bit-exact against the
    oracle, like the one-window case of test_directed_match_regularize_ekf_bit_exact."""
    frames, cam = c2_stream
    if head:
        monkeypatch.setenv("REBVIO_HIP_DM_HEAD", head)
    else:
        monkeypatch.delenv("REBVIO_HIP_DM_HEAD", raising=False)
    P = warm(orc_mod, B, frames, cam, 3, **KW_C2)
    om_old, om_new = P.om
    gm_old, gm_new = P.gm
    P.orc.build_distance_field(om_new)
    P.ctx.build_distance_field(gm_new)
    ro = P.orc.minimize_vel(om_old)
    P.orc.forward_match(om_old, om_new)
    gm_old.upload(om_old.keylines())
    gm_new.upload(om_new.keylines())
    a = 0.0007
    Rb = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    no, kfo = P.orc.directed_match(om_new, om_old, ro["vel"], ro["Rvel"], Rb, max_radius=100.0)
    ng, kfg = P.ctx.directed_match(gm_new, gm_old, ro["vel"], ro["Rvel"], Rb, max_radius=100.0)
    assert (no, kfo) == (ng, kfg) and no > 5000
    assert_keylines_equal(om_new.keylines(), gm_new.keylines(), what="directedMatch, radius 100")
    with pytest.raises(B.HipError):
        P.ctx.directed_match(gm_new, gm_old, ro["vel"], ro["Rvel"], Rb, max_radius=300.0)


def test_c3_stream_tracks_oracle_stream(orc_mod, B):
    """BASELINE config 3 as a STREAM (1280x960, ~58k keylines; k_lm_chain<512> on 125 workgroups, the directedMatch
    kernel with one lane per keyline) through rebvio_hip_push_frame_u8_device against the oracle driven over the
    same eight frames, state carried independently on both sides, with the bars of the divergence report
    (test_stream_divergence_report): while the Levenberg-Marquardt decisions agree the GPU translation is within 1e-2 of the
    oracle run with double-accumulated sums and no farther from the fp32 oracle than that oracle is from its own
    double-accumulated run (+1e-2); match counts within 1 %."""
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import divergence_report as D
    ref, wide, got = D.run(8, 0, W=1280, H=960, kref=60000, kmax=64000, density=2.0, threshold=0.006)
    assert len(ref) == 7 == len(got)
    first, rows = D.analyse(ref, wide, got)
    assert first is None, f"LM decisions differ from the oracle's at pair {first}"
    for k, mo, mg, d_ref, d_wide, d_own, ko, kg in rows:
        assert ko > 20000
        assert d_wide <= 1e-2, (k, d_wide)
        assert d_ref <= d_own + 1e-2, (k, d_ref, d_own)
        assert abs(ko - kg) <= 0.01 * ko + 2, (k, ko, kg)


def test_directed_match_degenerate_velocity(orc_mod, B, small_stream):
    """|t| <= 1e-6 branch of searchMatch: search along the keyline's own gradient."""
    frames, cam = small_stream
    P = warm(orc_mod, B, frames, cam, 2)
    om_old, om_new = P.om
    gm_old, gm_new = P.gm
    V = np.zeros(3, np.float32)
    I = np.eye(3, dtype=np.float32)
    no, kfo = P.orc.directed_match(om_new, om_old, V, I * 1e-6, I)
    ng, kfg = P.ctx.directed_match(gm_new, gm_old, V, I * 1e-6, I)
    assert (no, kfo) == (ng, kfg) and no > 50
    assert_keylines_equal(om_new.keylines(), gm_new.keylines(), what="directedMatch degenerate")


def test_track_pair_sequence(orc_mod, B, c2_stream):
    """End-to-end frame pairs with state carried on each side independently (no re-sync):
    per-pair tolerances of SURVEY.md H4, and agreement of the integer outputs."""
    frames, cam = c2_stream
    P = Pair(orc_mod, B, frames, cam, **KW_C2)
    P.detect(0)
    for i in range(1, len(frames)):
        om, gm = P.detect(i)
        po = P.orc.track_pair(P.om[0], P.om[1])
        pg = P.ctx.track_pair(P.gm[0], P.gm[1])
        assert po.status == pg.status == 0
        assert po.lm_accept_mask == pg.lm_accept_mask
        vo, vg = np.array(po.Vg), np.array(pg.Vg)
        assert np.abs(vo - vg).max() <= 1e-6 + 5e-2 * np.abs(vo).max(), (i, vo, vg)
        assert abs(po.klm_num - pg.klm_num) <= 0.01 * po.klm_num
        ko, kg = om.keylines(), gm.keylines()
        assert (ko["match_id"] == kg["match_id"]).mean() >= 0.97
        both = (ko["match_id"] == kg["match_id"]) & (ko["match_id"] >= 0)
        rel = np.abs(ko["rho"][both] - kg["rho"][both]) / np.abs(ko["rho"][both])
        assert np.median(rel) < 1e-3


def _record_words(po):
    """every field of a pair record as raw 32-bit words (floats by their bits)"""
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


@pytest.mark.parametrize("stream_id,config", [(0, "c2"), (1, "c2"), (2, "c2"), (0, "c3")],
                         ids=["c2-stream0", "c2-stream1", "c2-stream2", "c3-1280x960-64k"])
def test_whole_pipeline_is_bit_identical_with_the_sums_in_one_order(orc_mod, B, stream_id, config):
    """The stateful comparison without a tolerance. Thirty frames of a 640x480 stream (BASELINE config 2; ten of config 3's
    1280x960 with ~58k keylines), detection, tracking,
    glue, directedMatch, regularisation and depth filter with the state carried INDEPENDENTLY on both sides - the oracle adding
    the keyline sums of tryVel / extRotVel in the kernels' order (set_sum_order("device"); the terms, and every other operation
    of the path, are the restatement's own). Per pair: every word of the pair record - translation and its covariance, score, the
    visual 6-vector and its information matrix, the corrected one, rotation, sigma quantile, counters, accept mask - identical,
    through the per-pair API and through the streaming driver (device glue, speculative LM kernel); after the last pair every
    field of every keyline of the newest map identical. So over a whole stream the ONLY thing that separates the kernels
    from the restatement of the reference is how 37 fp32 sums are associated - which is what the tolerances of
    test_track_pair_sequence / test_stream_divergence_report measure, and nothing else."""
    from rebvio_amd import synth
    if config == "c2":
        W, H, npairs, kw, min_klm = 640, 480, 29, KW_C2, 8000
        frames, cam = synth.render_stream(W, H, 8, stream_id=stream_id)
    else:  # BASELINE config 3: ~58k keylines of a 64 000 budget, the non-speculative chain kernel, <64, 1> directedMatch
        W, H, npairs, kw, min_klm = 1280, 960, 9, KW_C3, 20000
        frames, cam = synth.render_stream(W, H, 8, stream_id=stream_id, density=2.0)
    order = synth.pingpong_indices(8, npairs + 1)
    orc = orc_mod.Oracle(params_for(orc_mod, cam, **kw))
    orc.set_sum_order("device")
    gpu = B.Context(params_for(B, cam, **kw))
    mo, mg, rec_o = [], [], []
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
        wo, wg = _record_words(po), _record_words(pg)
        assert np.array_equal(wo, wg), (k, np.flatnonzero(wo != wg)[:8], np.array(po.Vg), np.array(pg.Vg))
        rec_o.append(wo)
    assert rec_o[-1][0] != 0 and po.klm_num > min_klm
    # the map that carries the state into the next pair (the older one is dropped after its pair, rebvio.cpp:136-139)
    assert_keylines_equal(mo[1].keylines(), mg[1].keylines(), what=f"{config} stream {stream_id}: newest map after {npairs} pairs")
    gpu.close()
    # the streaming driver on the same frames: device glue, persistent speculative LM kernel, pairs queued in groups
    ctx = B.Context(params_for(B, cam, **kw))
    dev = ctx.upload_frames(frames)
    got = []
    for k, i in enumerate(order):
        out, _ = ctx.push_frame_u8_device(dev + int(i) * W * H, k * 50000)
        if out.status >= 0:
            got.append(_record_words(out))
    for out, _ in ctx.flush():
        got.append(_record_words(out))
    ctx.close()
    assert len(got) == len(rec_o) == npairs
    for k, (wo, wg) in enumerate(zip(rec_o, got)):
        assert np.array_equal(wo, wg), (k, np.flatnonzero(wo != wg)[:8])


def test_pose_deviation_is_the_references_own_rounding_noise(orc_mod, B, c2_stream):
    """Why the pose tolerance above is 5 %: the pose of a pair is a 6-step LM on sums over ~15k keylines, and early in a
    stream (all depths at their initial value) it is sensitive to the LAST BITS of those sums. Measured here: the oracle with
    the same fp32 terms accumulated in double (diagnostic mode, not the reference) against (a) the sequential-fp32 oracle and
    (b) the GPU's fixed-tree fp32 sums, states carried independently over the stream. The GPU must be no further from the
    wide-sum result than the reference's own order of summation is - and within 5e-3 of it."""
    frames, cam = c2_stream
    kw = dict(KW_C2)
    seq = orc_mod.Oracle(params_for(orc_mod, cam, **kw))
    wide = orc_mod.Oracle(params_for(orc_mod, cam, **kw))
    wide.set_wide_sums(True)
    gpu = B.Context(params_for(B, cam, **kw))
    ms, mw, mg = [], [], []
    d_seq = d_gpu = 0.0
    agree_seq = agree_gpu = 1.0
    for i in range(len(frames)):
        ms.append(seq.detect_u8(frames[i], i * 50000))
        mw.append(wide.detect_u8(frames[i], i * 50000))
        mg.append(gpu.detect_u8(frames[i], i * 50000))
        if len(ms) > 2:
            ms.pop(0)
            mw.pop(0)
            mg.pop(0).release()
        if i == 0:
            continue
        vs = np.array(seq.track_pair(ms[0], ms[1]).Vg)
        vw = np.array(wide.track_pair(mw[0], mw[1]).Vg)
        vg = np.array(gpu.track_pair(mg[0], mg[1]).Vg)
        d_seq = max(d_seq, float(np.abs(vs - vw).max() / np.abs(vw).max()))
        d_gpu = max(d_gpu, float(np.abs(vg - vw).max() / np.abs(vw).max()))
        kw_, ks_, kg_ = mw[1].keylines(), ms[1].keylines(), mg[1].keylines()
        agree_seq = min(agree_seq, float((ks_["match_id"] == kw_["match_id"]).mean()))
        agree_gpu = min(agree_gpu, float((kg_["match_id"] == kw_["match_id"]).mean()))
    assert d_gpu <= 5e-3, (d_gpu, d_seq)
    assert d_gpu <= max(d_seq, 1e-3), (d_gpu, d_seq)
    assert agree_gpu >= min(agree_seq, 0.995), (agree_gpu, agree_seq)


def test_full_size_properties(B):
    """1280x960 (~60k keylines): properties that need no oracle pass over the image at this size are cheap
    to state - here the oracle still finishes in well under a second per frame, so compare directly, plus
    raster-order / mask consistency invariants."""
    from oracle import oracle_py as O
    from rebvio_amd import synth
    frames, cam = synth.render_stream(1280, 960, 2, density=2.0)
    kw = dict(keylines_ref=60000, keylines_max=64000, threshold=0.006)   # ~58k keylines on the first frame
    orc = O.Oracle(params_for(O, cam, **kw))
    ctx = B.Context(params_for(B, cam, **kw))
    oms, gms = [], []
    for i in range(2):
        om, gm = orc.detect_u8(frames[i], i * 50000), ctx.detect_u8(frames[i], i * 50000)
        oms.append(om)
        gms.append(gm)
        kg = gm.keylines()
        assert len(kg) > 50000
        assert_keylines_equal(om.keylines(), kg, what=f"C3 frame {i}")
        mask = gm.mask()
        ys, xs = np.nonzero(mask >= 0)
        assert len(ys) == len(kg)
        assert np.array_equal(mask[ys, xs], np.arange(len(kg)))  # raster rank == keyline index
        px = np.floor(kg["pos"] + 0.5)
        assert (np.abs(kg["pos"][:, 0] - xs) <= 0.5).all() and (np.abs(kg["pos"][:, 1] - ys) <= 0.5).all()
        assert px.shape[0] == len(kg)
    # the pair step at full size: 63 workgroups of the persistent LM kernel, ~230 record groups
    po, pg = orc.track_pair(oms[0], oms[1]), ctx.track_pair(gms[0], gms[1])
    assert po.status == pg.status == 0 and po.lm_accept_mask == pg.lm_accept_mask
    vo, vg = np.array(po.Vg), np.array(pg.Vg)
    assert np.abs(vo - vg).max() <= 1e-6 + 5e-2 * np.abs(vo).max(), (vo, vg)
    assert abs(po.klm_num - pg.klm_num) <= 0.01 * po.klm_num
    ko, kg = oms[1].keylines(), gms[1].keylines()
    assert (ko["match_id"] == kg["match_id"]).mean() >= 0.97


# ---- front end: u8 -> x3 -> undistort on the device ----------------------------------------------------
EUROC_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0]  # camera.hpp:31-35


def test_front_end_undistort_bit_exact(orc_mod, B, c2_stream):
    frames, cam = c2_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam, **KW_C2))
    ctx = B.Context(params_for(B, cam, **KW_C2))
    for D in (EUROC_D, [0.6, -0.1, 1e-3, -2e-3, 0.05]):   # barrel (inside the frame) and pincushion (reads the zero border)
        ctx.set_undistort(cam.fm, cam.fm, cam.cx, cam.cy, D)
        for i in (0, 3):
            want = orc.front_end_u8(frames[i], cam.fm, cam.fm, cam.cx, cam.cy, D)
            got = ctx.front_end_u8(frames[i])
            assert _bits_equal(want, got), f"D={D} frame {i}: {np.abs(want - got).max()}"


def test_detect_through_device_front_end(orc_mod, B, c2_stream):
    """u8 host frame -> device (x3 + undistort + detect) == oracle front end + oracle detect, keyline for keyline; and
    with the lens model switched off the u8 entry equals the fp32 entry."""
    frames, cam = c2_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam, **KW_C2))
    ctx = B.Context(params_for(B, cam, **KW_C2))
    ctx.set_undistort(cam.fm, cam.fm, cam.cx, cam.cy, EUROC_D)
    for i in range(4):
        om = orc.detect(orc.front_end_u8(frames[i], cam.fm, cam.fm, cam.cx, cam.cy, EUROC_D), i * 50000)
        gm = ctx.detect_u8_host(frames[i], i * 50000)
        assert_keylines_equal(om.keylines(), gm.keylines(), what=f"undistorted frame {i}")
        assert om.threshold == gm.threshold
    ctx2 = B.Context(params_for(B, cam, **KW_C2))
    ctx3 = B.Context(params_for(B, cam, **KW_C2))
    ctx3.set_undistort(cam.fm, cam.fm, cam.cx, cam.cy, [0, 0, 0, 0, 0])
    for i in range(3):
        a = ctx2.detect_u8(frames[i], i * 50000)
        b = ctx3.detect_u8_host(frames[i], i * 50000)
        assert_keylines_equal(a.keylines(), b.keylines(), what=f"u8 entry frame {i}")


def test_edge_image_rendering(B, c2_stream):
    """Device-side edge image == what ros_rebvio.cpp:32-50 draws: grey -> RGB, keyline pixels red."""
    frames, cam = c2_stream
    ctx = B.Context(params_for(B, cam, **KW_C2))
    m = ctx.detect_u8(frames[0], 0)
    kl = m.keylines()
    want = np.repeat(frames[0][:, :, None], 3, axis=2)
    rows = np.floor(kl["pos"][:, 1] + 0.5).astype(int)   # std::round for the non-negative pixel coordinates
    cols = np.floor(kl["pos"][:, 0] + 0.5).astype(int)
    want[rows, cols] = (255, 0, 0)
    got = m.render_edge_image(frames[0])
    assert np.array_equal(got, want)
    black = m.render_edge_image(None)
    assert black.sum() == 255 * len(np.unique(rows * cam.width + cols))


def test_persistent_lm_kernel_equals_per_call_kernels(B, c2_stream, monkeypatch):
    """The persistent minimizeVel / forwardMatch / extRotVel kernel (grid exchange through tagged words, keylines in
    registers) and the seven per-evaluation kernels it replaces share the per-keyline code and the record order: every
    output of the pair step must agree bit for bit - for the speculative and the sequential form and for every workgroup
    size the kernels are built for (REBVIO_HIP_LM, REBVIO_HIP_LM_THREADS: read when the context is created). Also exercises
    workgroups without live keylines (15k keylines in a 16k-keyline launch) and the streaming pipeline on every path."""
    frames, cam = c2_stream

    def run(mode, threads):
        monkeypatch.setenv("REBVIO_HIP_LM", mode)
        monkeypatch.setenv("REBVIO_HIP_LM_THREADS", str(threads))
        ctx = B.Context(params_for(B, cam, **KW_C2))
        maps = [ctx.detect_u8(frames[i], i * 50000) for i in range(len(frames))]
        outs = [pair_tuple(ctx.track_pair(maps[i - 1], maps[i])) for i in range(1, len(frames))]
        last = maps[-1].keylines()
        ctx.close()
        # streaming driver on a fresh context of the same mode
        ctx2 = B.Context(params_for(B, cam, **KW_C2))
        dev = ctx2.upload_frames(frames)
        order = list(range(len(frames))) + list(range(len(frames) - 2, -1, -1)) + list(range(1, len(frames)))
        stream = [pair_tuple(o, n) for o, n in run_stream(ctx2, dev, order, cam.width * cam.height)]
        ctx2.close()
        return outs, last, stream

    a_out, a_kl, a_stream = run("persistent", 512)
    assert len(a_stream) == 3 * len(frames) - 2 - 1   # every pair of the stream
    for mode, threads in (("percall", 512), ("seq", 512), ("seq", 256), ("seq", 1024), ("spec", 256), ("spec3", 512), ("spec3", 256)):
        b_out, b_kl, b_stream = run(mode, threads)
        assert a_out == b_out, (mode, threads)
        assert_keylines_equal(a_kl, b_kl, what=f"last map, persistent vs {mode}/{threads}")
        assert a_stream == b_stream, (mode, threads)


def test_device_glue_equals_host_glue(B, c2_stream, monkeypatch):
    """The streaming driver keeps a pair on the device from end to end: the glue between its halves (sum of the extRotVel
    records, 6x6 solve, gyroBiasCorrection, SO3, Cholesky covariance, rebvio.cpp:177-233) runs in front of the directedMatch
    kernel, with the gyro-bias state in device memory (glue.hpp, track.hip: glue_prologue). The per-pair API
    (rebvio_hip_track_pair) runs the SAME statements on the host. Same frames -> the same records, bit for bit, pair by pair:
    velocity, covariance, extRotVel solution, corrected pose increment, rotation, counters, masks - and the same keylines in
    the last map. (sin / cos inside SO3::exp are hostmath.hpp's own double evaluation on both sides, so not even libm's last
    bit differs.) Also with one lane per keyline in the directedMatch kernel in place of eight."""
    frames, cam = c2_stream
    n = len(frames)
    order = list(range(n)) + list(range(n - 2, -1, -1)) + list(range(1, n))
    npx = cam.width * cam.height
    for head in ("", "compact1"):
        if head:
            monkeypatch.setenv("REBVIO_HIP_DM_HEAD", head)
        else:
            monkeypatch.delenv("REBVIO_HIP_DM_HEAD", raising=False)
        ctx = B.Context(params_for(B, cam, **KW_C2))
        dev = ctx.upload_frames(frames)
        maps = [ctx.detect_u8_device(dev + int(i) * npx, k * 50000) for k, i in enumerate(order)]
        want = [pair_tuple(ctx.track_pair(maps[k - 1], maps[k]), maps[k].size()) for k in range(1, len(order))]
        bg_host = ctx.gyro_state()
        ctx.close()
        ctx2 = B.Context(params_for(B, cam, **KW_C2))
        dev2 = ctx2.upload_frames(frames)
        got = [pair_tuple(o, nk) for o, nk in run_stream(ctx2, dev2, order, npx)]
        bg_dev = ctx2.gyro_state()          # mirrored from the device's filter state, exact after the flush
        ctx2.close()
        assert len(got) == len(order) - 1 == len(want)
        for k, (g, w) in enumerate(zip(got, want)):
            assert g == w, (head, k, [i for i in range(len(g)) if g[i] != w[i]])
        assert np.array_equal(bg_host[0].view(np.uint32), bg_dev[0].view(np.uint32)) and np.abs(bg_host[0]).max() > 0
        assert np.array_equal(bg_host[1].view(np.uint32), bg_dev[1].view(np.uint32))
        assert all(w[-2] == 0 and w[-6] > 5000 for w in want)   # status 0, thousands of matches


def test_glue_probe_random_inputs(B):
    """rebvio_hip_test_glue: the device form of the pair glue (glue_dev.hpp: three waves, one column / cofactor per lane) against
    the host form (glue.hpp: pair_glue_core, one thread) on random inputs - well-conditioned systems, rank-deficient ones (the
    6x6 solve falls back to the Jacobi pseudo-inverse), zero pivots in the LDL^T inverses, rotations in each branch of
    SO3::exp (|w|^2 < 1e-8, < 1e-6, larger, beyond pi), NaN inputs. Every output word must agree bit for bit."""
    ctx = B.Context(B.default_params(96, 128, keylines_ref=3000, keylines_max=4000))
    rng = np.random.default_rng(11)

    def words(a):
        a = np.ascontiguousarray(a, np.float32)
        w = a.view(np.uint32).copy()
        w[np.isnan(a)] = 0x7FC00000   # a NaN is a NaN (its sign bit is not part of the comparison)
        return w

    n_fallback = n_nan = 0
    for trial in range(160):
        kind = trial % 8
        n_new = int(rng.integers(1, 4000))
        nb = (n_new + 255) // 256
        # extRotVel records: sums of outer products of rows with realistic scales (pixels / focal length / depth)
        scale = np.array([400.0, 400.0, 200.0, 500.0, 500.0, 300.0], np.float64) * 10.0 ** rng.uniform(-2, 1)
        rows = rng.standard_normal((nb, 300, 6)) * scale
        if kind == 1:
            rows[:, :, 5] = rows[:, :, 4] * 2.0            # rank-deficient: Jacobi pseudo-inverse
        if kind == 2:
            rows[:, :, 3:] = 0.0                            # zero rotational block
        Y = rng.standard_normal((nb, 300)) * 0.3
        xrv = np.zeros((nb, 32), np.float32)
        iu = np.triu_indices(6)
        for b in range(nb):
            W = rows[b].T @ rows[b]
            xrv[b, :21] = W[iu]
            xrv[b, 21:27] = rows[b].T @ Y[b]
            xrv[b, 27] = 300
        if kind == 3:
            xrv[0, 2] = np.nan
        J = rng.standard_normal((200, 3)) * [300.0, 300.0, 150.0]
        JtJ = J.T @ J
        JtJ6 = np.array([JtJ[0, 0], JtJ[1, 1], JtJ[2, 2], JtJ[0, 1], JtJ[0, 2], JtJ[1, 2]], np.float32)
        vel = (rng.standard_normal(3) * 0.01).astype(np.float32)
        bg_scale = [1e-5, 3e-4, 1e-2, 1.0][trial % 4]      # SO3::exp branches of the prior rotation
        Bg = (rng.standard_normal(3) * bg_scale).astype(np.float32)
        if kind == 4:
            Bg[:] = 0
        W_Bg = (np.eye(3) * 10.0 ** rng.uniform(-2, 6) + rng.standard_normal((3, 3)) * 1e-3).astype(np.float32)
        W_Bg = ((W_Bg + W_Bg.T) / 2).astype(np.float32)
        from scipy.spatial.transform import Rotation
        Rp = Rotation.from_rotvec(rng.standard_normal(3) * [1e-3, 1e-2, 0.3][trial % 3]).as_matrix().astype(np.float32)
        if kind == 5:
            Y *= 40.0                                       # a large rotation increment (beyond pi/2 after scaling)
            for b in range(nb):
                xrv[b, 21:27] = rows[b].T @ Y[b]
        frame_dt = [0.05, 0.0333333, 0.1][trial % 3]
        (od, sd, gd), (oh, sh, gh) = ctx.test_glue(vel, JtJ6, float(rng.uniform(1, 1e4)), float(rng.uniform(0.1, 5)), trial & 31, xrv, n_new,
                                                   frame_dt, Bg, W_Bg, Rp)
        td, th = pair_tuple(od), pair_tuple(oh)
        for i, (x, y) in enumerate(zip(td, th)):
            assert np.array_equal(words(np.atleast_1d(x)), words(np.atleast_1d(y))), (trial, kind, "record field", i, x, y)
        assert np.array_equal(words(sd), words(sh)), (trial, kind, "state", sd, sh)
        assert np.array_equal(words(gd), words(gh)), (trial, kind, "second half", gd, gh)
        n_nan += int(oh.status == 1)
        if kind == 1:
            n_fallback += 1
            assert np.isfinite(np.array(oh.Xv)).all() or oh.ext_ok == 0
    assert n_fallback >= 15 and n_nan >= 15
    ctx.close()


def test_speculative_lm_kernel_rolls_back_when_a_later_step_is_accepted(B, monkeypatch):
    """The default persistent kernel evaluates minimizeVel's evaluations 2.. in one pass under the hypothesis that all of
    them are rejected (what consecutive frames do: accept mask 00001), checks the hypothesis with the real scores and
    otherwise goes on sequentially from the first accepted one. Pairs of frames several steps apart (and other search
    ranges / iteration counts) give masks with later accepts: on all of them the kernel must equal the sequential
    persistent kernel (REBVIO_HIP_LM=seq) bit for bit, and the sweep must actually contain such masks."""
    from rebvio_amd import synth
    streams = [synth.render_stream(640, 480, 12, stream_id=sid) for sid in (0, 1)]

    def run(mode):
        monkeypatch.setenv("REBVIO_HIP_LM", mode)   # read when the context is created
        res = {}
        for sid, variant, kw in [(sid, v, kw) for sid in (0, 1) for v, kw in (("default", {}), ("sr10", dict(search_range=10.0)),
                                                                            ("it7", dict(iterations=7)), ("it3", dict(iterations=3)))]:
            frames, cam = streams[sid]
            ctx = B.Context(params_for(B, cam, **dict(KW_C2, **kw)))
            for skip in (1, 2, 4, 7):
                for i in range(skip, len(frames)):
                    # fresh maps per pair: forwardMatch keys and depth state of a map belong to ONE pair step
                    maps = [ctx.detect_u8(frames[i - skip], 0), ctx.detect_u8(frames[i], skip * 50000)]
                    o = ctx.track_pair(maps[0], maps[1])
                    v = np.concatenate([np.array(o.Vg), np.array(o.P_Vg), [o.F, o.sigma_rho_min], np.array(o.Xv), np.array(o.W_Xv),
                                        np.array(o.V), np.array(o.P_V),
                                        [o.klm_num, o.kf_matches, o.reg_num, o.lm_accept_mask, o.status]]).astype(np.float32)
                    kl = maps[1].keylines()
                    res[(sid, variant, skip, i)] = (int(o.lm_accept_mask), v, kl["match_id"].copy(), kl["rho"].copy())
                    for m in maps:
                        m.release()
            ctx.close()
        return res

    a, b = run("spec"), run("seq")
    a3 = run("spec3")  # the same kernel with its first speculative evaluation at index 3 (hypothesis: nothing accepted after the second)
    masks = {m for m, *_ in b.values()}
    assert 1 in masks and len(masks - {1}) >= 3, masks           # the common case and several kinds of mis-speculation
    assert any(m & ~3 for m in masks), masks                      # an accept at the third step or later
    for k in b:
        for which, x in (("spec", a), ("spec3", a3)):
            if k[1] == "it3" and which == "spec3":
                continue  # (three iterations leave one evaluation behind index 3: that context runs the sequential kernel)
            assert x[k][0] == b[k][0], (which, k)
            assert _bits_equal(x[k][1], b[k][1]), (which, k)
            assert np.array_equal(x[k][2], b[k][2]) and _bits_equal(x[k][3], b[k][3]), (which, k)


def _vision_only_fusion(mid):
    """rebvio.cpp:195-203,225-233 without the inertial filter, in numpy (only has to be the SAME in every call order below)."""
    from scipy.spatial.transform import Rotation
    Xgv = np.array(mid.Xgv, np.float64)
    R = np.array(mid.R, np.float64).reshape(3, 3)
    R0 = Rotation.from_rotvec(Xgv[3:]).as_matrix()
    R = (R0 @ R.T).T
    V = R0 @ np.array(mid.Vg, np.float64) + Xgv[:3]
    P = np.linalg.inv(np.array(mid.W_Xgv, np.float64).reshape(6, 6))[:3, :3]
    return V.astype(np.float32), P.astype(np.float32), R.astype(np.float32), R0.astype(np.float32)


def test_pair_halves_report_the_same_counters_in_every_call_order(B, c2_stream):
    """rebvio_hip_track_pair_begin / _finish_async / _result: a pair's match counters come back with the next pair's first half
    when that pair continues from the same map (no copy), through a copy queued by _begin when it does not, and through a
    copy made by _result when no pair follows; rebvio_hip_track_pair_hint_next only moves a wait. Same inputs -> the same
    counters and the same keylines whichever way they travel."""
    frames, cam = c2_stream
    n = 8

    def run(order, hint):
        ctx = B.Context(params_for(B, cam, **KW_C2))
        maps = [ctx.detect_u8(frames[i], i * 50000) for i in range(n)]
        res = []
        if order == "overlapped":      # begin(k), finish_async(k), begin(k+1), result(k), ...: what rebvio::Rebvio does
            pending = False
            for k in range(n - 1):
                mid = ctx.track_pair_begin(maps[k], maps[k + 1])
                if pending:
                    res.append(ctx.track_pair_result())
                if hint and k + 2 < n:
                    ctx.track_pair_hint_next(maps[k + 2])
                ctx.track_pair_finish_async(maps[k], maps[k + 1], *_vision_only_fusion(mid))
                pending = True
            res.append(ctx.track_pair_result())
        elif order == "serial":        # begin(k), finish_async(k), result(k): _result copies
            for k in range(n - 1):
                mid = ctx.track_pair_begin(maps[k], maps[k + 1])
                ctx.track_pair_finish_async(maps[k], maps[k + 1], *_vision_only_fusion(mid))
                res.append(ctx.track_pair_result())
        else:                          # every other pair restarts from fresh maps: _begin queues the copy
            for k in range(0, n - 1, 2):
                mid = ctx.track_pair_begin(maps[k], maps[k + 1])
                if res or k:
                    res.append(ctx.track_pair_result())
                ctx.track_pair_finish_async(maps[k], maps[k + 1], *_vision_only_fusion(mid))
            res.append(ctx.track_pair_result())
        kl = maps[n - 1].keylines() if order != "disjoint" else maps[n - 1 - (n % 2)].keylines()
        for m in maps:
            m.release()
        ctx.close()
        return res, kl

    a, kla = run("overlapped", False)
    b, klb = run("serial", False)
    c, klc = run("overlapped", True)
    assert len(a) == n - 1 and a == b == c
    assert all(r[3] == 0 and r[0] > 5000 for r in a), a
    assert_keylines_equal(kla, klb, what="overlapped vs serial")
    assert_keylines_equal(kla, klc, what="with vs without the hint")
    d, _ = run("disjoint", False)
    e = []
    ctx = B.Context(params_for(B, cam, **KW_C2))   # the same disjoint pairs one at a time
    maps = [ctx.detect_u8(frames[i], i * 50000) for i in range(n)]
    for k in range(0, n - 1, 2):
        mid = ctx.track_pair_begin(maps[k], maps[k + 1])
        ctx.track_pair_finish_async(maps[k], maps[k + 1], *_vision_only_fusion(mid))
        e.append(ctx.track_pair_result())
    for m in maps:
        m.release()
    ctx.close()
    assert d == e and len(d) == n // 2
    # the new map released before its pair's result is fetched: the release copies the counters out first
    f = []
    ctx = B.Context(params_for(B, cam, **KW_C2))
    maps = [ctx.detect_u8(frames[i], i * 50000) for i in range(n)]
    for k in range(0, n - 1, 2):
        mid = ctx.track_pair_begin(maps[k], maps[k + 1])
        ctx.track_pair_finish_async(maps[k], maps[k + 1], *_vision_only_fusion(mid))
        maps[k + 1].release()
        maps[k].release()
        f.append(ctx.track_pair_result())
    ctx.close()
    assert f == e


def test_streaming_records_do_not_depend_on_the_lm_kernel_choice(B, c2_stream, monkeypatch):
    """The streaming driver picks the speculative or the sequential persistent LM kernel per pair from the stream's recent
    accept masks (api.hip, note_accept_mask). Replaying the frames in jumps makes the masks vary, so the default switches
    back and forth; its records must be those of either kernel pinned."""
    frames, cam = c2_stream
    rng = np.random.default_rng(3)
    order = np.cumsum(rng.integers(1, 6, size=160)) % len(frames)
    npx = cam.width * cam.height

    def run(mode):
        if mode is None:
            monkeypatch.delenv("REBVIO_HIP_LM", raising=False)
        else:
            monkeypatch.setenv("REBVIO_HIP_LM", mode)
        ctx = B.Context(params_for(B, cam, **KW_C2))
        dev = ctx.upload_frames(frames)
        rec = [(o.status, o.lm_accept_mask, tuple(o.Vg), tuple(o.Xv), o.klm_num, o.reg_num, n) for o, n in run_stream(ctx, dev, order, npx)]
        ctx.close()
        return rec

    a, b, c, d = run(None), run("seq"), run("spec"), run("spec3")
    assert len(a) > 100 and a == b == c == d
    masks = {r[1] for r in a}
    assert 1 in masks and len(masks) >= 3, masks   # consecutive-like pairs and several kinds of later accepts


def test_host_frame_stream_equals_device_frame_stream(B, c2_stream):
    """rebvio_hip_push_frame_u8 (MONO8 frames in host memory, staged through the pinned ring and copied ahead of the scans, the
    entry the reference's imageCallback corresponds to, rebvio.cpp:38-48) against frames resident in HBM: the same records.
    Frames are handed over from ONE reused host buffer (overwritten right after each call) and from a padded array (row
    pitch > cols)."""
    from rebvio_amd import synth
    frames, cam = c2_stream
    order = synth.pingpong_indices(len(frames), 45)
    npx = cam.width * cam.height
    ctx = B.Context(params_for(B, cam, **KW_C2))
    dev = ctx.upload_frames(frames)
    want = [pair_tuple(o, n) for o, n in run_stream(ctx, dev, order, npx)]
    ctx.close()
    ctx = B.Context(params_for(B, cam, **KW_C2))
    buf = np.zeros((cam.height, cam.width + 24), np.uint8)
    got = []
    for k, i in enumerate(order):
        view = buf[:, :cam.width] if k % 2 else np.ascontiguousarray(frames[i])
        if k % 2:
            view[:] = frames[i]
        out, n = ctx.push_frame_u8(view, k * 50000)
        buf[:] = 0                                    # the caller's buffer is its own again as soon as the push returns
        if out.status >= 0:
            got.append(pair_tuple(out, n))
    got.extend(pair_tuple(o, n) for o, n in ctx.flush())
    ctx.close()
    assert len(got) == len(order) - 1 and got == want


def test_stream_continues_cleanly_after_a_flush(B, c2_stream):
    """rebvio_hip_flush() in the middle of a stream: the stream that follows must start like a fresh one. The last second
    half before the flush has already binned the sigma histogram for a pair that never comes; those counts used to put the
    quantile cut of the first pair after the flush below every fresh keyline's sigma (status 1, no matches)."""
    from rebvio_amd import synth
    frames, cam = c2_stream
    ctx = B.Context(params_for(B, cam, **KW_C2))
    dev = ctx.upload_frames(frames)
    npx = cam.width * cam.height
    order = synth.pingpong_indices(len(frames), 90)

    def segment(k0):
        return [(o.status, o.lm_accept_mask, o.klm_num, float(o.sigma_rho_min)) for o, _ in run_stream(ctx, dev, order[k0:k0 + 30], npx, k0=k0)]

    first, second, third = segment(0), segment(30), segment(60)
    for rec in (first, second, third):
        assert len(rec) == 30 - 1
        assert all(r[0] == 0 for r in rec), rec[:4]
        assert rec[0][3] == first[0][3]            # every segment's first pair starts from the fresh maps' sigma (quantile of 1000s)
        assert all(r[2] > 5000 for r in rec), rec[:4]


def test_streaming_pipeline_tracks_oracle_stream(orc_mod, B, c2_stream):
    """The throughput pipeline (detect worker + five streams + persistent pair kernel + deferred counters) against the
    oracle's own stream driver on the same 30-frame ping-pong sequence, state carried independently on both sides:
    per-pair translation within 5 % of its magnitude (+1e-6), visual rotation increment within 1e-4 rad, match counts
    within 1 %; results arrive in pair order, a few pushes late, and flush() delivers nothing out of order."""
    from rebvio_amd import synth
    frames, cam = c2_stream
    order = synth.pingpong_indices(len(frames), 30)
    ref = orc_mod.Oracle(params_for(orc_mod, cam, **KW_C2)).run_stream(frames, order, threads=1)
    ctx = B.Context(params_for(B, cam, **KW_C2))
    dev = ctx.upload_frames(frames)
    npx = cam.width * cam.height
    got = [(np.array(o.Vg), np.array(o.Xgv[3:6]), o.klm_num, o.status) for o, _ in run_stream(ctx, dev, order, npx)]
    assert len(got) == len(order) - 1
    # the oracle's record k describes pair (k-1, k); the pipeline reports the pairs in the same order starting at pair 1
    for j, (vg, dw, klm, status) in enumerate(got):
        k = j + 1
        assert status == 0
        vo, wo = ref["pose"][k, :3], ref["pose"][k, 3:]
        assert np.abs(vo - vg).max() <= 1e-6 + 5e-2 * np.abs(vo).max(), (k, vo, vg)
        assert np.abs(wo - dw).max() <= 1e-4, (k, wo, dw)
        assert abs(int(ref["match_counts"][k]) - klm) <= 0.01 * ref["match_counts"][k] + 2


def test_streaming_results_do_not_depend_on_pipeline_depth(B, c2_stream, monkeypatch):
    """The detect stage leads the tracker by REBVIO_HIP_LEAD frames, pairs are queued on the track stream in groups of
    REBVIO_HIP_GROUP (one wait and one event per group), the context's three streams have priorities or not
    (REBVIO_HIP_PRIO=flat, for several contexts per process), in-kernel phase stamps are taken or not (REBVIO_HIP_LM_STAMPS):
    all of these only move work around. Same frames -> the same records, bit for bit, in the same order."""
    from rebvio_amd import synth
    frames, cam = c2_stream
    order = synth.pingpong_indices(len(frames), 40)
    npx = cam.width * cam.height

    def run(lead, group, prio=None, stamps=False, extra=None):
        monkeypatch.setenv("REBVIO_HIP_LEAD", str(lead))
        monkeypatch.setenv("REBVIO_HIP_GROUP", str(group))
        for name in ("REBVIO_HIP_DETECT_WORKER", "REBVIO_HIP_GYRO_PRE", "REBVIO_HIP_FUSE_DOG"):
            monkeypatch.delenv(name, raising=False)
        for name, val in (extra or {}).items():
            monkeypatch.setenv(name, val)
        if prio:
            monkeypatch.setenv("REBVIO_HIP_PRIO", prio)
        else:
            monkeypatch.delenv("REBVIO_HIP_PRIO", raising=False)
        if stamps:
            monkeypatch.setenv("REBVIO_HIP_LM_STAMPS", "1")
        else:
            monkeypatch.delenv("REBVIO_HIP_LM_STAMPS", raising=False)
        ctx = B.Context(params_for(B, cam, **KW_C2))
        dev = ctx.upload_frames(frames)
        rec = [pair_tuple(o, n) for o, n in run_stream(ctx, dev, order, npx)]
        ctx.close()
        return rec

    base = run(3, 1)
    assert len(base) == len(order) - 1
    for lead, group, prio, stamps in ((5, 1, None, False), (5, 4, None, False), (8, 2, None, False), (12, 6, None, False), (5, 4, "flat", False),
                                      (5, 3, None, True)):
        assert run(lead, group, prio, stamps) == base, (lead, group, prio, stamps)
    # who launches the detect kernels (the caller itself, the default, or the context's worker thread) moves no result either,
    # nor does who forms the data-independent 3x3 matrices of gyroBiasCorrection (the host from its shadow of W_Bg, the
    # default, or the device glue itself), nor the fused candidate kernel against k_dog_mag + k_keyline_flag
    for extra in ({"REBVIO_HIP_DETECT_WORKER": "1"}, {"REBVIO_HIP_GYRO_PRE": "0"}, {"REBVIO_HIP_FUSE_DOG": "0"},
                  {"REBVIO_HIP_DETECT_WORKER": "1", "REBVIO_HIP_GYRO_PRE": "0", "REBVIO_HIP_FUSE_DOG": "0"}):
        assert run(5, 4, extra=extra) == base, extra


def test_euroc_frame_size_with_lens_model(orc_mod, B):
    """752x480 (the reference's built-in EuRoC camera, camera.hpp:25-45: a width that is not a multiple of the 32- and
    64-pixel tiles) with its rad-tan lens model: front end + detection bit-exact, distance field exact, one pair step
    within the usual tolerances."""
    from rebvio_amd import synth
    W, H = 752, 480
    frames, cam = synth.render_stream(W, H, 3, dist=EUROC_D)
    kw = dict(fm=457.975, cx=367.215, cy=248.375, keylines_ref=12000, keylines_max=16000)
    orc = orc_mod.Oracle(orc_mod.default_params(H, W, **kw))
    ctx = B.Context(B.default_params(H, W, **kw))
    ctx.set_undistort(kw["fm"], kw["fm"], kw["cx"], kw["cy"], EUROC_D)
    oms, gms = [], []
    for i in range(3):
        om = orc.detect(orc.front_end_u8(frames[i], kw["fm"], kw["fm"], kw["cx"], kw["cy"], EUROC_D), i * 50000)
        gm = ctx.detect_u8_host(frames[i], i * 50000)
        assert_keylines_equal(om.keylines(), gm.keylines(), what=f"752x480 frame {i}")
        oms.append(om)
        gms.append(gm)
    orc.build_distance_field(oms[1])
    ctx.build_distance_field(gms[1])
    ido, dso = orc.distance_field()
    idg, dsg = ctx.distance_field()
    assert np.array_equal(ido, idg)
    assert np.array_equal(dso[ido >= 0], dsg[idg >= 0])
    po, pg = orc.track_pair(oms[0], oms[1]), ctx.track_pair(gms[0], gms[1])
    assert po.status == pg.status == 0 and po.lm_accept_mask == pg.lm_accept_mask
    vo, vg = np.array(po.Vg), np.array(pg.Vg)
    assert np.abs(vo - vg).max() <= 1e-6 + 5e-2 * np.abs(vo).max(), (vo, vg)
    assert abs(po.klm_num - pg.klm_num) <= 0.01 * po.klm_num + 2


@pytest.mark.parametrize("size", [(324, 250), (100, 37), (1028, 33), (642, 480), (323, 251), (129, 99)])
def test_ragged_sizes_detect_and_distance_field(orc_mod, B, size):
    """Widths / heights that are multiples of none of the tile sizes (4-row strips, 16-column strips, 64x4 keyline tiles,
    32x32 distance-field tiles): detection bit-exact, distance field exact."""
    from rebvio_amd import synth
    W, H = size
    frames, cam = synth.render_stream(W, H, 2)
    kw = dict(fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=3000, keylines_max=4000)
    orc = orc_mod.Oracle(orc_mod.default_params(H, W, **kw))
    ctx = B.Context(B.default_params(H, W, **kw))
    for i in range(2):
        om, gm = orc.detect_u8(frames[i], i * 50000), ctx.detect_u8(frames[i], i * 50000)
        assert_keylines_equal(om.keylines(), gm.keylines(), what=f"{W}x{H} frame {i}")
    orc.build_distance_field(om)
    ctx.build_distance_field(gm)
    ido, dso = orc.distance_field()
    idg, dsg = ctx.distance_field()
    assert np.array_equal(ido, idg)
    assert np.array_equal(dso[ido >= 0], dsg[idg >= 0])


def test_pair_step_failure_paths(orc_mod, B, small_stream):
    """rebvio.cpp:236-252: a pair whose directedMatch finds fewer than global_min_matches_threshold keylines ends with
    status 2 and no regularisation / depth update (the device gates both kernels on the counter), on the oracle and on
    the device alike; the streaming driver reports the same status and keeps running."""
    from rebvio_amd import synth
    frames, cam = small_stream
    other, _ = synth.render_stream(cam.width, cam.height, 1, stream_id=7)   # an unrelated scene
    kw = dict(keylines_ref=1500, keylines_max=2500, global_min_matches_threshold=900)
    orc = orc_mod.Oracle(params_for(orc_mod, cam, **kw))
    ctx = B.Context(params_for(B, cam, **kw))
    o0, g0 = orc.detect_u8(frames[0], 0), ctx.detect_u8(frames[0], 0)
    o1, g1 = orc.detect_u8(other[0], 50000), ctx.detect_u8(other[0], 50000)
    assert_keylines_equal(o1.keylines(), g1.keylines())
    po, pg = orc.track_pair(o0, o1), ctx.track_pair(g0, g1)
    assert po.status == 2 and pg.status == 2
    assert po.reg_num == 0 and pg.reg_num == 0
    assert abs(po.klm_num - pg.klm_num) <= 0.05 * max(po.klm_num, 1) + 5
    ko, kg = o1.keylines(), g1.keylines()
    unmatched = (ko["match_id"] < 0) & (kg["match_id"] < 0)
    assert unmatched.sum() > 100
    # no depth update happened: unmatched keylines still carry the initial depth state on both sides
    assert (ko["rho"][unmatched] == 1.0).all() and (kg["rho"][unmatched] == 1.0).all()
    assert (ko["sigma_rho"][unmatched] == 20.0).all() and (kg["sigma_rho"][unmatched] == 20.0).all()
    # streaming driver: the bad pair is reported with status 2, the following pairs are tracked again
    ctx2 = B.Context(params_for(B, cam, **kw))
    seq = np.concatenate([frames[:3], other, frames[3:12]])  # enough frames behind the bad one to drain the pipeline
    dev = ctx2.upload_frames(seq)
    npx = cam.width * cam.height
    st = [o.status for o, _ in run_stream(ctx2, dev, range(len(seq)), npx)]
    assert len(st) == len(seq) - 1 and st[0] == 0 and 2 in st and st[-1] == 0, st   # the bad pair is reported, the stream goes on


def test_record_read_before_its_pair_wrote_it_is_reported(B, small_stream):
    """Every pair carries a sequence stamp that its kernels store as the last word of each host-visible record (result slot:
    LM state + map state records; glue record: pose record and filter state), and every reader compares it before it uses the
    record. rebvio_hip_test_forge_record_stamp hands the NEXT pair's kernels a wrong number - what a record looks like when it
    is read before the pair has written it, the failure seen in round 3 with kernel-bound stop events (192x144 stream, this
    fixture) - and the reading call must fail with -12 on every path: per-pair API, split API, streaming driver (harvest
    inside a push or the flush), batch. One deterministic run each; no timing involved."""
    import ctypes
    from rebvio_amd import backend
    frames, cam = small_stream
    kw = dict(keylines_ref=1500, keylines_max=2500, global_min_matches_threshold=100)
    npx = cam.width * cam.height

    def stale(fn):
        with pytest.raises(backend.HipError) as e:
            fn()
        assert "error -12" in str(e.value), str(e.value)

    # per-pair API: the slot is read right behind the stream synchronisation
    ctx = B.Context(params_for(B, cam, **kw))
    g0, g1, g2 = (ctx.detect_u8(frames[i], i * 50000) for i in range(3))
    assert ctx.track_pair(g0, g1).status == 0   # (stamps in order: nothing to report)
    backend._chk(backend.lib().rebvio_hip_test_forge_record_stamp(ctx.h))
    stale(lambda: ctx.track_pair(g1, g2))
    ctx.close()
    # split API (rebvio::Rebvio's path)
    ctx = B.Context(params_for(B, cam, **kw))
    g0, g1 = (ctx.detect_u8(frames[i], i * 50000) for i in range(2))
    backend._chk(backend.lib().rebvio_hip_test_forge_record_stamp(ctx.h))
    stale(lambda: ctx.track_pair_begin(g0, g1, None, 0.05))
    ctx.close()
    # streaming driver: the forged pair is reported by whichever call harvests it
    ctx = B.Context(params_for(B, cam, **kw))
    dev = ctx.upload_frames(frames)

    def stream():
        for k in range(len(frames)):
            if k == 8:
                backend._chk(backend.lib().rebvio_hip_test_forge_record_stamp(ctx.h))
            ctx.push_frame_u8_device(dev + k * npx, k * 50000)
        ctx.flush()
    stale(stream)
    ctx.close()
    # a clean stream of the same frames: every record arrives
    ctx = B.Context(params_for(B, cam, **kw))
    dev = ctx.upload_frames(frames)
    assert len(run_stream(ctx, dev, range(len(frames)), npx)) == len(frames) - 1
    ctx.close()
    # batch: the last lane of one step
    bat = B.Batch(params_for(B, cam, **kw), 2)
    devs = [bat.lanes[l].upload_frames(frames) for l in range(2)]

    def batch():
        for k in range(len(frames)):
            if k == 8:
                backend._chk(backend.lib().rebvio_hip_batch_test_forge_record_stamp(bat.h))
            bat.push_u8_device([d + k * npx for d in devs], k * 50000)
        bat.flush()
    stale(batch)
    bat.close()


def test_pair_step_nan_path(orc_mod, B, small_stream):
    """rebvio.cpp:236-241: a NaN velocity (here forced through NaN inverse depths in the old map) ends the pair with
    status 1 before directedMatch; the new map keeps its detection state."""
    frames, cam = small_stream
    kw = dict(keylines_ref=1500, keylines_max=2500, global_min_matches_threshold=100)
    orc = orc_mod.Oracle(params_for(orc_mod, cam, **kw))
    ctx = B.Context(params_for(B, cam, **kw))
    o0, g0 = orc.detect_u8(frames[0], 0), ctx.detect_u8(frames[0], 0)
    o1, g1 = orc.detect_u8(frames[1], 50000), ctx.detect_u8(frames[1], 50000)
    k0 = o0.keylines().copy()
    k0["rho"][:] = np.nan
    o0.set_keylines(k0)
    g0.upload(k0)
    po, pg = orc.track_pair(o0, o1), ctx.track_pair(g0, g1)
    assert po.status == 1 and pg.status == 1
    assert po.klm_num == 0 and pg.klm_num == 0
    assert (g1.keylines()["match_id"] < 0).all() and (o1.keylines()["match_id"] < 0).all()


def test_context_creation_leaves_signal_dispositions_untouched(B):
    """A library inside a ROS node must not touch signal dispositions: SIGSEGV / SIGBUS are the process's own before and
    after rebvio_hip_create (round 1 probed a host-visible mapping under a handler of its own)."""
    import ctypes
    import signal
    libc = ctypes.CDLL(None, use_errno=True)

    def disposition(sig):
        buf = ctypes.create_string_buffer(152)  # struct sigaction on x86-64 Linux (glibc): handler, 128-byte mask, flags, restorer
        assert libc.sigaction(int(sig), None, buf) == 0
        raw = buf.raw  # glibc fills the kernel's 8 mask bytes only: compare handler, those, the flags and the restorer
        return raw[0:16], raw[136:140], raw[144:152]

    before = disposition(signal.SIGSEGV), disposition(signal.SIGBUS)
    ctx = B.Context(B.default_params(96, 128, keylines_ref=500, keylines_max=1000))
    after = disposition(signal.SIGSEGV), disposition(signal.SIGBUS)
    ctx.close()
    assert before == after


def test_map_handle_outlives_its_context(B, small_stream):
    """rebvio_hip_destroy with map handles still out (round 2: a Map released after its Context dumped core in
    rebvio_hip_map_release): the handle turns inert - size / download answer -10, release frees only the husk - and a handle
    released BEFORE the destroy leaves nothing behind."""
    frames, cam = small_stream
    ctx = B.Context(params_for(B, cam, keylines_ref=1500, keylines_max=2000))
    a, b, c = (ctx.detect_u8(frames[i], i * 50000) for i in range(3))
    assert a.size() > 100
    c.release()
    ctx.close()                      # rebvio_hip_destroy
    with pytest.raises(B.HipError, match="destroyed"):
        a.size()
    with pytest.raises(B.HipError, match="destroyed"):
        b.keylines()
    assert np.isnan(a.threshold)
    a.release()
    del b                            # Map.__del__ -> rebvio_hip_map_release on the husk


def test_prior_promise_broken_is_reported(B, c2_stream):
    """rebvio_hip_track_pair_finish_async(..., R_prior_next) rotates the new map in place for the NEXT pair's first
    rotateKeylines. A next _begin with another prior - or after the gyro state was replaced - cannot use that map any more:
    -7 with a message, not a silently different rotation. The same promise kept goes through."""
    frames, cam = c2_stream
    ctx = B.Context(params_for(B, cam, **KW_C2))
    maps = [ctx.detect_u8(frames[i], i * 50000) for i in range(4)]
    a = 0.002
    Rn = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
    mid = ctx.track_pair_begin(maps[0], maps[1])
    ctx.track_pair_finish_async(maps[0], maps[1], *_vision_only_fusion(mid), R_prior_next=Rn)
    assert ctx.track_pair_result()[3] == 0
    with pytest.raises(B.HipError, match="R_prior"):
        ctx.track_pair_begin(maps[1], maps[2], R_prior=np.eye(3, dtype=np.float32))
    mid = ctx.track_pair_begin(maps[1], maps[2], R_prior=Rn)     # the promised prior: accepted
    ctx.track_pair_finish_async(maps[1], maps[2], *_vision_only_fusion(mid), R_prior_next=Rn)
    assert ctx.track_pair_result()[3] == 0
    bg, w = ctx.gyro_state()
    ctx.set_gyro_state(bg + np.float32(1e-3), w)                  # a re-initialised bias between the pairs
    with pytest.raises(B.HipError, match="gyro state"):
        ctx.track_pair_begin(maps[2], maps[3], R_prior=Rn)
    ctx.close()


def _stand_alone_records(B, cam, frames, order, **kw):
    ctx = B.Context(params_for(B, cam, **dict(KW_C2, **kw)))
    dev = ctx.upload_frames(frames)
    r = [pair_tuple(o, nk) for o, nk in run_stream(ctx, dev, order, cam.width * cam.height)]
    ctx.close()
    return r


def _batch_records(B, cam, streams, order, lens=None, **kw):
    L = len(streams)
    npx = cam.width * cam.height
    bat = B.Batch(params_for(B, cam, **dict(KW_C2, **kw)), L)
    if lens is not None:
        for lane in bat.lanes:
            lane.set_undistort(cam.fm, cam.fm, cam.cx, cam.cy, lens)
    devs = [bat.lanes[s].upload_frames(streams[s]) for s in range(L)]
    got = [[] for _ in range(L)]
    for k, i in enumerate(order):
        outs, nks = bat.push_u8_device([d + int(i) * npx for d in devs], k * 50000)
        for s in range(L):
            if outs[s].status >= 0:
                got[s].append(pair_tuple(outs[s], nks[s]))
    for outs, nks in bat.flush():
        for s in range(L):
            got[s].append(pair_tuple(outs[s], nks[s]))
    bat.close()
    return got


def test_two_batches_share_one_gpu(B, c2_stream):
    """Two 8-lane batches in ONE process on one GPU, their steps pushed alternately so that both have kernels in flight: the
    persistent LM kernel of a batch needs all its workgroups resident together, and a batch alone sizes its launches to the
    whole device (8 lanes x 32 workgroups of 512 threads at 16 k keylines). The device-wide registry of persistent-kernel users
    (api.hip, Residency) gives each batch its share - half the device each here, two launches per step instead of one - so both
    run to the end (slower, not into the bounded poll's -9) and every lane still produces the records of a stand-alone context."""
    from rebvio_amd import synth
    cam = c2_stream[1]
    n, L = 24, 8
    npx = cam.width * cam.height
    streams = [c2_stream[0]] + [synth.render_stream(cam.width, cam.height, 8, stream_id=s)[0] for s in range(1, L)]
    order = synth.pingpong_indices(8, n)
    bats = [B.Batch(params_for(B, cam, **KW_C2), L) for _ in range(2)]
    devs = [[bat.lanes[s].upload_frames(streams[(s + 3 * b) % L]) for s in range(L)] for b, bat in enumerate(bats)]
    got = [[[] for _ in range(L)] for _ in range(2)]
    for k, i in enumerate(order):
        for b, bat in enumerate(bats):
            outs, nks = bat.push_u8_device([d + int(i) * npx for d in devs[b]], k * 50000)
            for s in range(L):
                if outs[s].status >= 0:
                    got[b][s].append(pair_tuple(outs[s], nks[s]))
    for b, bat in enumerate(bats):
        for outs, nks in bat.flush():
            for s in range(L):
                got[b][s].append(pair_tuple(outs[s], nks[s]))
    for bat in bats:
        bat.close()
    want = {s: _stand_alone_records(B, cam, streams[s], order) for s in (0, 3, 7)}
    for b in range(2):
        for s in range(L):
            assert len(got[b][s]) == n - 1, (b, s, len(got[b][s]))
        for lane in (0, 4, 7):
            src = (lane + 3 * b) % L
            if src in want:
                assert got[b][lane] == want[src], (b, lane)


@pytest.mark.parametrize("L,env", [(3, {}), (4, {}), (8, {}), (4, {"REBVIO_HIP_LM": "seq"}), (4, {"REBVIO_HIP_LM": "spec3"}), (2, {"REBVIO_HIP_BATCH_DM_HEAD": "compact1"}),
                                   (5, {"REBVIO_HIP_BATCH_DM_HEAD": "compact8", "REBVIO_HIP_BATCH_LEAD": "6", "REBVIO_HIP_BATCH_GROUP": "3"}),
                                   (3, {"REBVIO_HIP_DETECT_WORKER": "0", "REBVIO_HIP_FUSE_DOG": "0"}),
                                   (3, {"REBVIO_HIP_BATCH_FUSE_DOG": "1"})],
                         ids=["3-lanes", "4-lanes", "8-lanes", "4-lanes-seq-lm", "4-lanes-spec3-lm", "2-lanes-compact1", "5-lanes-compact8-lead6-group3",
                              "3-lanes-caller-launches-unfused-dog",
                              "3-lanes-fused-candidate-kernel"])
def test_batched_lanes_equal_stand_alone_streams(B, c2_stream, monkeypatch, L, env):
    """rebvio_hip_batch_*: L camera streams advanced in lock-step by batched launches (lane = blockIdx.z) produce, lane by lane,
    the records of L stand-alone contexts fed the same frames - bit for bit (same kernel bodies, same per-lane reduction
    order, the same glue statements), only delivered on a different schedule. The lane counts cover what the bench's
    streams_per_gpu figures run: from 4 lanes on the directedMatch kernel runs one lane per keyline (k_directed_match_c_b<64, 1>)
    and the LM kernel polls slowly; 8 lanes fill the chip with the persistent LM workgroups; the sequential LM kernel
    (k_lm_chain_b) and both directedMatch forms at lane counts where they are not the default."""
    from rebvio_amd import synth
    cam = c2_stream[1]
    n = 36
    streams = [c2_stream[0]] + [synth.render_stream(cam.width, cam.height, 8, stream_id=s)[0] for s in range(1, L)]
    order = synth.pingpong_indices(8, n)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    got = _batch_records(B, cam, streams, order)
    for k in env:
        if k.startswith("REBVIO_HIP_BATCH") or k in ("REBVIO_HIP_DETECT_WORKER", "REBVIO_HIP_FUSE_DOG"):
            monkeypatch.delenv(k)
    want = [_stand_alone_records(B, cam, streams[s], order) for s in range(L)]
    for s in range(L):
        assert len(want[s]) == n - 1 == len(got[s]), (s, len(got[s]), len(want[s]))
        assert got[s] == want[s], s
        assert all(r[-2] == 0 for r in got[s])
    assert got[0] != got[1]  # different scenes per lane


@pytest.mark.parametrize("lens", [None, EUROC_D], ids=["pinhole", "radtan-front-end"])
def test_batched_lanes_are_bit_identical_to_the_oracle_with_the_sums_in_one_order(orc_mod, B, c2_stream, lens):
    """The batch driver against the restatement directly (not through the stand-alone contexts of the tests around it): four
    lanes, four scenes, 16 frames, the oracle adding its keyline sums in the kernels' order - every word of every lane's pair
    records identical; with the EuRoC lens model the frames pass through the batched device front end (x3 + undistort) on one
    side and the oracle's on the other."""
    from rebvio_amd import synth
    cam = c2_stream[1]
    L, n = 4, 16
    streams = [synth.render_stream(cam.width, cam.height, 8, stream_id=10 + s, dist=lens)[0] for s in range(L)]
    order = synth.pingpong_indices(8, n)
    npx = cam.width * cam.height
    bat = B.Batch(params_for(B, cam, **KW_C2), L)
    if lens is not None:
        for lane in bat.lanes:
            lane.set_undistort(cam.fm, cam.fm, cam.cx, cam.cy, lens)
    devs = [bat.lanes[s].upload_frames(streams[s]) for s in range(L)]
    got = [[] for _ in range(L)]
    for k, i in enumerate(order):
        outs, _ = bat.push_u8_device([d + int(i) * npx for d in devs], k * 50000)
        for s in range(L):
            if outs[s].status >= 0:
                got[s].append(_record_words(outs[s]))
    for outs, _ in bat.flush():
        for s in range(L):
            got[s].append(_record_words(outs[s]))
    bat.close()
    for s in range(L):
        orc = orc_mod.Oracle(params_for(orc_mod, cam, **KW_C2))
        orc.set_sum_order("device")
        prev, want = None, []
        for k, i in enumerate(order):
            if lens is None:
                m = orc.detect_u8(streams[s][i], k * 50000)
            else:
                m = orc.detect(orc.front_end_u8(streams[s][i], cam.fm, cam.fm, cam.cx, cam.cy, lens), k * 50000)
            if prev is not None:
                want.append(_record_words(orc.track_pair(prev, m)))
            prev = m
        assert len(got[s]) == len(want) == n - 1
        for k, (wo, wg) in enumerate(zip(want, got[s])):
            assert np.array_equal(wo, wg), (s, k, np.flatnonzero(wo != wg)[:8])


def test_batched_lanes_with_the_euroc_lens_model(B, c2_stream):
    """A batch takes the reference's own camera (camera.hpp:25-45: rad-tan distortion): with a lens model on every lane the
    batched front end (k_front_end_u8_b: x3 + undistort, rebvio.cpp:43-47) runs ahead of the scans, and every lane's records
    equal those of a stand-alone context with the same model. A model on only some lanes is refused."""
    from rebvio_amd import synth
    cam = c2_stream[1]
    L, n = 3, 20
    streams = [synth.render_stream(cam.width, cam.height, 8, stream_id=s, dist=EUROC_D)[0] for s in range(L)]
    order = synth.pingpong_indices(8, n)
    got = _batch_records(B, cam, streams, order, lens=EUROC_D)
    npx = cam.width * cam.height
    for s in range(L):
        ctx = B.Context(params_for(B, cam, **KW_C2))
        ctx.set_undistort(cam.fm, cam.fm, cam.cx, cam.cy, EUROC_D)
        dev = ctx.upload_frames(streams[s])
        want = [pair_tuple(o, nk) for o, nk in run_stream(ctx, dev, order, npx)]
        ctx.close()
        assert len(want) == n - 1 and got[s] == want, s
    plain = _batch_records(B, cam, streams[:2], order[:8])
    assert plain[0] and plain[0] != got[0][:len(plain[0])]      # the model matters
    bat = B.Batch(params_for(B, cam, **KW_C2), 2)
    bat.lanes[0].set_undistort(cam.fm, cam.fm, cam.cx, cam.cy, EUROC_D)
    devs = [bat.lanes[s].upload_frames(streams[s]) for s in range(2)]
    with pytest.raises(B.HipError, match="every lane or on none"):
        bat.push_u8_device(devs, 0)
    bat.close()


def test_sixteen_lanes_equal_stand_alone_streams(B, c2_stream):
    """The widest batch the ABI takes: 16 lanes x 16k keylines = 480 LM workgroups of 512 threads. The workgroups of a lane
    wait for each other's records, so a launch of the batched persistent kernel carries only as many lanes as the device holds
    together (lm_chain_b_max_lanes, track.hip: 8 at this size); the driver covers 16 lanes with two launches per step and every
    lane's records still equal a stand-alone context's. Lane counts outside 1..16 are refused."""
    from rebvio_amd import synth
    cam = c2_stream[1]
    L, n = 16, 14
    streams = [c2_stream[0]] + [synth.render_stream(cam.width, cam.height, 8, stream_id=s)[0] for s in range(1, L)]
    order = synth.pingpong_indices(8, n)
    got = _batch_records(B, cam, streams, order)
    for s in (0, 7, 15):
        want = _stand_alone_records(B, cam, streams[s], order)
        assert len(want) == n - 1 and got[s] == want, s
    assert all(len(g) == n - 1 and all(r[-2] == 0 for r in g) for g in got)
    assert len({tuple(g) for g in got}) == L  # sixteen different scenes, sixteen different record sets
    with pytest.raises(B.HipError, match="lanes must be in"):
        B.Batch(params_for(B, cam, **KW_C2), 17)


@pytest.mark.parametrize("stream_id", [0, 2])
def test_stream_divergence_report(orc_mod, B, stream_id):
    """SURVEY.md H4 in place of a flat pose tolerance (tools/divergence_report.py, profiles/r02_divergence_report.txt): GPU
    pipeline and CPU oracle run the same 640x480 stream with their state carried independently; per pair the
    Levenberg-Marquardt accept / reject decisions (core.cpp:172-183) are compared and the first differing decision is where the
    two trajectories may legitimately part. Bars, as long as the LM paths agree: the GPU translation is within 1e-2 (relative,
    observed <= 5.5e-3) of the oracle run with double-accumulated sums, and no farther from the fp32 oracle than that oracle is
    from its own double-accumulated run (+1e-2; the sequential fp32 sums are the noisier side: observed up to 4.8e-2). After a
    differing decision only the loose 5e-2 bar against the wide-sum run applies. On these streams no decision differs."""
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import divergence_report as D
    ref, wide, got = D.run(30, stream_id)
    first, rows = D.analyse(ref, wide, got)
    assert len(rows) >= 22
    for k, mo, mg, d_ref, d_wide, d_own, ko, kg in rows:
        agree = first is None or k < first
        if agree:
            assert d_wide <= 1e-2, (k, d_wide)
            assert d_ref <= d_own + 1e-2, (k, d_ref, d_own)
            assert abs(ko - kg) <= 0.01 * ko + 2, (k, ko, kg)
        else:
            assert d_wide <= 5e-2, (k, d_wide, "after the first differing LM decision at pair %d" % first)
    assert first is None or first >= 5, f"LM decisions differ from the oracle's already at pair {first}"
