"""CPU tests of the oracle against analytically known answers and the one KAT the reference
holds (rebvio/test/test_rebvio.cpp:8-17). The reference has no golden vectors for the hot path ("parity unpinned"):
these cases are what pins the restatement."""
import numpy as np
import pytest

from conftest import params_for


def test_integral_image_of_ones(orc_mod):
    L = orc_mod.lib()
    import ctypes as C
    h, w = 37, 53
    a = np.ones((h, w), np.float32)
    out = np.empty_like(a)
    fp = C.POINTER(C.c_float)
    L.orc_integral_image(h, w, a.ctypes.data_as(fp), out.ctypes.data_as(fp))
    rr, cc = np.mgrid[1:h + 1, 1:w + 1]
    assert np.array_equal(out, (rr * cc).astype(np.float32))  # (r+1)(c+1), exact in fp32


@pytest.mark.parametrize("d", [7, 9])
def test_box_mean_of_constant_is_constant_incl_borders(orc_mod, d):
    L = orc_mod.lib()
    import ctypes as C
    h, w = 64, 96
    fp = C.POINTER(C.c_float)
    a = np.full((h, w), 8.0, np.float32)  # power of two: every partial sum is exact
    ii = np.empty_like(a)
    out = np.empty_like(a)
    L.orc_integral_image(h, w, a.ctypes.data_as(fp), ii.ctypes.data_as(fp))
    L.orc_box_average(h, w, d, ii.ctypes.data_as(fp), out.ctypes.data_as(fp))
    assert np.abs(out - 8.0).max() <= 8.0 * 2e-7  # only the reciprocal-divisor rounding remains


def test_kovesi_widths(orc_mod):
    orc = orc_mod.Oracle(orc_mod.default_params(64, 64))
    assert [orc.L.orc_filter_width(orc.h, 0, k) for k in range(3)] == [7, 7, 7]  # sigma 3.56359
    assert [orc.L.orc_filter_width(orc.h, 1, k) for k in range(3)] == [9, 9, 9]  # sqrt(12) * 1.2599


@pytest.mark.parametrize("sigma,n", [(3.56359, 1), (3.56359, 2), (2.2, 4), (3.0, 5)])
def test_box_filter_with_n_passes_approximates_the_gaussian(orc_mod, sigma, n):
    """FastGaussian for n other than 3 (scale_space.cpp:14-41): Kovesi's widths give the stated sigma_true, the smoothed constant
    stays constant, and the filter's response to an impulse has the variance n passes of those boxes must have:
    sum((w^2 - 1) / 12) per axis - an independent statement of what the n-pass loop of smooth() computes."""
    h = w = 96
    orc = orc_mod.Oracle(orc_mod.default_params(h, w))
    flat, widths = orc.smooth(np.full((h, w), 64.0, np.float32), sigma, n)
    assert len(widths) == n and all(x % 2 == 1 for x in widths)
    assert np.abs(flat - 64.0).max() <= 64.0 * n * 3e-7
    var_expected = sum((x * x - 1) / 12.0 for x in widths)
    if n >= 3:  # (for fewer passes the reference's m = round(...) leaves [0, n] and every pass gets the wider box)
        assert abs(var_expected - sigma * sigma) <= 0.6 * sigma  # Kovesi: within the rounding of m box widths
    img = np.zeros((h, w), np.float32)
    img[h // 2, w // 2] = 4096.0
    out, _ = orc.smooth(img, sigma, n)
    out = out.astype(np.float64)
    assert abs(out.sum() - 4096.0) <= 4096.0 * 1e-5
    yy, xx = np.mgrid[0:h, 0:w]
    for axis, c in ((xx, w // 2), (yy, h // 2)):
        var = ((axis - c) ** 2 * out).sum() / out.sum()
        assert abs(var - var_expected) <= 1e-3 * var_expected, (var, var_expected, widths)


def test_constant_image_has_no_keylines(orc_mod):
    orc = orc_mod.Oracle(orc_mod.default_params(96, 128))
    m = orc.detect(np.full((96, 128), 300.0, np.float32))
    assert m.size() == 0
    assert (m.mask(96, 128) == -1).all()
    ss = orc.scale_space(np.full((96, 128), 300.0, np.float32))
    assert np.abs(ss["dog"]).max() < 1e-3 and ss["mag"].max() < 1e-6


def test_step_edge_keylines(orc_mod):
    """Ideal vertical step at column c0: the DoG zero crossing sits between c0-1 and c0 (x = c0 - 0.5), one keyline per
    row, DoG gradient along x (DoG = scale1 - scale0 falls across a rising edge: negative x), chained along the edge."""
    h, w, c0 = 96, 128, 60
    img = np.zeros((h, w), np.float32)
    img[:, c0:] = 600.0
    orc = orc_mod.Oracle(orc_mod.default_params(h, w, gain=0.0, threshold=0.01))
    m = orc.detect(img)
    kl = m.keylines()
    rows_inner = h - 4
    assert len(kl) >= rows_inner  # one per row in [2, h-2), (border effects may add a few near the top/bottom)
    main = kl[np.abs(kl["pos"][:, 0] - (c0 - 0.5)) < 0.51]
    assert len(main) >= rows_inner - 2
    assert (np.abs(main["gradient"][:, 1]) < 1e-3 * np.abs(main["gradient"][:, 0]) + 1e-6).all()
    assert (main["gradient"][:, 0] < 0).all()
    assert np.allclose(main["pos_img"][:, 0], main["pos"][:, 0] - orc.p.cx)
    # init values of KeyLine (types/keyline.hpp:42-59)
    assert (kl["rho"] == 1.0).all() and (kl["sigma_rho"] == 20.0).all() and (kl["match_id"] == -1).all()
    linked = (main["id_next"] >= 0).sum()
    assert linked >= len(main) - 2


def test_threshold_servo_and_truncation(orc_mod, small_stream):
    frames, cam = small_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam, keylines_ref=150, keylines_max=200))
    m = orc.detect_u8(frames[0])
    # first frame: previous count 0 -> threshold -= gain * keylines_ref (edge_detector.cpp:33-36)
    assert np.float32(orc.threshold) == np.float32(np.float32(0.01) - np.float32(5e-7) * np.float32(150))
    assert m.size() == 200
    mask = m.mask(cam.height, cam.width)
    ys, xs = np.nonzero(mask >= 0)
    assert np.array_equal(mask[ys, xs], np.arange(200))  # raster rank == index, nothing after the 200th
    t0 = orc.threshold
    orc.detect_u8(frames[1])
    assert np.float32(orc.threshold) == np.float32(np.float32(t0) - np.float32(5e-7) * np.float32(150 - 200))


def test_quantile_rule(orc_mod, small_stream):
    frames, cam = small_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam))
    m = orc.detect_u8(frames[0])
    kl = m.keylines()
    n = len(kl)
    assert orc.quantile(m, 0.9, 100) == pytest.approx(1e3)  # all sigma_rho = 20 -> last bin -> never exceeded
    kl["sigma_rho"] = np.linspace(0.01, 5.0, n).astype(np.float32)
    m.set_keylines(kl)
    q = orc.quantile(m, 0.5, 100)
    # first bin edge whose cumulative count (before adding the bin) exceeds 0.5 n
    edges = np.arange(100) * (20.0 - 1e-3) / 100 + 1e-3
    hist = np.histogram(kl["sigma_rho"], bins=np.append(edges, 1e9))[0]
    a = 0
    exp = 1e3
    for i in range(100):
        if a > 0.5 * n:
            exp = edges[i]
            break
        a += hist[i]
    assert q == pytest.approx(exp, rel=1e-5)


def test_distance_field_winner_rule(orc_mod):
    """DistanceField::build: smallest |r| wins, ties go to the LAST (idx, r) visited (core.hpp:54-56)."""
    h, w = 64, 64
    orc = orc_mod.Oracle(orc_mod.default_params(h, w))
    m = orc.detect(np.full((h, w), 10.0, np.float32))  # empty map to fill by hand
    kl = np.zeros(2, orc_mod.KEYLINE_DTYPE)
    for i, x in enumerate((20.0, 26.0)):
        kl[i]["pos"] = (x, 32.0)
        kl[i]["gradient"] = (1.0, 0.0)
        kl[i]["gradient_norm"] = 1.0
        kl[i]["rho"], kl[i]["sigma_rho"] = 1.0, 20.0
    m.set_keylines(kl)
    m.threshold = 0.5
    orc.build_distance_field(m)
    ids, dist = orc.distance_field()
    row = ids[32]
    assert row[20] == 0 and row[26] == 1
    assert row[22] == 0 and dist[32, 22] == 2 and row[24] == 1 and dist[32, 24] == 2
    assert row[23] == 1 and dist[32, 23] == 3  # tie at |r| = 3: the later keyline wins
    assert (ids[:31] == -1).all() and (ids[34:] == -1).all()


def test_forward_match_rule(orc_mod):
    """forwardMatch: the writer with the largest rho wins, ties -> largest index (edge_map.cpp:83-93)."""
    h, w = 64, 64
    orc = orc_mod.Oracle(orc_mod.default_params(h, w))
    old = orc.detect(np.full((h, w), 10.0, np.float32))
    new = orc.detect(np.full((h, w), 10.0, np.float32))
    ko = np.zeros(4, orc_mod.KEYLINE_DTYPE)
    ko["match_id_forward"] = [0, 0, 0, -1]
    ko["rho"] = [2.0, 3.0, 3.0, 9.0]
    ko["sigma_rho"] = [0.1, 0.2, 0.3, 0.4]
    ko["matches"] = [5, 6, 7, 8]
    ko["match_id_keyframe"] = -1
    kn = np.zeros(1, orc_mod.KEYLINE_DTYPE)
    kn["match_id"] = -1
    kn["rho"], kn["sigma_rho"] = 1.0, 20.0
    old.set_keylines(ko)
    new.set_keylines(kn)
    assert orc.forward_match(old, new) == 3  # every writer passes the >= test in sequence
    r = new.keylines()[0]
    assert r["match_id"] == 2 and r["rho"] == 3.0 and r["sigma_rho"] == np.float32(0.3) and r["matches"] == 8


def test_ls4_acceleration_kat(orc_mod):
    """The reference's only unit test (rebvio/test/test_rebvio.cpp:8-17)."""
    import ctypes as C
    orc = orc_mod.Oracle(orc_mod.default_params(64, 64))
    fp = C.POINTER(C.c_float)
    Vgv = np.array([-4.06833e-05, 9.40667e-05, 5.70767e-05], np.float32)
    dt = np.float32(0.05)
    vel = (-Vgv / dt).astype(np.float32)
    Av = np.zeros(3, np.float32)
    R = np.array([1, 8.83134e-05, -7.48149e-05, -8.831e-05, 1, 4.57494e-05, 7.4819e-05, -4.57428e-05, 1], np.float32)
    orc.L.orc_ls4_reset(orc.h)
    orc.L.orc_estimate_ls4_acceleration(orc.h, vel.ctypes.data_as(fp), Av.ctypes.data_as(fp), R.ctypes.data_as(fp), dt)
    assert Av[0] == pytest.approx(0.0162734, abs=1e-5)
    assert Av[1] == pytest.approx(-0.0376267, abs=1e-5)
    assert Av[2] == pytest.approx(-0.0228307, abs=1e-5)


def test_so3_exp_and_solver(orc_mod):
    import ctypes as C
    L = orc_mod.lib()
    fp = C.POINTER(C.c_float)
    w = np.array([0.01, -0.02, 0.03], np.float32)
    R = np.zeros(9, np.float32)
    L.orc_so3_exp(w.ctypes.data_as(fp), R.ctypes.data_as(fp))
    R = R.reshape(3, 3).astype(np.float64)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6) and np.linalg.det(R) == pytest.approx(1.0, abs=1e-6)
    from scipy.spatial.transform import Rotation
    assert np.allclose(R, Rotation.from_rotvec(w.astype(np.float64)).as_matrix(), atol=1e-6)
    rng = np.random.default_rng(1)
    A = rng.normal(size=(6, 6))
    A = (A @ A.T + 6 * np.eye(6)).astype(np.float32)
    b = rng.normal(size=6).astype(np.float32)
    x = np.zeros(6, np.float32)
    L.orc_sym6_solve(A.ctypes.data_as(fp), b.ctypes.data_as(fp), x.ctypes.data_as(fp))
    assert np.allclose(x, np.linalg.solve(A.astype(np.float64), b.astype(np.float64)), rtol=1e-4, atol=1e-6)


def test_tracking_recovers_translation_direction(orc_mod, small_stream):
    """End-to-end sanity on the synthetic stream: the estimated translation opposes the camera motion and is stable."""
    frames, cam = small_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam, global_min_matches_threshold=50))
    prev = orc.detect_u8(frames[0])
    vs = []
    for i in range(1, len(frames)):
        m = orc.detect_u8(frames[i], i * 50000)
        out = orc.track_pair(prev, m)
        assert out.status == 0
        vs.append(np.array(out.Vg))
        prev = m
    v = np.array(vs[3:])
    assert (v[:, 0] < 0).all()  # scene moves towards -x for a camera moving +x
    assert np.std(v[:, 0]) < 0.5 * np.abs(np.mean(v[:, 0]))


def test_sum_orders_differ_by_rounding_only(orc_mod, small_stream):
    """The keyline sums of tryVel / extRotVel in the restatement's three orders - the reference's (fp32, index order), the
    double-accumulating diagnostic, the HIP kernels' tree (set_sum_order("device"): what the bit-for-bit GPU tests run the
    oracle with) - are sums of the SAME fp32 terms: a single tryVel's score and 3x3 / 3 sums agree to the rounding of a
    few-thousand-term fp32 sum, every per-keyline output (residuals, forward matches) is identical, and the association is a
    real one (the fp32 orders do not give the same bits). CPU only: the tree itself is pinned against the device by
    tests/test_parity_gpu.py::test_lm_sums_in_device_order_are_bit_exact."""
    frames, cam = small_stream
    got = {}
    for order in ("reference", "wide", "device"):
        orc = orc_mod.Oracle(params_for(orc_mod, cam, global_min_matches_threshold=50))
        prev = orc.detect_u8(frames[0])
        for i in range(1, 4):  # realistic depths / matches in the reference's order, then ONE evaluation in the order under test
            m = orc.detect_u8(frames[i], i * 50000)
            orc.track_pair(prev, m)
            prev = m
        m = orc.detect_u8(frames[4], 4 * 50000)
        orc.build_distance_field(m)
        orc.set_sum_order(order)
        res = np.zeros(prev.size(), np.float32)
        score, J, F = orc.try_vel(prev, [-0.01, -0.004, 0.002], orc.quantile(prev), res)
        got[order] = (np.float64(score), np.asarray(J, np.float64), np.asarray(F, np.float64), res.copy(), prev.keylines()["match_id_forward"].copy())
    ref, wide, dev = got["reference"], got["wide"], got["device"]
    assert (ref[4] >= 0).sum() > 500
    for other in (wide, dev):
        assert np.array_equal(ref[3].view(np.uint32), other[3].view(np.uint32)) and np.array_equal(ref[4], other[4])
        assert abs(other[0] - ref[0]) <= 2e-4 * abs(ref[0])
        assert np.abs(other[1] - ref[1]).max() <= 2e-4 * np.abs(np.diag(ref[1])).max()
        assert np.abs(other[2] - ref[2]).max() <= 2e-4 * np.sqrt(ref[0] * np.abs(np.diag(ref[1])).max())
    assert abs(dev[0] - wide[0]) <= abs(ref[0] - wide[0]) + 1e-6 * abs(wide[0])  # a tree is no worse than the running sum
    assert np.float32(dev[0]) != np.float32(ref[0]) or not np.array_equal(np.float32(dev[1]), np.float32(ref[1]))


# ---- front end: convertTo(CV_32F, 3.0) + cv::undistort -------------------------------------------------
EUROC_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0]  # camera.hpp:31-35


def _numpy_undistort(img, fm, cx, cy, D):
    """Independent float64 statement of the same published algorithm (map in double, 1/32 px quantisation, bilinear with
    zero border); every product is exact, so float64 == the fp32 result."""
    H, W = img.shape
    j, i = np.meshgrid(np.arange(W), np.arange(H))
    k1, k2, p1, p2, k3 = [float(np.float32(v)) for v in D]
    fm, cx, cy = float(np.float32(fm)), float(np.float32(cx)), float(np.float32(cy))
    x, y = (j - cx) / fm, (i - cy) / fm
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * 2 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + p2 * 2 * x * y
    iu, iv = np.rint((fm * xd + cx) * 32).astype(int), np.rint((fm * yd + cy) * 32).astype(int)
    sx, sy, ax, ay = iu >> 5, iv >> 5, (iu & 31) / 32.0, (iv & 31) / 32.0
    src = img.astype(np.float64) * 3

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        return np.where(ok, src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], 0.0)

    return ((1 - ay) * (1 - ax) * tap(sy, sx) + (1 - ay) * ax * tap(sy, sx + 1) + ay * (1 - ax) * tap(sy + 1, sx)
            + ay * ax * tap(sy + 1, sx + 1))


def test_front_end_identity_and_radial_model(orc_mod, small_stream):
    frames, cam = small_stream
    orc = orc_mod.Oracle(params_for(orc_mod, cam))
    img = np.maximum(frames[0], 1)   # no zero pixels, so a zero in the output can only be the border constant
    same = orc.front_end_u8(img, cam.fm, cam.fm, cam.cx, cam.cy, [0, 0, 0, 0, 0])
    assert np.array_equal(same, img.astype(np.float32) * np.float32(3.0))       # no distortion: x3 only, borders included
    out = orc.front_end_u8(img, cam.fm, cam.fm, cam.cx, cam.cy, EUROC_D)
    ref = _numpy_undistort(img, cam.fm, cam.cx, cam.cy, EUROC_D)
    assert np.abs(ref - out).max() <= 0.75                                      # a half-ulp map tie moves one tap by 1/32 px
    assert (ref != out).mean() < 1e-3
    # barrel distortion (k1 < 0): the undistorted view samples inside the distorted frame -> no border pixels, and the
    # principal point is a fixed point of the map
    assert out.min() > 0
    ci, cj = int(round(cam.cy)), int(round(cam.cx))
    assert abs(out[ci, cj] - 3.0 * img[ci, cj]) <= 3.0 * np.abs(np.diff(img[ci, cj - 1:cj + 2].astype(np.float32))).max()
    # pincushion (k1 > 0) reaches outside the frame: constant-zero border shows up in the corners
    pin = orc.front_end_u8(img, cam.fm, cam.fm, cam.cx, cam.cy, [0.6, 0, 0, 0, 0])
    assert pin[0, 0] == 0.0 and pin[-1, -1] == 0.0 and pin[ci, cj] > 0
