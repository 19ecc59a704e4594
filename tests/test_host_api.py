"""The C++ host API (include/rebvio/*.hpp + rebvio_amd/host): callers compile unchanged, the reference's own unit test
passes against it (CPU), and on the GPU box the ros_rebvio-style example runs a stream end to end."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "rebvio_amd", "_build")
INC = ["-I", os.path.join(ROOT, "include")]


@pytest.fixture(scope="module")
def host_lib():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rebvio_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "rebvio_amd", "host")], check=True)
    assert os.path.exists(os.path.join(BUILD, "librebvio.so"))
    return BUILD


def test_ros_rebvio_call_sites_compile():
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only"] + INC + [os.path.join(ROOT, "tests", "cpp", "callers_compile.cpp")],
                   check=True)


def test_reference_unit_test_ls4_kat(host_lib, tmp_path):
    exe = str(tmp_path / "ls4_kat")
    subprocess.run(["g++", "-std=c++17", "-O1"] + INC + [os.path.join(ROOT, "tests", "cpp", "test_ls4_kat.cpp"), "-o", exe,
                    "-L", host_lib, "-lrebvio", "-lrebvio_hip", f"-Wl,-rpath,{host_lib}", "-pthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_sab_estimator_analytic_cases(host_lib, tmp_path):
    exe = str(tmp_path / "sab_host")
    subprocess.run(["g++", "-std=c++17", "-O1"] + INC + [os.path.join(ROOT, "tests", "cpp", "test_sab_host.cpp"), "-o", exe,
                    "-L", host_lib, "-lrebvio", "-lrebvio_hip", f"-Wl,-rpath,{host_lib}", "-pthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_stream_io_png_asl_replay_and_odometry_format(host_lib, tmp_path):
    """SURVEY.md N4 on the CPU: PNG decoding (all five scanline filters, split IDAT), EuRoC folder parsing with unordered
    csv rows, replay ordering, and the odometry text format pinned by the reference's own regression file."""
    from pngutil import write_asl, write_png
    rng = np.random.default_rng(5)
    n, W, H = 6, 64, 40
    frames = rng.integers(0, 256, (n, H, W), dtype=np.uint8)
    frames[1] = np.arange(W, dtype=np.uint8)[None, :] * 3          # smooth rows: exercises Sub / Average / Paeth predictions
    ts = 1000000 + np.arange(n) * 50000
    its = np.concatenate([ts[i] + (np.arange(10) + 1) * 5000 for i in range(n - 1)])
    gyro = np.tile(np.array([0.0, 0.25, 0.0], np.float32), (len(its), 1))
    acc = np.tile(np.array([0.0, 0.0, -9.5], np.float32), (len(its), 1))
    write_asl(str(tmp_path / "mav0"), frames, ts, its, gyro, acc, shuffle_seed=3)
    frames.tofile(tmp_path / "frames.u8")
    exe = str(tmp_path / "stream_io")
    subprocess.run(["g++", "-std=c++17", "-O1"] + INC + [os.path.join(ROOT, "tests", "cpp", "test_stream_io.cpp"), "-o", exe,
                    "-L", host_lib, "-lrebvio", "-lrebvio_hip", f"-Wl,-rpath,{host_lib}", "-pthread"], check=True)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    write_png(str(tmp_path / "rgb.png"), rgb, filters=(4, 1))
    ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14
     ).astype(np.uint8).tofile(tmp_path / "rgb.want")
    g16 = rng.integers(0, 65536, (H, W), dtype=np.uint16)
    write_png(str(tmp_path / "g16.png"), g16, filters=(3, 2), depth=16)
    (g16 >> 8).astype(np.uint8).tofile(tmp_path / "g16.want")
    r = subprocess.run([exe, str(tmp_path / "mav0"), str(tmp_path / "frames.u8"), str(W), str(H), str(n),
                        os.path.join(ROOT, "tests", "golden", "MH_03_medium_test_15s-30s_odometry.txt"),
                        str(tmp_path / "rgb.png"), str(tmp_path / "rgb.want"), str(tmp_path / "g16.png"), str(tmp_path / "g16.want")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_host_library_exports_reference_classes(host_lib):
    out = subprocess.run(["nm", "-DC", os.path.join(host_lib, "librebvio.so")], capture_output=True, text=True, check=True).stdout
    for sym in ("rebvio::Rebvio::Rebvio(rebvio::RebvioConfig&)", "rebvio::Rebvio::imageCallback(rebvio::types::Image&&)",
                "rebvio::Rebvio::imuCallback(rebvio::types::Imu&&)", "rebvio::EdgeDetector::detect(rebvio::types::Image&)",
                "rebvio::EdgeMap::rotateKeylines", "rebvio::EdgeMap::directedMatch", "rebvio::EdgeMap::regularize1Iter",
                "rebvio::Core::minimizeVel", "rebvio::Core::extRotVel", "rebvio::Core::updateInverseDepth",
                "rebvio::Core::buildDistanceField", "rebvio::Core::tryVel", "rebvio::Core::estimateBias",
                "rebvio::SABEstimator::gaussNewton", "rebvio::SABEstimator::problem",
                "rebvio::DistanceField::build", "rebvio::DistanceField::operator[](int)", "rebvio::Core::testfk",
                "rebvio::Core::calculatefJ", "rebvio::Core::updateInverseDepthARLU", "rebvio::EdgeMap::searchMatch",
                "rebvio::ScaleSpace::build", "rebvio::ScaleSpace::dog() const", "rebvio::ScaleSpace::mag() const",
                "rebvio::FastGaussian::smooth"):
        assert sym in out, sym


@pytest.mark.gpu
def test_stream_example_runs_like_ros_rebvio(host_lib, tmp_path):
    from rebvio_amd import synth
    n = 20
    frames, cam = synth.render_stream(320, 240, n)
    p = tmp_path / "frames.u8"
    frames.tofile(p)
    exe = os.path.join(host_lib, "rebvio_stream_example")
    r = subprocess.run([exe, str(p), "320", "240", str(n), str(cam.fm), str(cam.cx), str(cam.cy), "3000", "4000"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln and ln[0].isdigit()]
    assert len(lines) == n - 1
    vals = np.array([[float(x) for x in ln.split()] for ln in lines])
    assert np.isfinite(vals).all()
    assert (vals[:15, 1:7] == 0).all()     # pose integration starts after 4 + init_bias_frame_num frames (rebvio.cpp:263)
    assert np.abs(vals[-1, 4:]).max() > 0  # and then moves
    assert "running=1" in r.stderr


@pytest.mark.gpu
def test_edge_maps_outlive_their_pipeline(host_lib, tmp_path):
    """Map lifetime (rebvio.hpp:104-109, ros_rebvio.cpp:32-51): EdgeMap::SharedPtrs kept by an edge-image consumer stay readable
    after ~Rebvio and release cleanly; at the C-ABI a handle that outlives rebvio_hip_destroy answers -10 and its release frees
    only the husk (round 2 dumped core here: a release dereferenced the freed pool)."""
    from rebvio_amd import synth
    n = 6
    frames, cam = synth.render_stream(256, 192, n)
    p = tmp_path / "frames.u8"
    frames.tofile(p)
    exe = str(tmp_path / "map_lifetime")
    subprocess.run(["g++", "-std=c++17", "-O1"] + INC + [os.path.join(ROOT, "tests", "cpp", "test_map_lifetime.cpp"), "-o", exe,
                    "-L", host_lib, "-lrebvio", "-lrebvio_hip", f"-Wl,-rpath,{host_lib}", "-pthread"], check=True)
    r = subprocess.run([exe, str(p), "256", "192", str(n), repr(cam.fm), repr(cam.cx), repr(cam.cy)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", (r.returncode, r.stdout[-1000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_host_class_modes_publish_the_same_records(host_lib, tmp_path):
    """rebvio::Rebvio with its edge-image callback reading a keyline of every fresh map (as ros_rebvio.cpp:44 does), under the
    backend's remaining switches: the sequential and the per-call LM kernels, the other workgroup sizes, the
    one-lane-per-keyline directedMatch kernel, flat stream priorities. None of them may change a digit of the output."""
    from rebvio_amd import synth
    n = 40
    frames, cam = synth.render_stream(320, 240, n)
    scene = synth.make_scene(0)
    ts, gyro, acc = synth.imu_samples(scene, n, noise_seed=1)
    fp, ip = tmp_path / "frames.u8", tmp_path / "imu.bin"
    frames.tofile(fp)
    _write_imu(ip, ts, gyro, acc)
    exe = os.path.join(host_lib, "rebvio_stream_example")

    def run(**env):
        r = subprocess.run([exe, str(fp), "320", "240", str(n), str(cam.fm), str(cam.cx), str(cam.cy), "3000", "4000", str(ip), "100"],
                           capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        assert r.returncode == 0, (env, r.stdout[-1000:], r.stderr[-1000:])
        lines = [ln for ln in r.stdout.strip().splitlines() if ln and ln[0].isdigit()]
        assert len(lines) == n - 1
        return lines

    base = run()
    assert run(REBVIO_HIP_LM="seq") == base
    assert run(REBVIO_HIP_LM="percall") == base
    assert run(REBVIO_HIP_LM_THREADS="1024") == base
    assert run(REBVIO_HIP_DM_HEAD="compact1") == base
    assert run(REBVIO_HIP_PRIO="flat") == base


def _write_imu(path, ts, gyro, acc):
    rec = np.zeros(len(ts), dtype=[("ts", "<i8"), ("gyro", "<f4", 3), ("acc", "<f4", 3)])
    rec["ts"], rec["gyro"], rec["acc"] = ts, gyro, acc
    assert rec.dtype.itemsize == 32
    rec.tofile(path)


EUROC_D = [-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0]  # camera.hpp:31-35


# (size, frame period, keyline budget, min matches): the small 20 Hz stream of round 1, and BASELINE config 5 as stated -
# 640x480 @ 30 Hz camera + 200 Hz IMU, the bench's keyline budget, the reference's default match threshold
VIO_SMALL = (256, 192, 50000, 2500, 3500, 100)
VIO_CONFIG5 = (640, 480, 33333, 15000, 16000, 500)


@pytest.mark.gpu
@pytest.mark.parametrize("dist,shape", [(None, VIO_SMALL), (EUROC_D, VIO_SMALL), (None, VIO_CONFIG5), (EUROC_D, VIO_CONFIG5)],
                         ids=["pinhole", "radtan", "config5-640x480-30Hz", "config5-640x480-30Hz-radtan"])
def test_full_vio_config5_tracks_oracle(host_lib, tmp_path, orc_mod, dist, shape):
    """BASELINE config 5 (camera + IMU, SAB scale/attitude/bias filter): rebvio::Rebvio on the device against the oracle's
    restatement of rebvio.cpp:92-293 on the same frames and IMU samples. The per-keyline work is bit-exact, the reductions
    feeding the 3x3 / 6x6 / 7x7 solves are not (REL_SUM in test_parity_gpu.py), so the fused state is compared in tolerance:
    1e-4 rad / 1e-4 m on pose after 29 pairs, 1e-4 on scale, 2e-3 m/s^2 on gravity, 2e-6 rad/frame on gyro bias (observed
    on MI355X: <= 2e-6 everywhere, match counts identical). The "radtan" case adds the EuRoC lens model: the MONO8 frames
    then pass through the device front end (x3 + undistort) on one side and the oracle's on the other.
    The two "config5" cases are BASELINE config 5 at its stated size: 640x480 @ 30 Hz (frame period 33 333 us) with a
    200 Hz IMU, ~15k keylines, the reference's default global_min_matches_threshold = 500; same tolerances."""
    from rebvio_amd import synth
    W, H, dt_us, kref, kmax, min_matches = shape
    n = 30
    frames, cam = synth.render_stream(W, H, n, dist=dist)
    scene = synth.make_scene(0)
    ts, gyro, acc = synth.imu_samples(scene, n, frame_dt_us=dt_us, noise_seed=1)
    assert ts[1] - ts[0] == 5000  # 200 Hz
    fp, ip = tmp_path / "frames.u8", tmp_path / "imu.bin"
    frames.tofile(fp)
    _write_imu(ip, ts, gyro, acc)
    exe = os.path.join(host_lib, "rebvio_stream_example")
    env = dict(os.environ, REBVIO_EXAMPLE_FRAME_DT_US=str(dt_us))
    if dist is not None:
        env["REBVIO_EXAMPLE_DISTORTION"] = ",".join(repr(float(np.float32(v))) for v in dist)
    r = subprocess.run([exe, str(fp), str(W), str(H), str(n), str(cam.fm), str(cam.cx), str(cam.cy), str(kref), str(kmax), str(ip),
                        str(min_matches)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = np.array([[float(x) for x in ln.split()] for ln in r.stdout.strip().splitlines() if ln and ln[0].isdigit()])
    assert got.shape == (n - 1, 15)

    p = orc_mod.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=kref, keylines_max=kmax,
                               global_min_matches_threshold=min_matches)
    orc = orc_mod.Oracle(p)
    orc.vio_reset()
    prev, k, want = None, 0, []
    for i in range(n):
        if dist is None:
            m = orc.detect_u8(frames[i], i * dt_us)
        else:
            m = orc.detect(orc.front_end_u8(frames[i], cam.fm, cam.fm, cam.cx, cam.cy, dist), i * dt_us)
        while k < len(ts) and ts[k] <= i * dt_us:
            orc.vio_add_imu(m, ts[k], gyro[k], acc[k])
            k += 1
        if prev is not None:
            o = orc.vio_step(prev, m)
            assert o.pair.status == 0
            want.append([i * dt_us] + list(o.orientation) + list(o.position) + [o.K] + list(o.g_est) + list(o.Bg) + [o.pair.klm_num])
        prev = m
    want = np.array(want)
    assert (got[:, 0] == want[:, 0]).all()
    assert want[-1, 7] > 0 and np.abs(want[-1, 4:7]).max() > 0.05      # the filter ran and the pose moved
    np.testing.assert_allclose(got[:, 1:4], want[:, 1:4], atol=1e-4)    # orientation
    np.testing.assert_allclose(got[:, 4:7], want[:, 4:7], atol=1e-4)    # position
    np.testing.assert_allclose(got[:, 7], want[:, 7], atol=1e-4)        # scale
    np.testing.assert_allclose(got[:, 8:11], want[:, 8:11], atol=2e-3)  # gravity
    np.testing.assert_allclose(got[:, 11:14], want[:, 11:14], atol=2e-6)  # gyro bias
    # directedMatch counts: identical +-2 at 256x192; at 640x480 the ~1e-6 velocity difference (tree vs running sums over
    # 15k keylines) moves a few acceptance tests out of ~14 500 (observed <= 16)
    assert (np.abs(got[:, 14] - want[:, 14]) <= 2 + 0.002 * want[:, 14]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dist", [None, EUROC_D], ids=["config5-640x480-30Hz", "config5-640x480-30Hz-radtan"])
def test_full_vio_config5_is_bit_identical_with_the_sums_in_one_order(host_lib, tmp_path, orc_mod, dist):
    """BASELINE config 5 without a tolerance: rebvio::Rebvio (camera + 200 Hz IMU, front end, detection, both device halves of a
    pair, Ls4 / mean acceleration, the scale-attitude-bias filter, gravity-aligned pose integration) against the oracle's
    restatement of rebvio.cpp:92-293 with its keyline sums added in the kernels' order (set_sum_order("device"), a diagnostic
    of the restatement) - orientation, position, scale, gravity, gyro bias and match count of all 29 odometry records equal as
    32-bit floats. What test_full_vio_config5_tracks_oracle bounds by 1e-4 is therefore the association of those sums and
    nothing else on the path, host fusion included."""
    from rebvio_amd import synth
    W, H, dt_us, kref, kmax, min_matches = VIO_CONFIG5
    n = 30
    frames, cam = synth.render_stream(W, H, n, dist=dist)
    ts, gyro, acc = synth.imu_samples(synth.make_scene(0), n, frame_dt_us=dt_us, noise_seed=1)
    fp, ip = tmp_path / "frames.u8", tmp_path / "imu.bin"
    frames.tofile(fp)
    _write_imu(ip, ts, gyro, acc)
    env = dict(os.environ, REBVIO_EXAMPLE_FRAME_DT_US=str(dt_us), REBVIO_EXAMPLE_PRECISE="1")
    if dist is not None:
        env["REBVIO_EXAMPLE_DISTORTION"] = ",".join(repr(float(np.float32(v))) for v in dist)
    r = subprocess.run([os.path.join(host_lib, "rebvio_stream_example"), str(fp), str(W), str(H), str(n), str(cam.fm), str(cam.cx), str(cam.cy),
                        str(kref), str(kmax), str(ip), str(min_matches)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = np.array([[float(x) for x in ln.split()] for ln in r.stdout.strip().splitlines() if ln and ln[0].isdigit()])
    assert got.shape == (n - 1, 15)
    orc = orc_mod.Oracle(orc_mod.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=kref, keylines_max=kmax,
                                                global_min_matches_threshold=min_matches))
    orc.set_sum_order("device")
    orc.vio_reset()
    prev, k, want = None, 0, []
    for i in range(n):
        m = orc.detect_u8(frames[i], i * dt_us) if dist is None else orc.detect(orc.front_end_u8(frames[i], cam.fm, cam.fm, cam.cx, cam.cy, dist), i * dt_us)
        while k < len(ts) and ts[k] <= i * dt_us:
            orc.vio_add_imu(m, ts[k], gyro[k], acc[k])
            k += 1
        if prev is not None:
            o = orc.vio_step(prev, m)
            assert o.pair.status == 0
            want.append(list(o.orientation) + list(o.position) + [o.K] + list(o.g_est) + list(o.Bg) + [o.pair.klm_num])
        prev = m
    want32 = np.array(want, np.float32)
    got32 = got[:, 1:].astype(np.float32)
    bad = np.argwhere(want32.view(np.uint32) != got32.view(np.uint32))
    assert bad.size == 0, (len(bad), [(int(r_), int(c), float(got32[r_, c]), float(want32[r_, c])) for r_, c in bad[:6]])
    assert want32[-1, 6] > 0 and np.abs(want32[-1, 3:6]).max() > 0.05  # the filter ran and the pose moved


@pytest.mark.gpu
def test_replay_asl_folder_equals_raw_stream(host_lib, tmp_path):
    """rebvio_replay over an EuRoC-layout folder (PNG frames, csv IMU) and over the same data as raw files writes the same
    odometry file, in the reference's regression format, and agrees with the ros_rebvio-style example."""
    from pngutil import write_asl
    from rebvio_amd import synth
    n, W, H = 22, 256, 192
    frames, cam = synth.render_stream(W, H, n)
    ts, gyro, acc = synth.imu_samples(synth.make_scene(0), n, noise_seed=1)
    write_asl(str(tmp_path / "mav0"), frames, np.arange(n) * 50000, ts, gyro, acc, shuffle_seed=1, filters=(0, 4))
    frames.tofile(tmp_path / "frames.u8")
    _write_imu(tmp_path / "imu.bin", ts, gyro, acc)
    common = ["--camera", str(cam.fm), str(cam.cx), str(cam.cy), "--keylines", "2500", "3500", "--min-matches", "100"]
    exe = os.path.join(host_lib, "rebvio_replay")
    a = subprocess.run([exe, "--asl", str(tmp_path / "mav0"), "--out", str(tmp_path / "a.txt")] + common, capture_output=True, text=True,
                       timeout=300)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([exe, "--raw", str(tmp_path / "frames.u8"), "--size", str(W), str(H), "--imu", str(tmp_path / "imu.bin"),
                        "--out", str(tmp_path / "b.txt")] + common, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    ta, tb = (tmp_path / "a.txt").read_text(), (tmp_path / "b.txt").read_text()
    assert ta == tb and len(ta.splitlines()) == n - 1
    vals = np.array([[float(x) for x in ln.split()] for ln in ta.splitlines()])
    assert (vals[:15, 1:] == 0).all() and np.abs(vals[-1, 4:]).max() > 0
    ex = subprocess.run([os.path.join(host_lib, "rebvio_stream_example"), str(tmp_path / "frames.u8"), str(W), str(H), str(n), str(cam.fm),
                         str(cam.cx), str(cam.cy), "2500", "3500", str(tmp_path / "imu.bin"), "100"], capture_output=True, text=True,
                        timeout=300)
    assert ex.returncode == 0
    ev = np.array([[float(x) for x in ln.split()[:7]] for ln in ex.stdout.strip().splitlines() if ln and ln[0].isdigit()])
    np.testing.assert_allclose(ev, vals, atol=1e-6)


@pytest.mark.gpu
def test_public_cpp_surface_matches_oracle(host_lib, tmp_path, orc_mod):
    """The parts of the reference's public C++ surface no ros_rebvio caller reaches - rebvio::DistanceField (build ->
    operator[]), Core::testfk / calculatefJ / updateInverseDepthARLU, EdgeMap::searchMatch, ScaleSpace, FastGaussian, the
    timer macros - through tests/cpp/test_public_surface.cpp, every result bit-exact against the oracle's restatement."""
    import ctypes as C
    from rebvio_amd import synth
    W, H = 320, 240
    frames, cam = synth.render_stream(W, H, 2)
    frames.tofile(tmp_path / "frames.u8")
    exe = str(tmp_path / "public_surface")
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off"] + INC + [os.path.join(ROOT, "tests", "cpp", "test_public_surface.cpp"),
                    "-o", exe, "-L", host_lib, "-lrebvio", "-lrebvio_hip", f"-Wl,-rpath,{host_lib}", "-pthread"], check=True)
    r = subprocess.run([exe, str(tmp_path / "frames.u8"), str(W), str(H), repr(cam.fm), repr(cam.cx), repr(cam.cy), "3000", "4000",
                        str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "[rebvio timer]" in r.stdout and "inner" in r.stdout  # the static timers reported at exit

    def load(name, dtype):
        return np.fromfile(tmp_path / name, dtype=dtype)

    O = orc_mod
    L = O.lib()
    orc = O.Oracle(O.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=3000, keylines_max=4000))
    om = [orc.detect_u8(frames[i], i * 50000) for i in range(2)]
    ko = [m.keylines() for m in om]
    n0, n1 = len(ko[0]), len(ko[1])
    for i in range(2):
        kg = load(f"kl{i}.bin", O.KEYLINE_DTYPE)
        assert len(kg) == len(ko[i])
        for fld in ("pos", "pos_img", "gradient", "gradient_norm", "rho", "sigma_rho", "id_prev", "id_next"):
            assert np.array_equal(kg[fld].view(np.uint32), ko[i][fld].view(np.uint32)), (i, fld)

    # DistanceField::build -> operator[]
    orc.build_distance_field(om[1])
    ido, dso = orc.distance_field()
    idg, dsg = load("df_id.bin", np.int32).reshape(H, W), load("df_dist.bin", np.int32).reshape(H, W)
    assert np.array_equal(ido, idg) and (ido >= 0).sum() > 10000
    assert np.array_equal(dso[ido >= 0], dsg[idg >= 0])
    assert (dsg[idg < 0] == np.iinfo(np.int32).max).all()

    # Core::calculatefJ / testfk
    K = 400
    fj, fji = load("fj.bin", np.float32).reshape(K, 4), load("fji.bin", np.int32).reshape(K, 3)
    f32 = np.float32
    matched = 0
    for k in range(K):
        kl = ko[0][(k * 37) % n0:(k * 37) % n0 + 1].copy()
        kl["sigma_rho"] = f32(0.5) + f32(0.01) * f32(k)
        x, y = int(f32(kl["pos"][0, 0]) + f32(0.5)), int(f32(kl["pos"][0, 1]) + f32(0.5))
        dx, dy, fi, mnum = C.c_float(-1), C.c_float(-1), C.c_float(123.0), C.c_int(0)
        rr = L.orc_calculate_fj(orc.h, y * W + x, C.byref(dx), C.byref(dy), kl.ctypes.data, float(kl["pos"][0, 0]),
                                float(kl["pos"][0, 1]), C.byref(mnum), C.byref(fi))
        want = np.array([rr, dx.value, dy.value, fi.value], np.float32)
        assert np.array_equal(want.view(np.uint32), fj[k].view(np.uint32)), (k, want, fj[k])
        assert (mnum.value, int(kl["match_id_forward"][0])) == (fji[k, 0], fji[k, 1]), k
        other = ko[1][(k * 11) % n1:(k * 11) % n1 + 1]
        assert L.orc_test_fk(other.ctypes.data, kl.ctypes.data, 0.5) == fji[k, 2], k
        matched += mnum.value
    assert 50 < matched < K  # both outcomes of the lookup occur

    # EdgeMap::searchMatch
    vel = np.array([0.004, -0.002, 0.003], np.float32)
    Rvel = np.diag([f32(1e-6) * f32(i + 1) for i in range(3)]).astype(np.float32)
    Rback = np.eye(3, dtype=np.float32)
    Rback[0, 2], Rback[2, 0] = 2e-3, -2e-3
    sm, sm0 = load("sm.bin", np.int32), load("sm0.bin", np.int32)
    want, want0 = [], []
    for k in range(K):
        q = ko[1][(k * 29) % n1:(k * 29) % n1 + 1].copy()
        want0.append(orc.search_match(om[0], q, np.zeros(3), Rvel, np.eye(3)))
        q["rho"] = f32(0.4) + f32(0.002) * f32(k)
        q["sigma_rho"] = f32(0.05) + f32(0.01) * f32(k % 50)
        want.append(orc.search_match(om[0], q, vel, Rvel, Rback))
    assert np.array_equal(sm, np.array(want, np.int32)) and (sm >= 0).sum() > 20 and (sm < 0).sum() > 20
    assert np.array_equal(sm0, np.array(want0, np.int32)) and (sm0 >= 0).sum() > 20

    # Core::updateInverseDepthARLU (both clamps occur)
    ekf = load("ekf.bin", np.float32).reshape(K, 2)
    vs = [np.array([0.01, -0.004, z], np.float32) for z in (0.02, -0.01, 200.0)]
    for k in range(K):
        v = vs[1 if k % 7 == 0 else (2 if k % 7 == 1 else 0)]
        kl = ko[1][(k * 13) % n1:(k * 13) % n1 + 1].copy()
        kl["match_pos_img"][0, 0] = kl["pos_img"][0, 0] + (f32(0.5) - f32(0.01) * f32(k % 90))
        kl["match_pos_img"][0, 1] = kl["pos_img"][0, 1] + (f32(-0.3) + f32(0.02) * f32(k % 40))
        kl["match_gradient"], kl["match_gradient_norm"], kl["match_id"] = kl["gradient"], kl["gradient_norm"], 1
        kl["rho"] = f32(19.99) if k % 7 == 0 else (f32(0.0011) if k % 7 == 1 else f32(0.002) + f32(0.03) * f32(k))
        kl["sigma_rho"] = f32(0.001) if k % 7 <= 1 else f32(0.01) + f32(0.05) * f32(k % 100)
        L.orc_update_inverse_depth_arlu(orc.h, kl.ctypes.data, v.ctypes.data_as(C.POINTER(C.c_float)))
        want = np.array([kl["rho"][0], kl["sigma_rho"][0]], np.float32)
        assert np.array_equal(want.view(np.uint32), ekf[k].view(np.uint32)), (k, want, ekf[k])
    assert (ekf[:, 0] == np.float32(20.0)).any() and (ekf[:, 0] == np.float32(1e-3)).any()

    # ScaleSpace / FastGaussian
    img = frames[0].astype(np.float32) * np.float32(3.0)
    so = orc.scale_space(img)
    assert np.array_equal(so["dog"].view(np.uint32).ravel(), load("dog.bin", np.uint32))
    assert np.array_equal(so["mag"].view(np.uint32).ravel(), load("mag.bin", np.uint32))
    sm_o, widths = orc.smooth(img, 2.2)
    meta = load("smooth_meta.bin", np.float32)
    assert [int(v) for v in meta[3:6]] == widths and meta[0] == 3 and meta[1] == np.float32(2.2)
    assert abs(meta[2] - np.sqrt((sum(w * w for w in widths) - 3) / 12.0)) < 1e-5
    assert np.array_equal(sm_o.view(np.uint32).ravel(), load("smooth.bin", np.uint32))
    d, h = widths[0], widths[0] // 2
    ext = lambda i, n: np.where(i <= h, i + h + 1, np.where(i >= n - h, n - i + h, d))  # noqa: E731
    cnt = ext(np.arange(H), H)[:, None] * ext(np.arange(W), W)[None, :]
    assert np.array_equal((1.0 / cnt.astype(np.float32).astype(np.float64)).astype(np.float32).view(np.uint32).ravel(),
                          load("div0.bin", np.uint32))
