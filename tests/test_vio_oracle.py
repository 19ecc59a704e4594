"""Oracle restatement of the inertial half (BASELINE config 5): analytic behaviour of the scale / attitude /
bias filter on the synthetic stream whose ground truth is known (rebvio_amd/synth.py). CPU only."""
import numpy as np

from rebvio_amd import synth


def _run(orc_mod, n, W, H, gyro_bias=(0, 0, 0), noise_seed=1):
    frames, cam = synth.render_stream(W, H, n)
    scene = synth.make_scene(0)
    ts, gyro, acc = synth.imu_samples(scene, n, noise_seed=noise_seed, gyro_bias=gyro_bias)
    p = orc_mod.default_params(H, W, fm=cam.fm, cx=cam.cx, cy=cam.cy, keylines_ref=1500, keylines_max=2500,
                               global_min_matches_threshold=100)
    orc = orc_mod.Oracle(p)
    orc.vio_reset()
    prev, k, outs = None, 0, []
    for i in range(n):
        m = orc.detect_u8(frames[i], i * 50000)
        while k < len(ts) and ts[k] <= i * 50000:
            orc.vio_add_imu(m, ts[k], gyro[k], acc[k])
            k += 1
        if prev is not None:
            outs.append(orc.vio_step(prev, m))
        prev = m
    return scene, outs


def test_vio_initialisation_and_filter_schedule(orc_mod):
    scene, outs = _run(orc_mod, 22, 192, 144)
    init = [o.initialized for o in outs]
    sab = [o.sab_active for o in outs]
    # gyro bias initialised once num_gyro_init exceeds init_bias_frame_num = 10 (rebvio.cpp:150), i.e. on the 12th pair
    assert init.index(1) == 11 and all(init[11:]) and not any(init[:11])
    # SAB branch from num_frames > 4 + init_bias_frame_num (rebvio.cpp:210): the 16th pair on
    assert sab.index(1) == 15 and all(sab[15:])
    # before that the pose is not integrated (rebvio.cpp:263)
    for o in outs[:15]:
        assert tuple(o.position) == (0.0, 0.0, 0.0) and o.K == 1.0
    # Bg initialises to the mean integrated gyro = yaw rate per frame about y
    yaw_per_frame = np.radians(scene.yaw_deg)
    assert abs(outs[11].Bg[1] - yaw_per_frame) < 2e-4 and abs(outs[11].Bg[0]) < 2e-4 and abs(outs[11].Bg[2]) < 2e-4


def test_vio_gravity_scale_and_pose(orc_mod):
    scene, outs = _run(orc_mod, 26, 192, 144)
    last = outs[-1]
    g = np.array(last.g_est)
    assert abs(np.linalg.norm(g) - 9.81) < 0.05 and g[1] > 9.7     # gravity along +y of the camera
    assert np.isfinite(last.K) and last.K >= 0.0
    pos = np.array([o.position for o in outs[15:]])
    assert np.isfinite(pos).all() and np.abs(pos[-1]).max() > 0     # pose integrates once the filter is on
    ori = np.array([o.orientation for o in outs])
    assert np.isfinite(ori).all()
