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


def test_host_library_exports_reference_classes(host_lib):
    out = subprocess.run(["nm", "-DC", os.path.join(host_lib, "librebvio.so")], capture_output=True, text=True, check=True).stdout
    for sym in ("rebvio::Rebvio::Rebvio(rebvio::RebvioConfig&)", "rebvio::Rebvio::imageCallback(rebvio::types::Image&&)",
                "rebvio::Rebvio::imuCallback(rebvio::types::Imu&&)", "rebvio::EdgeDetector::detect(rebvio::types::Image&)",
                "rebvio::EdgeMap::rotateKeylines", "rebvio::EdgeMap::directedMatch", "rebvio::EdgeMap::regularize1Iter",
                "rebvio::Core::minimizeVel", "rebvio::Core::extRotVel", "rebvio::Core::updateInverseDepth",
                "rebvio::Core::buildDistanceField", "rebvio::Core::tryVel"):
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
    assert (vals[:15, 1:] == 0).all()      # pose integration starts after 4 + init_bias_frame_num frames (rebvio.cpp:263)
    assert np.abs(vals[-1, 4:]).max() > 0  # and then moves
    assert "running=1" in r.stderr
