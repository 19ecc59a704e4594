"""Diagnostic: frames/s of the C++ host class rebvio::Rebvio (camera + IMU, full fusion) replaying a raw 640x480 stream."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rebvio_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
frames, cam = synth.render_stream(640, 480, 24)
order = synth.pingpong_indices(24, n)
d = tempfile.mkdtemp()
frames[order].tofile(os.path.join(d, "f.u8"))
scene = synth.make_scene(0)
ts, gyro, acc = synth.imu_samples(scene, n, noise_seed=1)
rec = np.zeros(len(ts), dtype=[("ts", "<i8"), ("gyro", "<f4", 3), ("acc", "<f4", 3)])
rec["ts"], rec["gyro"], rec["acc"] = ts, gyro * 0, acc      # ping-pong replay: keep the gyro still
rec.tofile(os.path.join(d, "imu.bin"))
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rebvio_amd", "_build", "rebvio_replay")
t0 = time.perf_counter()
r = subprocess.run([exe, "--raw", os.path.join(d, "f.u8"), "--size", "640", "480", "--imu", os.path.join(d, "imu.bin"), "--camera",
                    str(cam.fm), str(cam.cx), str(cam.cy), "--keylines", "15000", "16000", "--out", os.path.join(d, "o.txt")],
                   capture_output=True, text=True)
dt = time.perf_counter() - t0
print('\n'.join(r.stderr.strip().splitlines()[-4:]))
print("rebvio::Rebvio replay: %d frames in %.2f s (process start, file reads and context creation included) = %.0f frames/s" % (n, dt, n / dt))
