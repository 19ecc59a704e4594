#!/usr/bin/env python3
"""Records the inputs and outputs of every Core::estimateBias call of a rebvio::Rebvio run (REBVIO_DUMP_FUSION, see
rebvio_amd/host/rebvio.cpp) on the bench's 640x480 stream with its synthetic IMU, and keeps a sample of them as
tests/golden/estimate_bias_calls.npz: the first 40 calls (the filter's start-up, where most solves of the Gauss-Newton go
through the pseudo-inverse) and every 20th after that. Needs a GPU (the calls' inputs come out of the device's first halves).

The committed file was recorded with the DENSE form of SABEstimator::problem / Core::estimateBias (11x11 and 7x7 products as
the reference writes them, the commit before the structured form): tests/test_fusion_math.py::test_estimate_bias_replays_recorded_calls
holds the structured form to those records bit for bit.

  python tools/record_fusion_calls.py [frames=3100] [out.npz]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rebvio_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "estimate_bias_calls.npz")
frames, cam = synth.render_stream(640, 480, 24)
d = tempfile.mkdtemp()
frames[synth.pingpong_indices(24, n)].tofile(os.path.join(d, "f.u8"))
ts, gyro, acc = synth.imu_samples(synth.make_scene(0), n, noise_seed=1)
rec = np.zeros(len(ts), dtype=[("ts", "<i8"), ("gyro", "<f4", 3), ("acc", "<f4", 3)])
rec["ts"], rec["gyro"], rec["acc"] = ts, gyro * 0, acc
rec.tofile(os.path.join(d, "imu.bin"))
exe = os.path.join(ROOT, "rebvio_amd", "_build", "rebvio_replay")
dump = os.path.join(d, "calls.f32")
subprocess.run([exe, "--raw", os.path.join(d, "f.u8"), "--size", "640", "480", "--imu", os.path.join(d, "imu.bin"), "--camera",
                str(cam.fm), str(cam.cx), str(cam.cy), "--keylines", "15000", "16000", "--out", os.path.join(d, "o.txt")],
               check=True, env=dict(os.environ, REBVIO_DUMP_FUSION=dump), timeout=300)
calls = np.fromfile(dump, np.float32).reshape(-1, 237)
idx = sorted(set(list(range(40)) + list(range(40, min(3000, len(calls)), 20))))
np.savez_compressed(out, call_index=np.array(idx, np.int32), inputs=calls[idx, :168], outputs=calls[idx, 168:])
print(f"{len(calls)} calls recorded, {len(idx)} kept in {out}")
