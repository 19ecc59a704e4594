#!/usr/bin/env python3
"""rebvio::Rebvio (rebvio_replay) on the bench's 640x480 stream with the host timers on, under a few environment variants:
per-pair times of the fusion thread, per-frame time of the acquisition thread, wall-clock rate between first and last record."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rebvio_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
variants = [v for v in sys.argv[2:]] or ["", "REBVIO_HIP_PAIR_PRELAUNCH=1"]
frames, cam = synth.render_stream(640, 480, 24)
order = synth.pingpong_indices(24, n)
d = tempfile.mkdtemp()
frames[order].tofile(os.path.join(d, "f.u8"))
ts, gyro, acc = synth.imu_samples(synth.make_scene(0), n, noise_seed=1)
rec = np.zeros(len(ts), dtype=[("ts", "<i8"), ("gyro", "<f4", 3), ("acc", "<f4", 3)])
rec["ts"], rec["gyro"], rec["acc"] = ts, gyro * 0, acc
rec.tofile(os.path.join(d, "imu.bin"))
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rebvio_amd", "_build", "rebvio_replay")
for v in variants:
    env = dict(os.environ, REBVIO_HOST_TIMERS="1", REBVIO_HIP_DEBUG="1")
    for kv in v.split():
        k, _, val = kv.partition("=")
        env[k] = val
    try:
        r = subprocess.run([exe, "--raw", os.path.join(d, "f.u8"), "--size", "640", "480", "--imu", os.path.join(d, "imu.bin"), "--camera",
                            str(cam.fm), str(cam.cx), str(cam.cy), "--keylines", "15000", "16000", "--out", os.path.join(d, "o.txt")],
                           capture_output=True, text=True, env=env, timeout=120)
        lines = [ln for ln in r.stderr.splitlines() if "[Rebvio]" in ln or "[replay]" in ln or "frames=" in ln or "track_pair_begin" in ln]
        print(f"--- {v or 'default'} (rc {r.returncode})\n" + "\n".join(lines[-5:]), flush=True)
    except subprocess.TimeoutExpired as e:
        print(f"--- {v or 'default'}: TIMEOUT after 120 s\n" + (e.stderr or b"").decode()[-600:], flush=True)
