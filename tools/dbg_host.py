import os, sys, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
from rebvio_amd import synth
n = 20
frames, cam = synth.render_stream(320, 240, n)
os.makedirs("gpurun_out/r2i", exist_ok=True)
p = "gpurun_out/r2i/frames.u8"
frames.tofile(p)
exe = "rebvio_amd/_build/rebvio_stream_example"
for env in ({"REBVIO_HIP_PRELAUNCH": "0"}, {}):
    try:
        r = subprocess.run([exe, p, "320", "240", str(n), str(cam.fm), str(cam.cx), str(cam.cy), "3000", "4000"], capture_output=True, text=True,
                           timeout=40, env=dict(os.environ, **env))
        print(env, "rc", r.returncode, r.stderr[-300:], flush=True)
    except subprocess.TimeoutExpired as e:
        print(env, "TIMEOUT", (e.stderr or b"")[-600:], flush=True)
