#!/bin/bash
# Kernel trace of rebvio::Rebvio (rebvio_replay) on the bench's stream: per-queue timeline of a window in steady state.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/host_trace
D=/tmp/host_trace
mkdir -p $OUT $D
cd $GRAFT_REPO_ROOT
python3 - <<PY
import numpy as np, os
from rebvio_amd import synth
n = 1500
frames, cam = synth.render_stream(640, 480, 24)
frames[synth.pingpong_indices(24, n)].tofile("$D/f.u8")
ts, gyro, acc = synth.imu_samples(synth.make_scene(0), n, noise_seed=1)
rec = np.zeros(len(ts), dtype=[("ts", "<i8"), ("gyro", "<f4", 3), ("acc", "<f4", 3)])
rec["ts"], rec["gyro"], rec["acc"] = ts, gyro * 0, acc
rec.tofile("$D/imu.bin")
open("$D/cam.txt", "w").write(f"{cam.fm} {cam.cx} {cam.cy}")
PY
read FM CX CY < $D/cam.txt || true
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/rebvio_amd/_build:$LD_LIBRARY_PATH
rocprofv3 --kernel-trace --output-format csv -d $D/t -- $GRAFT_REPO_ROOT/rebvio_amd/_build/rebvio_replay --raw $D/f.u8 --size 640 480 --imu $D/imu.bin --camera $FM $CX $CY --keylines 15000 16000 --out $D/o.txt > $OUT/run.log 2>&1
f=$(ls $D/t/*/*kernel_trace.csv | head -1)
python3 tools/timeline.py $f 12000 60 > $OUT/window.txt
rm -rf $D
tail -3 $OUT/run.log
cat $OUT/window.txt
