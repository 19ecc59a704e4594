#!/bin/bash
# Kernel trace of the default bench in steady state: a window of the per-queue timeline (tools/timeline.py).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_trace
D=/tmp/bench_trace
mkdir -p $OUT $D
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $D/t -- python3 bench.py --steps 600 --warmup 1000 --no-cpu-baseline --lanes 0 --no-host-class > $OUT/run.log 2>&1
f=$(ls $D/t/*/*kernel_trace.csv | head -1)
n=$(wc -l < $f)
python3 tools/timeline.py $f $((n - 4000)) 90 > $OUT/window.txt
rm -rf $D
cat $OUT/window.txt
