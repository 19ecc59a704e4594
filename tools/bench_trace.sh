#!/bin/bash
# kernel stats of the default bench command under rocprofv3 -> gpurun_out/<tag>/{kernel_stats.txt,kernel_stats.csv,bench.log}
TAG=${1:-trace}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 bench.py --steps 600 --warmup 1000 --no-cpu-baseline --lanes 0 --no-host-class "$@" > $OUT/bench.log 2>&1
f=$(ls $OUT/raw/*/*_kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv; python3 tools/kstats.py $f > $OUT/kernel_stats.txt
t=$(ls $OUT/raw/*/*_kernel_trace.csv | head -1); python3 tools/trace_gaps.py $t > $OUT/gaps.txt 2>&1; cat $OUT/gaps.txt
rm -rf $OUT/raw
head -16 $OUT/kernel_stats.txt
