#!/bin/bash
# Collects the round's profile records on the GPU box into gpurun_out/prof_rNN (copy the summaries into profiles/ afterwards):
#   kernel stats of the default bench command (steady state), of config c3 and of a 4-lane batch; two PMC passes (HBM bytes)
set -e
R=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "kernel trace c2"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 bench.py --steps 600 --warmup 1000 --no-cpu-baseline --lanes 0 --no-host-class > $OUT/c2.log 2>&1
echo "kernel trace c3"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 bench.py --config c3 --steps 300 --no-cpu-baseline --lanes 0 --no-host-class > $OUT/c3.log 2>&1
echo "kernel trace batch4"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b4 -- python3 tools/batch_rate.py 4 400 800 > $OUT/b4.log 2>&1
echo "pmc fetch"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 200 --warmup 400 --no-cpu-baseline --lanes 0 --no-host-class > $OUT/pmc_fetch.log 2>&1
echo "pmc write"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 200 --warmup 400 --no-cpu-baseline --lanes 0 --no-host-class > $OUT/pmc_write.log 2>&1
for d in c2 c3 b4; do f=$(ls $OUT/$d/*/*_kernel_stats.csv | head -1); cp $f $OUT/${d}_kernel_stats.csv; python3 tools/kstats.py $f > $OUT/${d}_kernel_stats.txt; done
python3 tools/pmc_summary.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) > $OUT/pmc_hbm.json
rm -rf $OUT/c2/*/*kernel_trace.csv $OUT/c3/*/*kernel_trace.csv $OUT/b4/*/*kernel_trace.csv $OUT/pmc_fetch $OUT/pmc_write
head -14 $OUT/c2_kernel_stats.txt
